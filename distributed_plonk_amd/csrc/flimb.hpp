// flimb.hpp — generic "unsaturated limb" Montgomery arithmetic: NL limbs of B bits in 32-bit registers,
// R' = 2^(B*NL).  Used for the base-field work of the MSM bucket accumulation:
//     BN254 Fq      : NL = 9,  B = 29  (R' = 2^261)
//     BLS12-381 Fq  : NL = 14, B = 28  (R' = 2^392)
// Rationale and measurements: see fp29.hpp (v_mad_u64_u32 issues as fast as an add-with-carry on
// gfx950, so carry-free product scanning halves the instruction count of a saturated CIOS multiply).
//
// Values are residues kept LAZILY: limbs normalised (< 2^B, excess in the top limb) but the integer
// may exceed p by a small bounded factor.  Every subtraction a - b is computed as a + K*p - b with a
// pre-lifted constant K*p (each limb raised by 2^31, borrowed from the next) so no limb underflows.
// The curve formulas in ec_lazy.hpp carry the bound bookkeeping; results are canonicalised before
// they leave the kernel, so what reaches HBM is the same fully reduced R = 2^(32N) Montgomery form
// the reference uses (utils.rs:27-43).
#pragma once
#include <stdint.h>
#include "fp.hpp"

template <int NL, int B> struct FL { uint32_t l[NL]; };

// Pins a column accumulator after every v_mad_u64_u32 so the products of a column stay ONE chain; hipcc otherwise
// reassociates them into parallel partial sums and joins them with 64-bit adds that cost as much as the mads.
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(PLONK_PIN_NONE)
#define FL_CHAIN(acc) ((void)0)
#elif defined(PLONK_PIN_USE)
#define FL_CHAIN(acc) asm volatile("" : : "v"(acc))
#else
#define FL_CHAIN(acc) asm("" : "+v"(acc))
#endif
#else
#define FL_CHAIN(acc) ((void)0)
#endif

template <int NL, int B> struct FLParams {
    uint32_t p[NL];       // modulus, normalised
    uint32_t p2[NL];      // 2p, normalised (zero tests)
    uint32_t c2[NL];      // 2p lifted   (a + c2 - b, b < 2p,  b limbs < 2^31)
    uint32_t c4[NL];      // 4p lifted
    uint32_t c8[NL];      // 8p lifted
    uint32_t one[NL];     // R' mod p                         (the field's 1 in R' form)
    uint32_t r_std[NL];   // 2^(32N) mod p     : x*R' --mul--> x*2^(32N)
    uint32_t r2fix[NL];   // R'^2 / 2^(32N)    : x*2^(32N) --mul--> x*R'
    uint32_t inv;         // -p^{-1} mod 2^B
};

template <int NL, int B> FP_HD FL<NL, B> fl_zero() {
    FL<NL, B> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = 0;
    return r;
}
template <int NL, int B> FP_HD FL<NL, B> fl_load_const(const uint32_t* c) {
    FL<NL, B> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = c[i];
    return r;
}
template <int NL, int B> FP_HD bool fl_all_zero(const FL<NL, B>& a) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) t |= a.l[i];
    return t == 0;
}
template <int NL, int B> FP_HD bool fl_equals_const(const FL<NL, B>& a, const uint32_t* c) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) t |= a.l[i] ^ c[i];
    return t == 0;
}

// saturated (N x 32) <-> limbs.  to_sat needs normalised limbs and value < 2^(32N).
template <int NL, int B, int N> FP_HD FL<NL, B> fl_from_sat(const Fp<N>& a) {
    FL<NL, B> r;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const int bit = B * k, w = bit >> 5, off = bit & 31;
        uint32_t v = (w < N) ? (a.l[w] >> off) : 0;
        if (off + B > 32 && w + 1 < N) v |= a.l[w + 1] << (32 - off);
        r.l[k] = (k == NL - 1) ? v : (v & ((1u << B) - 1));
    }
    return r;
}
template <int NL, int B, int N> FP_HD Fp<N> fl_to_sat(const FL<NL, B>& a) {
    Fp<N> r;
#pragma unroll
    for (int j = 0; j < N; j++) {
        const int bit = 32 * j, k = bit / B, sh = bit - B * k;
        uint32_t v = a.l[k] >> sh;
        if (k + 1 < NL) v |= a.l[k + 1] << (B - sh);
        if (k + 2 < NL && 2 * B - sh < 32) v |= a.l[k + 2] << (2 * B - sh);
        r.l[j] = v;
    }
    return r;
}

template <int NL, int B> FP_HD void fl_norm(FL<NL, B>& a) {
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        a.l[i + 1] += a.l[i] >> B;
        a.l[i] &= (1u << B) - 1;
    }
}
template <int NL, int B> FP_HD FL<NL, B> fl_add(const FL<NL, B>& a, const FL<NL, B>& b) {
    FL<NL, B> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a + C - b, C a lifted multiple of p; a normalised, b limbs < 2^31.  Result is NOT normalised.
template <int NL, int B> FP_HD FL<NL, B> fl_sub(const FL<NL, B>& a, const FL<NL, B>& b, const uint32_t* C) {
    FL<NL, B> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + C[i] - b.l[i];
    return r;
}

// Montgomery product x*y/R' mod p; both operands normalised (one may have limbs < 2^31 when B*2+4+log2(NL) allows).
// Result normalised, value < x*y/R' + p.
template <int NL, int B> FP_HD FL<NL, B> fl_mul(const FL<NL, B>& x, const FL<NL, B>& y, const FLParams<NL, B>& P) {
    constexpr uint32_t MASK = (1u << B) - 1;
    uint64_t acc = 0;
    uint32_t m[NL];
    FL<NL, B> r;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)x.l[i] * y.l[k - i]; FL_CHAIN(acc); }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * P.p[k - i]; FL_CHAIN(acc); }
        m[k] = ((uint32_t)acc * P.inv) & MASK;
        { acc += (uint64_t)m[k] * P.p[0]; FL_CHAIN(acc); }
        acc >>= B;
        FL_CHAIN(acc);
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc += (uint64_t)x.l[i] * y.l[k - i]; FL_CHAIN(acc); }
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc += (uint64_t)m[i] * P.p[k - i]; FL_CHAIN(acc); }
        r.l[k - NL] = (uint32_t)acc & MASK;
        acc >>= B;
        FL_CHAIN(acc);
    }
    r.l[NL - 1] = (uint32_t)acc;
    return r;
}

// (x0*y0 + x1*y1) / R' mod p with ONE Montgomery reduction (saves NL^2 + NL mads against two products).  All four operands
// normalised (limbs < 2^B; the top limbs may be larger as long as every column — 2*NL products below 2^(2B) plus NL reduction
// products — stays below 2^64: 27 * 2^58 for NL = 9, B = 29; 42 * 2^56 for NL = 14, B = 28).  Result normalised, < sum/R' + p.
template <int NL, int B> FP_HD FL<NL, B> fl_dot2(const FL<NL, B>& x0, const FL<NL, B>& y0, const FL<NL, B>& x1, const FL<NL, B>& y1,
                                                 const FLParams<NL, B>& P) {
    constexpr uint32_t MASK = (1u << B) - 1;
    uint64_t acc = 0;
    uint32_t m[NL];
    FL<NL, B> r;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)x0.l[i] * y0.l[k - i]; FL_CHAIN(acc); }
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)x1.l[i] * y1.l[k - i]; FL_CHAIN(acc); }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * P.p[k - i]; FL_CHAIN(acc); }
        m[k] = ((uint32_t)acc * P.inv) & MASK;
        { acc += (uint64_t)m[k] * P.p[0]; FL_CHAIN(acc); }
        acc >>= B;
        FL_CHAIN(acc);
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc += (uint64_t)x0.l[i] * y0.l[k - i]; FL_CHAIN(acc); }
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc += (uint64_t)x1.l[i] * y1.l[k - i]; FL_CHAIN(acc); }
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc += (uint64_t)m[i] * P.p[k - i]; FL_CHAIN(acc); }
        r.l[k - NL] = (uint32_t)acc & MASK;
        acc >>= B;
        FL_CHAIN(acc);
    }
    r.l[NL - 1] = (uint32_t)acc;
    return r;
}

// Montgomery square: the cross products x_i*x_j (i != j) are taken once against the doubled limb.
// x normalised (limbs < 2^B, top limb may be larger but 2*x_top must stay below 2^31).
template <int NL, int B> FP_HD FL<NL, B> fl_sqr(const FL<NL, B>& x, const FLParams<NL, B>& P) {
    constexpr uint32_t MASK = (1u << B) - 1;
    uint64_t acc = 0;
    uint32_t m[NL], x2[NL];
    FL<NL, B> r;
#pragma unroll
    for (int i = 0; i < NL; i++) x2[i] = x.l[i] << 1;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) { acc += (uint64_t)x.l[i] * x2[k - i]; FL_CHAIN(acc); }
        if ((k & 1) == 0) { acc += (uint64_t)x.l[k / 2] * x.l[k / 2]; FL_CHAIN(acc); }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * P.p[k - i]; FL_CHAIN(acc); }
        m[k] = ((uint32_t)acc * P.inv) & MASK;
        { acc += (uint64_t)m[k] * P.p[0]; FL_CHAIN(acc); }
        acc >>= B;
        FL_CHAIN(acc);
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; 2 * i < k; i++) { acc += (uint64_t)x.l[i] * x2[k - i]; FL_CHAIN(acc); }
        if ((k & 1) == 0) { acc += (uint64_t)x.l[k / 2] * x.l[k / 2]; FL_CHAIN(acc); }
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc += (uint64_t)m[i] * P.p[k - i]; FL_CHAIN(acc); }
        r.l[k - NL] = (uint32_t)acc & MASK;
        acc >>= B;
        FL_CHAIN(acc);
    }
    r.l[NL - 1] = (uint32_t)acc;
    return r;
}

// value (normalised) == 0 mod p, given value < 3p
template <int NL, int B> FP_HD bool fl_is_zero_mod_p_lt3p(const FL<NL, B>& a, const FLParams<NL, B>& P) {
    return fl_all_zero(a) || fl_equals_const(a, P.p) || fl_equals_const(a, P.p2);
}

// one conditional subtraction: normalised a < 2p -> canonical
template <int NL, int B> FP_HD FL<NL, B> fl_canon_lt2p(const FL<NL, B>& a, const FLParams<NL, B>& P) {
    constexpr uint32_t MASK = (1u << B) - 1;
    FL<NL, B> d;
    int32_t br = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int32_t t = (int32_t)a.l[i] - (int32_t)P.p[i] + br;
        br = t >> 31;
        d.l[i] = (i == NL - 1) ? (uint32_t)t : ((uint32_t)t & MASK);
    }
    FL<NL, B> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = br ? a.l[i] : d.l[i];
    return r;
}
// full canonicalisation of a small multiple (cold paths): up to `maxk` conditional subtractions
template <int NL, int B> FP_HD FL<NL, B> fl_canon_small(FL<NL, B> a, const FLParams<NL, B>& P, int maxk) {
    for (int k = 0; k < maxk; k++) a = fl_canon_lt2p(a, P);
    return a;
}

// ---- host-side parameter construction from the saturated field parameters
template <int NL, int B, int N> inline FLParams<NL, B> fl_make_params(const FpParams<N>& P) {
    FLParams<NL, B> q;
    constexpr uint32_t MASK = (1u << B) - 1;
    Fp<N> pm;
    for (int i = 0; i < N; i++) pm.l[i] = P.p[i];
    FL<NL, B> p29 = fl_from_sat<NL, B, N>(pm);
    for (int i = 0; i < NL; i++) q.p[i] = p29.l[i];
    auto mult = [&](uint32_t K, uint32_t* out) {        // K*p, normalised
        uint64_t carry = 0;
        for (int i = 0; i < NL; i++) {
            uint64_t v = (uint64_t)K * q.p[i] + carry;
            if (i < NL - 1) { out[i] = (uint32_t)(v & MASK); carry = v >> B; } else out[i] = (uint32_t)v;
        }
    };
    auto lift = [&](uint32_t K, uint32_t* out) {         // K*p with every limb raised by 2^31
        uint32_t d[NL];
        mult(K, d);
        const uint32_t borrow = 1u << (31 - B);
        out[0] = d[0] + (1u << 31);
        for (int i = 1; i < NL - 1; i++) out[i] = d[i] + (1u << 31) - borrow;
        out[NL - 1] = d[NL - 1] - borrow;
    };
    mult(2, q.p2);
    lift(2, q.c2); lift(4, q.c4); lift(8, q.c8);
    uint32_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - q.p[0] * x;
    q.inv = (0u - x) & MASK;
    const int d = B * NL - 32 * N;
    Fp<N> o;
    for (int i = 0; i < N; i++) o.l[i] = P.one[i];                      // 2^(32N) mod p
    FL<NL, B> t = fl_from_sat<NL, B, N>(o);
    for (int i = 0; i < NL; i++) q.r_std[i] = t.l[i];
    for (int i = 0; i < d; i++) o = fp_add(o, o, P);                    // R' mod p
    t = fl_from_sat<NL, B, N>(o);
    for (int i = 0; i < NL; i++) q.one[i] = t.l[i];
    for (int i = 0; i < d; i++) o = fp_add(o, o, P);                    // R'^2 / 2^(32N) mod p
    t = fl_from_sat<NL, B, N>(o);
    for (int i = 0; i < NL; i++) q.r2fix[i] = t.l[i];
    return q;
}
