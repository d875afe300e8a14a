// fp29.hpp — 9 x 29-bit "unsaturated" limb arithmetic for the 254/255-bit scalar fields, gfx950 VALU.
//
// Why: measured on MI355X (profiles/r01_valu_microbench.txt) v_mad_u64_u32 issues at the same rate
// (~4.5 clk per wave-instruction) as v_add_co/v_addc/v_lshl_add_u64, so a saturated 8x32 CIOS
// multiplier spends as many issue slots on carry plumbing (and v_mov zero-extensions) as on products
// (~550 VALU instructions).  With 29-bit limbs a column of nine 58-bit products plus the Montgomery
// correction fits a 64-bit accumulator: product scanning needs ONE v_mad_u64_u32 per limb product and
// no carries at all (162 mads + 9 v_mul_lo + 17 shifts/masks ~ 205 instructions), additions are nine
// independent v_add_u32, and reduction is lazy.
//
// Semantics: HBM keeps the reference's representation (8 x u32, a*2^256 mod p, fully reduced,
// utils.rs:27-43).  Inside a kernel a value is re-limbed to 9 x 29 bits (same residue).  Every
// multiplication on the NTT path is data x precomputed constant; constants are stored as c*2^261 mod p,
// so  mont261(x, c*2^261) = x*c  keeps x's 2^256 factor untouched.  Values are canonicalised
// (conditional subtraction) before the final store, so results are bit-identical to ark-ff's.
//
// Bounds (p < 2^255, R = 2^261):  f29_mul accepts x < 2^261 (170 p on BN254, 70 p on BLS12-381's Fr) with limbs < 2^31 and a
// normalised constant w < p, and returns a normalised value < x*p/2^261 + p (< 1.36 p for x < 2^259.4, < 2 p at the limit).  x + C - t with C = "2p in borrow form"
// needs t normalised (a product).  Decimation-in-time butterflies grow the bound by at most 2p per
// stage, so up to ~20 stages run between canonicalisations.
#pragma once
#include <stdint.h>
#include "fp.hpp"

struct F29 { uint32_t l[9]; };

struct F29Params {
    uint32_t p[9];      // modulus, normalised 29-bit limbs
    uint32_t c2p[9];    // 2p with limbs lifted by 2^30 (borrowed from the next limb): every limb >= any product limb
    uint32_t c4p[9];    // 4p lifted the same way: for subtracting an un-normalised sum of two values (< 2.8p, limbs < 2^30)
    uint32_t one[9];    // 2^261 mod p  (the constant that multiplies by 1)
    uint32_t inv;       // -p^{-1} mod 2^29
    float inv_top;      // (1 - 2^-20) / (p[8] + 1): quotient estimate for f29_canon_lazy
    uint32_t pbar[9];   // 2^261 - p, normalised: the low half of -q*p for the precomputed-quotient multiplier (f29_mul_shoup)
};

// A constant prepared for the precomputed-quotient ("Shoup") multiplier: c < p as a PLAIN residue (the data operand keeps whatever
// Montgomery factor it carries) and cq = floor(c * 2^261 / p).  80 bytes = five 128-bit loads.
struct alignas(16) F29S {
    uint32_t c[9];
    uint32_t cq[9];
    uint32_t pad[2];
};

#define F29_MASK 0x1fffffffu
// Pins a column accumulator after every v_mad_u64_u32 so a column's products stay ONE chain: otherwise hipcc's SLP vectoriser treats the
// column as a horizontal reduction, splits it into parallel partial sums and joins them with 64-bit adds that cost as much as the mads
// (229 -> 213 VALU instructions per product).  Round 1 had to switch the pins off for the NTT pass kernel (F29_NO_PINS: "the
// pinned form spills at 128 VGPRs, 7.9 -> 12.6-22 ms per pass"); round 2 found the cause — not register pressure but the unroll
// budget: with one asm per mad the butterfly loops were no longer unrolled and the lane's element array went to scratch memory.
// build.py raises the budget for that translation unit and the pins are on everywhere (111 VGPRs, no scratch, -11.8 %).
// Round 4 (profiles/r04_pin_nop_experiment.txt): the "+v" pin DEFINES a VGPR in an inline asm, and hipcc's hazard recognizer then puts an
// `s_nop 0` in front of the next VALU that reads it — 9072 s_nop in the 2^8 NTT pass kernel, one per mad.  Three forms were measured per
// translation unit (build.py: build_variant); the default ships everywhere:
//   default          asm("" : "+v"(acc))            the round 1-3 form; its s_nops are hidden at 2 or 4 waves per SIMD (tools/mb), its ordering
//                                                   constraints keep the register pressure lowest, and it is the only form that is never slow
//   PLONK_PIN_USE    asm volatile("" :: "v"(acc))   a use, not a definition: no s_nop, 20 % less code; NTT and quotient kernels within 2 % of the
//                                                   default, but `volatile` costs the MSM accumulate kernel its SROA (a limb array in LDS, +50 %)
//   PLONK_PIN_NONE   nothing                        with -mllvm -slp-vectorize-hor=false (no splitting, so no pins needed): MSM kernels 5-25 %
//                                                   faster on one class of boxes, the accumulation 46 % slower on another; NTT +4 % (128 VGPRs)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(F29_NO_PINS)
#if defined(PLONK_PIN_NONE)
#define F29_CHAIN(acc) ((void)0)
#elif defined(PLONK_PIN_USE)
#define F29_CHAIN(acc) asm volatile("" : : "v"(acc))
#else
#define F29_CHAIN(acc) asm("" : "+v"(acc))
#endif
#else
#define F29_CHAIN(acc) ((void)0)
#endif

FP_HD F29 f29_from_sat(const Fp<8>& a) {
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, w = bit >> 5, off = bit & 31;
        uint32_t v = a.l[w] >> off;
        if (off + 29 > 32 && w + 1 < 8) v |= a.l[w + 1] << (32 - off);
        r.l[k] = (k == 8) ? v : (v & F29_MASK);
    }
    return r;
}

// normalised limbs, value < 2^256
FP_HD Fp<8> f29_to_sat(const F29& a) {
    Fp<8> r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, k = bit / 29, sh = bit - 29 * k;       // word j starts inside limb k at bit sh
        uint32_t v = a.l[k] >> sh;
        v |= a.l[k + 1] << (29 - sh);
        r.l[j] = v;
    }
    return r;
}

// carry pass: limbs 0..7 back below 2^29, excess accumulates in limb 8
FP_HD void f29_norm(F29& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a.l[i + 1] += a.l[i] >> 29;
        a.l[i] &= F29_MASK;
    }
}

FP_HD F29 f29_add(const F29& a, const F29& b) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a + 2p - t   (t normalised, t < 2p)
FP_HD F29 f29_sub2p(const F29& a, const F29& t, const F29Params& P) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + P.c2p[i] - t.l[i];
    return r;
}
// a + 4p - t   (t < 4p - with limbs <= 2^30 - 2, e.g. the lazy sum of two values below 1.4p)
FP_HD F29 f29_sub4p(const F29& a, const F29& t, const F29Params& P) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + P.c4p[i] - t.l[i];
    return r;
}

// Montgomery product x*w/2^261 mod p, product scanning with one 64-bit accumulator.
// x: limbs < 2^31, value < 2^261 ; w: normalised, < p.  Result normalised, < x*p/2^261 + p (< 1.36 p for x < 2^259.4; tests/test_fp29_host.py
// checks the host build against Python integers up to 40 p).
FP_HD F29 f29_mul(const F29& x, const F29& w, const F29Params& P) {
    uint64_t acc = 0;
    uint32_t m[9];
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)x.l[i] * w.l[k - i]; F29_CHAIN(acc); }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * P.p[k - i]; F29_CHAIN(acc); }
        m[k] = ((uint32_t)acc * P.inv) & F29_MASK;
        acc += (uint64_t)m[k] * P.p[0];
        acc >>= 29;
        F29_CHAIN(acc);
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) { acc += (uint64_t)x.l[i] * w.l[k - i]; F29_CHAIN(acc); }
#pragma unroll
        for (int i = k - 8; i <= 8; i++) { acc += (uint64_t)m[i] * P.p[k - i]; F29_CHAIN(acc); }
        r.l[k - 9] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
        F29_CHAIN(acc);
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// x * c mod p for a PREPARED constant (F29S): every NTT butterfly product is data x twiddle, so the quotient is precomputed too.
//     q = floor(x * cq / 2^261)  ->  floor(x*c/p) - 2 <= q <= floor(x*c/p)      (x < 2^261; the dropped low columns cost < 2^-20 of a unit)
//     r = x*c - q*p  in [0, 3p)  and only its low 261 bits are needed:  r = (x*c + q*pbar) mod 2^261,  pbar = 2^261 - p
// Cost: columns 7..16 of x*cq (53 limb products) + columns 0..8 of x*c and of q*pbar (45 + 45) = 143 limb products and no v_mul_lo,
// against 171 for the Montgomery product above (profiles/r02_multiplier_variants_static.txt: 213 -> 188 VALU instructions).
// x: limbs < 2^31, value < 2^261 (f29_mul's contract); result normalised, value < 3p — butterflies subtract it from 4p (c4p).
FP_HD F29 f29_mul_shoup(const F29& x, const uint32_t* c, const uint32_t* cq, const F29Params& P) {
    uint64_t acc = 0;
    uint32_t q[9];
#pragma unroll
    for (int k = 7; k < 17; k++) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) { acc += (uint64_t)x.l[i] * cq[k - i]; F29_CHAIN(acc); }
        if (k >= 9) q[k - 9] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
        F29_CHAIN(acc);
    }
    q[8] = (uint32_t)acc;
    F29 r;
    acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)x.l[i] * c[k - i]; F29_CHAIN(acc); }
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)q[i] * P.pbar[k - i]; F29_CHAIN(acc); }
        r.l[k] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
        F29_CHAIN(acc);
    }
    return r;
}

// Sum of up to three products with ONE Montgomery reduction:  (a0*b0 + a1*b1 + a2*b2) / 2^261 mod p.
// Every operand must be normalised (limbs 0..7 < 2^29, top limb < 2^29): a column then holds at most 27 products below 2^58 plus
// the 9 reduction products — 36 * 2^58 < 2^63.2 fits the 64-bit accumulator.  Result normalised, value < sum(a_i*b_i)/2^261 + p.
// Saves 81 + 9 mads per fused product: the quotient kernel's selector sums (dispatcher2.rs:459-477) are dot products.
template <int T>
FP_HD F29 f29_dot_n(const F29& a0, const F29& b0, const F29& a1, const F29& b1, const F29& a2, const F29& b2, const F29Params& P) {
    static_assert(T >= 1 && T <= 3, "at most three products fit one accumulator");
    uint64_t acc = 0;
    uint32_t m[9];
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)a0.l[i] * b0.l[k - i]; F29_CHAIN(acc); }
        if (T > 1) {
#pragma unroll
            for (int i = 0; i <= k; i++) { acc += (uint64_t)a1.l[i] * b1.l[k - i]; F29_CHAIN(acc); }
        }
        if (T > 2) {
#pragma unroll
            for (int i = 0; i <= k; i++) { acc += (uint64_t)a2.l[i] * b2.l[k - i]; F29_CHAIN(acc); }
        }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * P.p[k - i]; F29_CHAIN(acc); }
        m[k] = ((uint32_t)acc * P.inv) & F29_MASK;
        acc += (uint64_t)m[k] * P.p[0];
        acc >>= 29;
        F29_CHAIN(acc);
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) { acc += (uint64_t)a0.l[i] * b0.l[k - i]; F29_CHAIN(acc); }
        if (T > 1) {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) { acc += (uint64_t)a1.l[i] * b1.l[k - i]; F29_CHAIN(acc); }
        }
        if (T > 2) {
#pragma unroll
            for (int i = k - 8; i <= 8; i++) { acc += (uint64_t)a2.l[i] * b2.l[k - i]; F29_CHAIN(acc); }
        }
#pragma unroll
        for (int i = k - 8; i <= 8; i++) { acc += (uint64_t)m[i] * P.p[k - i]; F29_CHAIN(acc); }
        r.l[k - 9] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
        F29_CHAIN(acc);
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
FP_HD F29 f29_dot2(const F29& a0, const F29& b0, const F29& a1, const F29& b1, const F29Params& P) { return f29_dot_n<2>(a0, b0, a1, b1, a1, b1, P); }
FP_HD F29 f29_dot3(const F29& a0, const F29& b0, const F29& a1, const F29& b1, const F29& a2, const F29& b2, const F29Params& P) {
    return f29_dot_n<3>(a0, b0, a1, b1, a2, b2, P);
}
template <int T>
FP_HD F29 f29_dot(const F29* a, const F29* b, const F29Params& P) {
    return f29_dot_n<T>(a[0], b[0], a[T > 1 ? 1 : 0], b[T > 1 ? 1 : 0], a[T > 2 ? 2 : 0], b[T > 2 ? 2 : 0], P);
}

// Montgomery square x*x/2^261: the cross products x_i*x_j (i != j) are taken once against the doubled limb (45 + 81 mads).
// x normalised (limbs < 2^29; the doubled top limb must stay below 2^31).  Result normalised, < x^2/2^261 + p.
FP_HD F29 f29_sqr(const F29& x, const F29Params& P) {
    uint64_t acc = 0;
    uint32_t m[9], x2[9];
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) x2[i] = x.l[i] << 1;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) { acc += (uint64_t)x.l[i] * x2[k - i]; F29_CHAIN(acc); }
        if ((k & 1) == 0) { acc += (uint64_t)x.l[k / 2] * x.l[k / 2]; F29_CHAIN(acc); }
#pragma unroll
        for (int i = 0; i < k; i++) { acc += (uint64_t)m[i] * P.p[k - i]; F29_CHAIN(acc); }
        m[k] = ((uint32_t)acc * P.inv) & F29_MASK;
        acc += (uint64_t)m[k] * P.p[0];
        acc >>= 29;
        F29_CHAIN(acc);
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) { acc += (uint64_t)x.l[i] * x2[k - i]; F29_CHAIN(acc); }
        if ((k & 1) == 0) { acc += (uint64_t)x.l[k / 2] * x.l[k / 2]; F29_CHAIN(acc); }
#pragma unroll
        for (int i = k - 8; i <= 8; i++) { acc += (uint64_t)m[i] * P.p[k - i]; F29_CHAIN(acc); }
        r.l[k - 9] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
        F29_CHAIN(acc);
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// normalised a < 2p  ->  a mod p (canonical), still normalised
FP_HD F29 f29_canon(const F29& a, const F29Params& P) {
    F29 d;
    int32_t br = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t t = (int32_t)a.l[i] - (int32_t)P.p[i] + br;
        br = t >> 31;                                  // -1 on borrow
        d.l[i] = (i == 8) ? (uint32_t)t : ((uint32_t)t & F29_MASK);
    }
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = br ? a.l[i] : d.l[i];
    return r;
}

// normalised a < 48p  ->  a mod p (canonical), without a product: q = floor(a / p) is estimated from the top limb
// (a_8 / (p_8 + 1), scaled down by 2^-20 so that float rounding can only under-estimate: q_est is q or q - 1, the error terms
// being < 48 * 2^-20 + 48 / p_8 << 1), a - q_est * p lands in [0, 2p), one conditional subtraction finishes.  (Montgomery
// butterflies deliver < 21.4p after nine stages, the precomputed-quotient ones < 36p; a must stay below 2^261.)
// Replaces the "multiply by one" the NTT's last pass used to bring its lazy values (< 21.4 p after nine stages) below 1.36p:
// 9 mads + 9 carries + a canon instead of a 171-mad product + a canon.
FP_HD F29 f29_canon_lazy(const F29& a, const F29Params& P) {
    const uint32_t q = (uint32_t)((float)a.l[8] * P.inv_top);
    const int32_t nq = -(int32_t)q;
    int64_t acc = 0;
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc += (int64_t)a.l[i];
        acc += (int64_t)nq * (int64_t)(int32_t)P.p[i];
        r.l[i] = (i == 8) ? (uint32_t)acc : ((uint32_t)acc & F29_MASK);
        acc >>= 29;                                    // arithmetic: borrows travel as negative carries
    }
    return f29_canon(r, P);
}

// ---- host-side helpers (table construction)
// canonical integer limbs (8 x u32, < p) -> F29
inline F29 f29_from_canonical_host(const Fp<8>& c) { return f29_from_sat(c); }

inline F29Params f29_make_params(const FpParams<8>& P) {
    F29Params q;
    Fp<8> pm;
    for (int i = 0; i < 8; i++) pm.l[i] = P.p[i];
    F29 p29 = f29_from_sat(pm);
    for (int i = 0; i < 9; i++) q.p[i] = p29.l[i];
    // 2p, normalised, then lift: c[0] += 2^30 ; c[i] += 2^30 - 2 (0<i<8) ; c[8] -= 2
    uint32_t d[9];
    uint32_t carry = 0;
    for (int i = 0; i < 9; i++) {
        uint32_t v = 2 * q.p[i] + carry;
        carry = (i < 8) ? (v >> 29) : 0;
        d[i] = (i < 8) ? (v & F29_MASK) : v;
    }
    q.c2p[0] = d[0] + (1u << 30);
    for (int i = 1; i < 8; i++) q.c2p[i] = d[i] + (1u << 30) - 2;
    q.c2p[8] = d[8] - 2;
    carry = 0;
    for (int i = 0; i < 9; i++) {
        uint32_t v = 4 * q.p[i] + carry;
        carry = (i < 8) ? (v >> 29) : 0;
        d[i] = (i < 8) ? (v & F29_MASK) : v;
    }
    q.c4p[0] = d[0] + (1u << 30);
    for (int i = 1; i < 8; i++) q.c4p[i] = d[i] + (1u << 30) - 2;
    q.c4p[8] = d[8] - 2;
    // inv = -p^{-1} mod 2^29 (Newton)
    uint32_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - q.p[0] * x;
    q.inv = (0u - x) & F29_MASK;
    q.inv_top = (float)((1.0 - 1.0 / 1048576.0) / ((double)q.p[8] + 1.0));
    // one = 2^261 mod p : R256 mod p doubled five times (as residues)
    Fp<8> o;
    for (int i = 0; i < 8; i++) o.l[i] = P.one[i];
    for (int i = 0; i < 5; i++) o = fp_add(o, o, P);
    F29 o29 = f29_from_sat(o);
    for (int i = 0; i < 9; i++) q.one[i] = o29.l[i];
    // pbar = 2^261 - p on 29-bit limbs: (2^29 - p_0, 2^29 - 1 - p_1, ..., 2^29 - 1 - p_8)
    q.pbar[0] = (1u << 29) - q.p[0];
    for (int i = 1; i < 9; i++) q.pbar[i] = F29_MASK - q.p[i];
    uint32_t cy = 0;
    for (int i = 0; i < 9; i++) { const uint32_t v = q.pbar[i] + cy; q.pbar[i] = v & F29_MASK; cy = v >> 29; }
    return q;
}

// c given in the reference's Montgomery form (c*2^256) -> the prepared constant (plain residue and floor(c * 2^261 / p)); host only
inline F29S f29_shoup_from_mont256(const Fp<8>& c_mont, const FpParams<8>& P) {
    const Fp<8> c = fp_from_mont(c_mont, P);
    F29S s;
    const F29 c29 = f29_from_sat(c);
    for (int i = 0; i < 9; i++) s.c[i] = c29.l[i];
    // binary long division of c * 2^261 by p on 9 x 32-bit words
    uint32_t rem[9] = {0}, quo[9] = {0};
    for (int i = 0; i < 8; i++) rem[i] = c.l[i];
    for (int b = 260; b >= 0; b--) {
        uint32_t cy = 0;
        for (int i = 0; i < 9; i++) { const uint32_t v = rem[i]; rem[i] = (v << 1) | cy; cy = v >> 31; }
        bool ge = rem[8] != 0;
        if (!ge) {
            ge = true;
            for (int i = 7; i >= 0; i--) if (rem[i] != P.p[i]) { ge = rem[i] > P.p[i]; break; }
        }
        if (ge) {
            uint64_t br = 0;
            for (int i = 0; i < 9; i++) {
                const uint64_t t = (uint64_t)rem[i] - (i < 8 ? P.p[i] : 0) - br;
                rem[i] = (uint32_t)t;
                br = (t >> 32) & 1;
            }
            quo[b >> 5] |= 1u << (b & 31);
        }
    }
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, w = bit >> 5, off = bit & 31;
        uint64_t v = quo[w] >> off;
        if (off + 29 > 32 && w + 1 < 9) v |= (uint64_t)quo[w + 1] << (32 - off);
        s.cq[k] = (uint32_t)v & F29_MASK;
    }
    s.pad[0] = s.pad[1] = 0;
    return s;
}

// value v given in the reference's Montgomery form (v*2^256) -> constant form v*2^261 mod p, 29-bit limbs
inline F29 f29_const_from_mont256(const Fp<8>& v_mont, const FpParams<8>& P) {
    Fp<8> r261;                                   // 2^261 mod p as a plain residue
    for (int i = 0; i < 8; i++) r261.l[i] = P.one[i];
    for (int i = 0; i < 5; i++) r261 = fp_add(r261, r261, P);
    Fp<8> k_mont = fp_to_mont(r261, P);           // (2^261) * 2^256
    Fp<8> prod = fp_mul(v_mont, k_mont, P);       // v * 2^261 * 2^256
    return f29_from_sat(fp_from_mont(prod, P));   // v * 2^261 mod p, canonical
}
