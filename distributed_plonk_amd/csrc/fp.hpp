// fp.hpp — Montgomery prime-field arithmetic on N x u32 little-endian limbs (gfx950 VALU).
//
// Semantics to match: ark-ff 0.3.0 Fp256/Fp384 as used by the reference's hot path
// (/root/reference/src/worker.rs:79,82-93,105-113,122,179): R = 2^(32N) (identical to ark's
// 2^(64*limbs64)), values stored as a*R mod p and ALWAYS fully reduced to [0,p), so limbs are
// bit-identical to the reference's in-memory / on-wire representation (utils.rs:27-43).
//
// No MFMA: this is carry-chain big-integer work.  The multiplier is v_mad_u64_u32 (32x32+64),
// everything else is v_add_co/v_addc_co/v_cndmask.  All four moduli leave the top bit of the top
// limb clear, which lets CIOS run with N+1 accumulator limbs ("no-carry" variant).
//
// The same code compiles for the host (table precomputation, final MSM fold) and the device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FP_HD __host__ __device__ __forceinline__
#else
#define FP_HD inline
#endif

template <int N>
struct alignas(16) Fp {
    uint32_t l[N];
};

template <int N>
struct FpParams {
    uint32_t p[N];     // modulus
    uint32_t one[N];   // R mod p
    uint32_t r2[N];    // R^2 mod p
    uint32_t inv;      // -p^{-1} mod 2^32
    int bits;          // MODULUS_BITS
};

template <int N> FP_HD Fp<N> fp_zero() {
    Fp<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = 0;
    return r;
}
template <int N> FP_HD Fp<N> fp_one(const FpParams<N>& P) {
    Fp<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = P.one[i];
    return r;
}
template <int N> FP_HD Fp<N> fp_from_limbs(const uint32_t* s) {
    Fp<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = s[i];
    return r;
}
template <int N> FP_HD bool fp_is_zero(const Fp<N>& a) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t |= a.l[i];
    return t == 0;
}
template <int N> FP_HD bool fp_eq(const Fp<N>& a, const Fp<N>& b) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < N; i++) t |= a.l[i] ^ b.l[i];
    return t == 0;
}

// r = a - p if a >= p else a     (a < 2p, given as N limbs + optional carry bit)
template <int N> FP_HD void fp_cond_sub_p(Fp<N>& a, uint32_t carry, const FpParams<N>& P) {
    uint32_t d[N];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)a.l[i] - P.p[i] - br;
        d[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    bool ge = carry || !br;
#pragma unroll
    for (int i = 0; i < N; i++) a.l[i] = ge ? d[i] : a.l[i];
}

template <int N> FP_HD Fp<N> fp_add(const Fp<N>& a, const Fp<N>& b, const FpParams<N>& P) {
    Fp<N> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    fp_cond_sub_p(r, (uint32_t)c, P);
    return r;
}

template <int N> FP_HD Fp<N> fp_sub(const Fp<N>& a, const Fp<N>& b, const FpParams<N>& P) {
    Fp<N> r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)a.l[i] - b.l[i] - br;
        r.l[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    // add p back if we borrowed
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (uint64_t)r.l[i] + (P.p[i] & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <int N> FP_HD Fp<N> fp_dbl(const Fp<N>& a, const FpParams<N>& P) { return fp_add(a, a, P); }

template <int N> FP_HD Fp<N> fp_neg(const Fp<N>& a, const FpParams<N>& P) {
    Fp<N> z = fp_zero<N>();
    return fp_sub(z, a, P);     // 0 - a; yields 0 for a == 0
}

// Montgomery product a*b*R^{-1} mod p, fully reduced.  CIOS, N+1 accumulator limbs.
#if defined(__HIPCC__) && !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
// HOST pass of the library only (hipcc; the g++ builds — tests/hostemu, tests/host_cpp — keep executing the device statement below): the same Montgomery
// product on 64-bit limbs.  The MSM's host fold (msm_engine.hip: c doublings and ~11 additions per window and vector, ~660 point operations per commitment) ran
// at 1.7 - 2.4 us per point operation on the 32-bit code — ~1.2 ms of host time per commitment at the END of every launch set, with the GPU idle behind it at every
// round boundary of a proof (13 commitments: 5 % of a 2^20 proof, 7 % of an 8-rank proof).  Same bits: 8 x 32 and 4 x 64 limbs hold the same integer and R = 2^(32 N).
#define FP_HOST64 1
template <int N> inline Fp<N> fp_mul_host64(const Fp<N>& a, const Fp<N>& b, const FpParams<N>& P) {
    constexpr int M = N / 2;
    typedef unsigned __int128 u128;
    uint64_t A[M], B[M], Q[M], t[M + 2];
    for (int i = 0; i < M; i++) {
        A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        B[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
        Q[i] = (uint64_t)P.p[2 * i] | ((uint64_t)P.p[2 * i + 1] << 32);
    }
    uint64_t x = (uint64_t)(uint32_t)(0u - P.inv);          // p^-1 mod 2^32 (P.inv = -p^-1 mod 2^32); one Newton step doubles the precision
    x *= 2 - Q[0] * x;
    const uint64_t inv = (uint64_t)0 - x;
    for (int i = 0; i < M + 2; i++) t[i] = 0;
    for (int i = 0; i < M; i++) {
        u128 c = 0;
        for (int j = 0; j < M; j++) { c += (u128)A[j] * B[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[M]; t[M] = (uint64_t)c; t[M + 1] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * inv;
        c = (u128)m * Q[0] + t[0];
        c >>= 64;
        for (int j = 1; j < M; j++) { c += (u128)m * Q[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[M]; t[M - 1] = (uint64_t)c; t[M] = t[M + 1] + (uint64_t)(c >> 64);
    }
    bool ge = t[M] != 0;
    if (!ge) {
        ge = true;
        for (int i = M - 1; i >= 0; i--) if (t[i] != Q[i]) { ge = t[i] > Q[i]; break; }
    }
    if (ge) {
        uint64_t br = 0;
        for (int i = 0; i < M; i++) { const u128 d = (u128)t[i] - Q[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    }
    Fp<N> r;
    for (int i = 0; i < M; i++) { r.l[2 * i] = (uint32_t)t[i]; r.l[2 * i + 1] = (uint32_t)(t[i] >> 32); }
    return r;
}
#endif

template <int N> FP_HD Fp<N> fp_mul(const Fp<N>& a, const Fp<N>& b, const FpParams<N>& P) {
#if defined(FP_HOST64)
    if constexpr (N % 2 == 0) return fp_mul_host64<N>(a, b, P);
#endif
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < N; j++) {
            c = (uint64_t)a.l[j] * bi + t[j] + c;
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        uint32_t tn = t[N] + (uint32_t)c;          // cannot overflow: top bit of p is clear
        const uint32_t m = t[0] * P.inv;
        c = (uint64_t)m * P.p[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c = (uint64_t)m * P.p[j] + t[j] + c;
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += tn;
        t[N - 1] = (uint32_t)c;
        t[N] = (uint32_t)(c >> 32);
    }
    Fp<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    fp_cond_sub_p(r, t[N], P);
    return r;
}

template <int N> FP_HD Fp<N> fp_sqr(const Fp<N>& a, const FpParams<N>& P) { return fp_mul(a, a, P); }

// into_repr(): Montgomery -> canonical
template <int N> FP_HD Fp<N> fp_from_mont(const Fp<N>& a, const FpParams<N>& P) {
    Fp<N> o = fp_zero<N>();
    o.l[0] = 1;
    return fp_mul(a, o, P);
}
// from_repr(): canonical -> Montgomery
template <int N> FP_HD Fp<N> fp_to_mont(const Fp<N>& a, const FpParams<N>& P) {
    Fp<N> r2;
#pragma unroll
    for (int i = 0; i < N; i++) r2.l[i] = P.r2[i];
    return fp_mul(a, r2, P);
}

template <int N> FP_HD Fp<N> fp_pow_u64(const Fp<N>& a, uint64_t e, const FpParams<N>& P) {
    Fp<N> acc = fp_one(P), b = a;
    while (e) {
        if (e & 1) acc = fp_mul(acc, b, P);
        b = fp_sqr(b, P);
        e >>= 1;
    }
    return acc;
}

// a^(p-2) (0 -> 0).  Not unrolled: used off the hot loop only (affine normalisation, table setup).
template <int N> FP_HD Fp<N> fp_inv(const Fp<N>& a, const FpParams<N>& P) {
    uint32_t e[N];
    uint64_t br = 2;
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)P.p[i] - br;
        e[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    Fp<N> acc = fp_one(P);
    for (int i = 32 * N - 1; i >= 0; i--) {
        acc = fp_sqr(acc, P);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fp_mul(acc, a, P);
    }
    return acc;
}
