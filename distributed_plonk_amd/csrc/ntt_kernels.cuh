// ntt_kernels.cuh — radix-2 NTT over Fr as LDS-tiled multi-pass kernels for gfx950.
//
// What it replaces (reference /root/reference/src): ark-poly Radix2EvaluationDomain::{fft,ifft}_in_place
// at worker.rs:82,84 (row NTT), :105,107 (column NTT), :398 and dispatcher.rs:594,632,667,
// dispatcher2.rs:507 (whole-vector NTT); the per-element `Fr::pow` twiddle / coset loops at
// worker.rs:75-80,91-93,109-114; and the transposes of dispatcher2.rs:754,786 / transpose.rs:413
// (folded into strided addressing).
//
// Decomposition (natural order in -> natural order out, X[k] = sum_j x[j] w^{jk}):
//   an array of size M = R_1*...*R_P is transformed by P passes.  With r_p = M/(R_1..R_p):
//   pass p<P : every contiguous sub-array of size r_{p-1} is viewed as an R_p x r_p matrix; each
//              column (stride r_p) gets a size-R_p NTT in LDS, is multiplied by w_{r_{p-1}}^{b*i}
//              and written back in place (row i).
//   pass P   : contiguous runs of R_P elements get a size-R_P NTT and are scattered to their
//              natural-order position  m + j*(M/R_P)  (m = mixed-radix digit reversal of the run index).
//   One workgroup owns a tile of T columns x R_p elements: HBM is touched in T*32-byte (coalesced
//   256-bit-limb) pieces, exactly once per pass; butterflies run out of LDS (limb-major SoA so that
//   consecutive lanes hit consecutive banks), 8 elements per lane held in VGPRs for up to three
//   radix-2 stages between LDS exchanges.  The in-LDS transform is decimation-in-frequency, so the
//   last stage carries no multiplication; the bit-reversed result order is undone for free in the
//   store addressing.
//
// Roofline: algorithmic bytes 2*32 B per element per transform; the kernels are
// VALU (v_mad_u64_u32) bound, not HBM bound — see DESIGN.md.
#pragma once
#include "fp.cuh"

typedef Fp<8> Fr;
typedef FpParams<8> FrParams;

#define NTT_LOG_RMAX 10   // largest in-LDS transform (2^10 elements)
#define NTT_MAX_PASSES 4

// exponent = idx * (bq*q + b0) + (aq*q + a0); value = lo[e & mask] * hi[e >> lt]
struct TwoLevelScale {
    const Fr* lo;       // base^e,        e < 2^lt
    const Fr* hi;       // base^(e<<lt),  e < 2^lt   (may be null when exponents are < 2^lt)
    uint64_t aq, a0, bq, b0;
    uint32_t lt;
    uint32_t enabled;
};

struct NttPassParams {
    const Fr* in;
    Fr* out;
    FrParams fp;
    const Fr* tw_small;       // w_Rmax^e, e < Rmax/2 (direction already chosen)
    const Fr* tw_lo;          // w_Nmax^e, e < 2^tw_lt      (inter-pass twiddles; may carry 1/N)
    const Fr* tw_hi;          // w_Nmax^(e << tw_lt)
    uint32_t tw_lt;
    uint32_t tw_shift;        // exponent = (b*i) << tw_shift   (log Nmax - log r_{p-1})
    uint32_t log_t;           // tile columns
    uint32_t tile_pitch;      // LDS row pitch (T or T+1)
    uint32_t load_a_fast;     // 1: consecutive lanes walk `a` on load (contiguous last pass)
    uint32_t is_last;
    // tile decode: ti -> x0 = ti % n0, x1 = (ti / n0) % n1, x2 = ti / (n0*n1)
    uint64_t n0, n1;
    uint64_t ls0, ls1, ls2;   // load base = x0*ls0 + x1*ls1 + x2*ls2
    uint64_t l_astride, l_tstride;
    uint64_t ss0, ss1, ss2;   // store base (non-last)
    uint64_t s_istride, s_tstride;
    // twiddle column b' = x0*bs0 + x1*bs1 + t*tb ; array q = x0*qs0 + x1*qs1 + x2*qs2 + t*tq
    uint64_t bs0, bs1, tb;
    uint64_t qs0, qs1, qs2, tq;
    // position of element (a,t) within its array for the prologue scale: pos = a*pa + x0*ps0 + x1*ps1 + t*pt
    uint64_t pa, ps0, ps1, pt;
    // last pass: k = m0 + t*tk + j*kstride ; m0 = x0*ms0 | digitrev(x1) ; out = q*oq + k*ok  (or split)
    uint64_t ms0, tk, kstride;
    uint32_t rev_ndig;                    // digits in x1 (top..bottom)
    uint32_t rev_w[NTT_MAX_PASSES];       // widths of those digits
    uint32_t rev_shift0;                  // shift of the first reversed digit in m
    uint64_t oq, ok;
    int32_t split_log;                    // >=0: out = (k>>split_log)*split_blk + q<<split_log + (k & mask)
    uint64_t split_blk;
    uint32_t scale_const_enabled;         // multiply outputs by `scale_const` (1/N when P==1)
    Fr scale_const;
    TwoLevelScale pro;                    // prologue (first pass) scale, idx = pos
    TwoLevelScale epi;                    // epilogue (last pass) scale, idx = k
};

__device__ __forceinline__ Fr load_fr(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void store_fr(Fr* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__device__ __forceinline__ Fr two_level(const TwoLevelScale& s, uint64_t idx, uint64_t q, const FrParams& fp) {
    uint64_t e = idx * (s.bq * q + s.b0) + (s.aq * q + s.a0);
    uint64_t mask = ((uint64_t)1 << s.lt) - 1;
    Fr v = load_fr(s.lo + (e & mask));
    uint64_t eh = e >> s.lt;
    if (s.hi != nullptr) {
        Fr h = load_fr(s.hi + (eh & mask));
        v = fp_mul(v, h, fp);
    }
    return v;
}

__device__ __forceinline__ uint32_t brev(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// LDS tile: limb-major.  word(l, idx) = l*plane + idx
struct LdsTile {
    uint32_t* base;
    uint32_t plane;
    __device__ __forceinline__ Fr get(uint32_t idx) const {
        Fr r;
#pragma unroll
        for (int l = 0; l < 8; l++) r.l[l] = base[l * plane + idx];
        return r;
    }
    __device__ __forceinline__ void put(uint32_t idx, const Fr& v) const {
#pragma unroll
        for (int l = 0; l < 8; l++) base[l * plane + idx] = v.l[l];
    }
};

// K radix-2 DIF stages (s0 .. s0+K-1 of a size-2^LOG_R transform) on the EPT elements a lane holds.
template <int LOG_R, int K, int EPT>
__device__ __forceinline__ void ntt_step(const LdsTile& tile, const Fr* tw_lds, int s0, uint32_t w, uint32_t t,
                                         uint32_t pitch, const FrParams& fp) {
    constexpr int R = 1 << LOG_R;
    constexpr int RADIX = 1 << K;
    constexpr int NG = EPT / RADIX;
    constexpr int GROUPS_PER_STRIDE = R / EPT;     // lanes along `w`
    const int logh = LOG_R - s0 - K;
    const uint32_t h = 1u << logh;
    const bool last_step = (logh == 0);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const uint32_t G = w + g * GROUPS_PER_STRIDE;
        const uint32_t lo = G & (h - 1), hi = G >> logh;
        const uint32_t a0 = (hi << (logh + K)) | lo;
        Fr v[RADIX];
#pragma unroll
        for (int k = 0; k < RADIX; k++) v[k] = tile.get((a0 + k * h) * pitch + t);
#pragma unroll
        for (int ds = 0; ds < K; ds++) {
            const int span = RADIX >> (ds + 1);
#pragma unroll
            for (int k = 0; k < RADIX; k++) {
                if (k & span) continue;
                const int kk = k & (span - 1);
                Fr x = v[k], y = v[k + span];
                v[k] = fp_add(x, y, fp);
                Fr d = fp_sub(x, y, fp);
                // twiddle exponent ((lo + kk*h) << (s0+ds)); it is 0 when lo==0 && kk==0
                if (last_step && kk == 0) {
                    v[k + span] = d;                      // w = 1 (lo == 0 in the last step)
                } else {
                    const uint32_t e = (lo + kk * h) << (s0 + ds);
                    Fr tw = load_fr(tw_lds + e);
                    v[k + span] = fp_mul(d, tw, fp);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RADIX; k++) tile.put((a0 + k * h) * pitch + t, v[k]);
    }
}

template <int LOG_R>
__global__ void __launch_bounds__(512) ntt_pass_kernel(const NttPassParams P) {
    constexpr int R = 1 << LOG_R;
    constexpr int EPT = (LOG_R >= 3) ? 8 : R;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t T = 1u << P.log_t;
    const uint32_t pitch = P.tile_pitch;
    const uint32_t plane = R * pitch;
    const uint32_t nthreads = (R * T) / EPT;
    const uint32_t u = threadIdx.x;
    LdsTile tile{smem, plane};
    Fr* tw_lds = reinterpret_cast<Fr*>(smem + 8 * plane);

    // ---- tile decode (scalar)
    const uint64_t ti = blockIdx.x;
    const uint64_t x0 = ti % P.n0, x12 = ti / P.n0, x1 = x12 % P.n1, x2 = x12 / P.n1;
    const uint64_t lbase = x0 * P.ls0 + x1 * P.ls1 + x2 * P.ls2;
    const uint64_t b0 = x0 * P.bs0 + x1 * P.bs1;
    const uint64_t q0 = x0 * P.qs0 + x1 * P.qs1 + x2 * P.qs2;

    // ---- small twiddle table -> LDS : tw_lds[e] = w_R^e , e < R/2
    for (uint32_t e = u; e < (R / 2 > 0 ? R / 2 : 1); e += nthreads)
        store_fr(tw_lds + e, load_fr(P.tw_small + ((uint64_t)e << (NTT_LOG_RMAX - LOG_R))));

    // ---- load tile
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const uint32_t e = u + i * nthreads;
        uint32_t a, t;
        if (P.load_a_fast) { a = e & (R - 1); t = e >> LOG_R; }
        else               { t = e & (T - 1); a = e >> P.log_t; }
        Fr v = load_fr(P.in + lbase + (uint64_t)a * P.l_astride + (uint64_t)t * P.l_tstride);
        if (P.pro.enabled) {
            const uint64_t pos = (uint64_t)a * P.pa + x0 * P.ps0 + x1 * P.ps1 + (uint64_t)t * P.pt;
            Fr s = two_level(P.pro, pos, q0 + (uint64_t)t * P.tq, P.fp);
            v = fp_mul(v, s, P.fp);
        }
        tile.put(a * pitch + t, v);
    }
    __syncthreads();

    // ---- in-LDS DIF transform, stages grouped (LOG_R % 3 first, then threes)
    {
        const uint32_t t = u & (T - 1), w = u >> P.log_t;
        constexpr int K0 = LOG_R % 3;
        int s = 0;
        if constexpr (LOG_R < 3) {
            ntt_step<LOG_R, LOG_R, EPT>(tile, tw_lds, 0, w, t, pitch, P.fp);
        } else {
            if constexpr (K0 != 0) {
                ntt_step<LOG_R, K0, EPT>(tile, tw_lds, 0, w, t, pitch, P.fp);
                s = K0;
                __syncthreads();
            }
#pragma unroll 1
            for (; s < LOG_R; s += 3) {
                ntt_step<LOG_R, 3, EPT>(tile, tw_lds, s, w, t, pitch, P.fp);
                __syncthreads();
            }
        }
        if constexpr (LOG_R < 3) __syncthreads();
    }

    // ---- store: LDS position a holds output index i = brev(a)
    uint64_t m0 = 0;
    if (P.is_last) {
        m0 = x0 * P.ms0;
        uint64_t rest = x1;
        // x1 holds rev_ndig digits, most significant first; digit d goes to shift rev_shift0 + sum of earlier widths
        uint32_t total = 0;
        for (uint32_t d = 0; d < P.rev_ndig; d++) total += P.rev_w[d];
        uint32_t sh = P.rev_shift0, consumed = 0;
        for (uint32_t d = 0; d < P.rev_ndig; d++) {
            consumed += P.rev_w[d];
            const uint64_t dig = (rest >> (total - consumed)) & (((uint64_t)1 << P.rev_w[d]) - 1);
            m0 |= dig << sh;
            sh += P.rev_w[d];
        }
    }
    const uint64_t sbase = x0 * P.ss0 + x1 * P.ss1 + x2 * P.ss2;
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const uint32_t e = u + i * nthreads;
        const uint32_t t = e & (T - 1), a = e >> P.log_t;
        const uint32_t idx = brev(a, LOG_R);
        Fr v = tile.get(a * pitch + t);
        if (!P.is_last) {
            const uint64_t b = b0 + (uint64_t)t * P.tb;
            const uint64_t ex = (b * idx) << P.tw_shift;
            const uint64_t mask = ((uint64_t)1 << P.tw_lt) - 1;
            Fr tw = load_fr(P.tw_lo + (ex & mask));
            const uint64_t eh = (ex >> P.tw_lt) & mask;
            Fr th = load_fr(P.tw_hi + eh);
            tw = fp_mul(tw, th, P.fp);
            v = fp_mul(v, tw, P.fp);
            store_fr(P.out + sbase + (uint64_t)idx * P.s_istride + (uint64_t)t * P.s_tstride, v);
        } else {
            const uint64_t q = q0 + (uint64_t)t * P.tq;
            const uint64_t k = m0 + (uint64_t)t * P.tk + (uint64_t)idx * P.kstride;
            if (P.scale_const_enabled) v = fp_mul(v, P.scale_const, P.fp);
            if (P.epi.enabled) {
                Fr s = two_level(P.epi, k, q, P.fp);
                v = fp_mul(v, s, P.fp);
            }
            uint64_t addr;
            if (P.split_log >= 0)
                addr = (k >> P.split_log) * P.split_blk + (q << P.split_log) + (k & (((uint64_t)1 << P.split_log) - 1));
            else
                addr = q * P.oq + k * P.ok;
            store_fr(P.out + addr, v);
        }
    }
}
