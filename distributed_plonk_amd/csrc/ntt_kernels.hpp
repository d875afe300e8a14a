// ntt_kernels.hpp — radix-2 NTT over Fr as LDS-tiled multi-pass kernels for gfx950.
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
//   256-bit-limb) pieces, exactly once per pass, in the reference's 8x32-bit Montgomery layout.
//   Inside the kernel elements live as 9 x 29-bit limbs (fp29.hpp): butterflies run out of LDS
//   (limb-major SoA: consecutive lanes hit consecutive banks), 8 elements per lane are held in
//   VGPRs for up to three decimation-in-time stages between LDS exchanges; reduction is lazy
//   (bounds grow by 2p per stage), only the final store canonicalises.  The bit-reversed input
//   order DIT needs is produced for free by the load addressing; its first stage has no products.
//
// Roofline: algorithmic bytes 2*32 B per element per transform; the kernels are
// VALU (v_mad_u64_u32) bound, not HBM bound — see DESIGN.md.
#pragma once
#include "fp.hpp"
#include "fp29.hpp"

typedef Fp<8> Fr;
typedef FpParams<8> FrParams;

#define NTT_LOG_RMAX 9    // largest in-LDS transform (2^9 elements: 4096-element tile x 36 B = 144 KiB)
#define NTT_MAX_PASSES 4
// coefficients beyond the size M of a shared-input evaluation fold back onto it: at most this many pieces of M (8: the class-local size-n iNTT of
// an 8-rank prover reads n evaluations as the coefficients of an (n/8)-point evaluation, class_prover.py)
#define NTT_MAX_FOLD 8

// exponent = idx * (bq*q + b0) + (aq*q + a0); value = lo[e & mask] * hi[e >> lt]   (constants: c*2^261 mod p)
struct TwoLevelScale {
    const F29* lo;
    const F29* hi;
    uint64_t aq, a0, bq, b0;
    uint32_t lt;
    uint32_t enabled;
};

struct NttPassParams {
    const Fr* in;
    Fr* out;
    F29Params fp;
    const F29* tw_small;      // w_Rmax^e, e < Rmax/2 (direction already chosen)
    const F29S* tw_shoup;     // the same twiddles prepared for f29_mul_shoup (plain residue + precomputed quotient), read through L1
    const F29* tw_lo;         // w_Nmax^e, e < 2^tw_lt      (inter-pass twiddles; may carry 1/N)
    const F29* tw_hi;         // w_Nmax^(e << tw_lt)
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
    const Fr* tw_plane;                   // non-last pass: precomputed inter-pass factors, index i*r_p + b (nullable)
    uint64_t plane_rp;                    // r_p (row pitch of the plane)
    const F29* pro_rowtab;                // first pass: per-row input scale G[a] (coset shift g^(a*r_1)), nullable
    uint64_t rowtab_qstride;              // entries between the row tables of consecutive arrays (0: one table for all)
    uint64_t plane_qstride;               // elements between the planes of consecutive arrays (0: one plane for all)
    // Shared zero-padded input (class-decomposed coset evaluation, ntt_engine.hip: NttCall::shared_in): every array q reads the SAME
    // coefficient vector — element `pos` is in[pos] for pos < in_len, else 0 — and coefficients beyond the array size M fold back
    // with the constants of the array's own coset:  + sum_{u=1}^{nfold-1} fold_c[q][u] * in[pos + u*M]   (X^M = h_q^M on coset q).
    uint64_t in_len;                      // 0: dense per-array input
    uint64_t fold_m;
    uint32_t nfold;
    const F29* fold_c;                    // [class][NTT_MAX_FOLD], constant form
    // Arrays = rows x classes, array A = row << cls_log | class: the classes of a row read that row's coefficients
    // (in + row * in_row_pitch) and share the per-class tables; in the last pass the classes of a row interleave into that row's
    // natural order: output index k' = k << cls_log | class, output array = row.  cls_log = 0: every array is its own row (the
    // dense transforms), the formulas reduce to the identity.
    uint32_t cls_log;
    uint64_t in_row_pitch;
    const Fr* epi_plane;                  // last pass: precomputed output factors, index q*epi_qstride + k (nullable)
    uint64_t epi_qstride;
    uint32_t scale_const_enabled;         // multiply outputs by `scale_const` (1/N when P==1)
    F29 scale_const;
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
__device__ __forceinline__ F29 load_f29(const F29* p) {
    F29 r;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = q[i];
    return r;
}
__device__ __forceinline__ F29 params_one(const F29Params& P) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = P.one[i];
    return r;
}

__device__ __forceinline__ F29 two_level(const TwoLevelScale& s, uint64_t idx, uint64_t q, const F29Params& fp) {
    const uint64_t e = idx * (s.bq * q + s.b0) + (s.aq * q + s.a0);
    const uint64_t mask = ((uint64_t)1 << s.lt) - 1;
    F29 v = load_f29(s.lo + (e & mask));
    F29 h = load_f29(s.hi + ((e >> s.lt) & mask));
    return f29_mul(v, h, fp);
}

__device__ __forceinline__ uint32_t brev(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// LDS tile: limb-major.  word(l, idx) = l*plane + idx
struct LdsTile {
    uint32_t* base;
    uint32_t plane;
    __device__ __forceinline__ F29 get(uint32_t idx) const {
        F29 r;
#pragma unroll
        for (int l = 0; l < 9; l++) r.l[l] = base[l * plane + idx];
        return r;
    }
    __device__ __forceinline__ void put(uint32_t idx, const F29& v) const {
#pragma unroll
        for (int l = 0; l < 9; l++) base[l * plane + idx] = v.l[l];
    }
};

// Row swizzle for 8-column tiles: word(l, row, t) sits in bank (row % 4) * 8 + t, so a wave whose 8 rows differ only ABOVE bit 1 —
// the bit-reversed rows of the load phase, the stride-4 rows of the first step — hits 8 banks 8 lanes deep (PMC, round 1: 58 % of
// the LDS-active cycles of the pass kernels were bank-conflict cycles).  XOR-folding all higher 2-bit groups of the row into its
// low two bits spreads every access pattern of the kernel over all 32 banks (2 lanes per bank: the floor for 64 lanes).  The
// map is an involution on rows and XOR-linear, so row a0 ^ (k << s0) costs one fold of a0 per step plus scalar constants.
template <bool SWZ> __device__ __forceinline__ uint32_t sw_fold(uint32_t row) {
    if (!SWZ) return 0;
    uint32_t y = row >> 2;
    y ^= y >> 4;
    y ^= y >> 2;
    return y & 3;
}

// K radix-2 decimation-in-time stages (s0 .. s0+K-1 of a size-2^LOG_R transform whose input sits in
// bit-reversed order) on the EPT elements a lane holds.  Tile contents are normalised on entry and exit.
// prepared twiddle e of the size-2^LOG_R transform (table entry e << (9 - LOG_R)): five 128-bit loads
template <int LOG_R>
__device__ __forceinline__ void shoup_tw_load(const F29S* __restrict__ tab, uint32_t e, uint32_t* cw) {
    const uint4* src = reinterpret_cast<const uint4*>(tab + ((size_t)e << (NTT_LOG_RMAX - LOG_R)));
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint4 d = src[j];
        cw[4 * j] = d.x; cw[4 * j + 1] = d.y; cw[4 * j + 2] = d.z; cw[4 * j + 3] = d.w;
    }
}
// The K radix-2 stages s0 .. s0+K-1 on the 2^K elements of one group (v[k] = row a0 + k * 2^s0, lo = the group's low row bits).
template <int LOG_R, int K, bool FIRST, bool SHOUP>
__device__ __forceinline__ void ntt_butterflies(F29 (&v)[1 << K], int s0, uint32_t lo, const uint32_t* tw_lds, const F29S* __restrict__ tw_shoup,
                                                const F29Params& fp) {
    constexpr int R = 1 << LOG_R;
    constexpr int RADIX = 1 << K;
    constexpr int TWN = (R / 2 > 0) ? R / 2 : 1;
    const uint32_t h = 1u << s0;
#pragma unroll
    for (int ds = 0; ds < K; ds++) {
        const int span = 1 << ds;
#pragma unroll
        for (int k = 0; k < RADIX; k++) {
            if (k & span) continue;
            const int kk = k & (span - 1);
            F29 tt;
            const F29 x = v[k];
            if (FIRST && kk == 0 && ds <= 1) {
                // first step (s0 = 0, lo = 0): exponent kk << ... is 0, the twiddle is 1 — no product.
                // ds == 0: the partner is a fresh input (< 1.4p, normalised): x + 2p - y.
                // ds == 1: the partner is the lazy sum of two inputs (< 2.8p, limbs <= 2^30 - 2), so the
                //          offset must be 4p or the VALUE can go negative (caught by tests/test_gpu_fullsize.py).
                tt = v[k + span];
                v[k] = f29_add(x, tt);
                v[k + span] = (ds == 0) ? f29_sub2p(x, tt, fp) : f29_sub4p(x, tt, fp);
            } else {
                const uint32_t e = (lo + kk * h) << (LOG_R - s0 - ds - 1);
                if constexpr (SHOUP) {
                    uint32_t cw[20];
                    shoup_tw_load<LOG_R>(tw_shoup, e, cw);
                    tt = f29_mul_shoup(v[k + span], cw, cw + 9, fp);
                    v[k] = f29_add(x, tt);
                    v[k + span] = f29_sub4p(x, tt, fp);
                } else {
                    F29 tw;
#pragma unroll
                    for (int l = 0; l < 9; l++) tw.l[l] = tw_lds[l * TWN + e];
                    tt = f29_mul(v[k + span], tw, fp);
                    v[k] = f29_add(x, tt);
                    v[k + span] = f29_sub2p(x, tt, fp);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one butterfly's temporaries live at a time (interleaving two spills at 128 VGPRs: measured 1.6x slower)
        }
        if (ds == 1 || ds == K - 1) {
#pragma unroll
            for (int k = 0; k < RADIX; k++) f29_norm(v[k]);
        }
    }
}

// CONTRACT of the SHOUP path.  f29_mul_shoup needs x < 2^261 (its quotient estimate is then at most 2 short: result < 3p) and limbs < 2^31 (its
// second loop holds 9 * 2^31 * 2^29 + 9 * 2^58 + carry < 2^63.6 in the 64-bit accumulator).  Limbs: an operand of a butterfly product has seen
// AT MOST ONE un-normalised f29_sub4p / f29_add since the last f29_norm — ntt_butterflies normalises after ds == 1 and after the last stage of a
// step — so its limbs stay below 2^29 + (2^30 + 2^29) = 2^31.  Values: tile values enter a pass below L = 1.36p (first pass: a Montgomery product
// of canonical data) or L = 1.6p (later passes: the previous pass's plane product of a value below 38p returns < 38p * p / 2^261 + p); the two
// product-free stages take them to < 2L + 4p + ... < 7.6p, every further stage adds 4p (x + 4p - t, t < 3p): < 7.6p + 4p * 7 = 35.6p after the
// nine stages of the widest pass.  2^261 = 140p on BN254 and 70p on BLS12-381: both fields qualify (rounds 1-3 excluded BLS12-381 because
// f29_mul's DOCUMENTED range is 2^259.4 = 23p there; its real requirement is the same x < 2^261, limbs < 2^31, with a result below
// x * p / 2^261 + p < 2p instead of 1.36p — still below 2^256 for the 8 x 32-bit store and below f29_canon's 2p).  The last pass ends in
// f29_canon_lazy (< 48p, below 2^261 on both fields).  Host build of this code against Python integers over these ranges: tests/test_fp29_host.py.
// SHOUP: the butterfly products use the precomputed-quotient multiplier with twiddles fetched from the 80-byte global table (five
// 128-bit loads per product on the otherwise idle vector-memory pipe; the table is L1-resident) instead of Montgomery products
// with twiddles from LDS: 143 limb products instead of 171 + 9.
template <int LOG_R, int K, int EPT, bool FIRST, bool SWZ, bool SHOUP>
__device__ __forceinline__ void ntt_step(const LdsTile& tile, const uint32_t* tw_lds, const F29S* __restrict__ tw_shoup, int s0, uint32_t w, uint32_t t,
                                         uint32_t pitch, const F29Params& fp) {
    constexpr int R = 1 << LOG_R;
    constexpr int RADIX = 1 << K;
    constexpr int NG = EPT / RADIX;
    constexpr int GROUPS_PER_STRIDE = R / EPT;     // lanes along `w`
    const uint32_t h = 1u << s0;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const uint32_t G = w + g * GROUPS_PER_STRIDE;
        const uint32_t lo = G & (h - 1), hi = G >> s0;
        const uint32_t a0 = (hi << (s0 + K)) | lo;
        F29 v[RADIX];
        const uint32_t fa = sw_fold<SWZ>(a0);
#pragma unroll
        for (int k = 0; k < RADIX; k++) v[k] = tile.get(((a0 + k * h) ^ fa ^ sw_fold<SWZ>(k * h)) * pitch + t);
        ntt_butterflies<LOG_R, K, FIRST, SHOUP>(v, s0, lo, tw_lds, tw_shoup, fp);
#pragma unroll
        for (int k = 0; k < RADIX; k++) tile.put(((a0 + k * h) ^ fa ^ sw_fold<SWZ>(k * h)) * pitch + t, v[k]);
    }
}

#ifdef NTT_XLANE
// EXPERIMENT (north_star: "wavefront-shuffle reductions"; built only with -DNTT_XLANE, profiles/r03_ntt_xlane_experiment.txt): stages
// 0..3 of the in-LDS transform WITHOUT the LDS round trip between the first two steps.  After stages 0-1 lane w holds rows 4w + k;
// stages 2-3 want rows (w >> 2) * 16 + (w & 3) + 4k: a 4 x 4 transpose between the register index k and the two low bits of w.
// With those two bits placed on lane bits 4 and 5 (the lane <-> w assignment is free as long as it is used consistently) the
// transpose is two rounds of gfx950's v_permlane16_swap / v_permlane32_swap — one VALU instruction per limb and register pair, no
// select, no LDS traffic, no barrier: 36 instructions per lane instead of 36 ds_write + 36 ds_read + s_barrier.
template <int LOG_R, bool SWZ, bool SHOUP>
__device__ __forceinline__ void ntt_steps0123_xlane(const LdsTile& tile, const uint32_t* tw_lds, const F29S* __restrict__ tw_shoup, uint32_t w, uint32_t t,
                                                    uint32_t pitch, const F29Params& fp) {
    // w = u >> 3: its bits 0, 1, 2 are lane bits 3, 4, 5.  w_eff: bit 0 <- lane bit 4, bit 1 <- lane bit 5, bit 2 <- lane bit 3.
    const uint32_t we = (w & ~7u) | ((w >> 1) & 3u) | ((w & 1u) << 2);
    F29 v[4];
    {
        const uint32_t a0 = we << 2, fa = sw_fold<SWZ>(a0);
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = tile.get(((a0 + k) ^ fa ^ sw_fold<SWZ>(k)) * pitch + t);
    }
    ntt_butterflies<LOG_R, 2, true, SHOUP>(v, 0, 0, tw_lds, tw_shoup, fp);
#pragma unroll
    for (int l = 0; l < 9; l++) {                                       // register bit 0 <-> lane bit 4
        auto r01 = __builtin_amdgcn_permlane16_swap(v[0].l[l], v[1].l[l], false, false);
        auto r23 = __builtin_amdgcn_permlane16_swap(v[2].l[l], v[3].l[l], false, false);
        v[0].l[l] = r01[0]; v[1].l[l] = r01[1]; v[2].l[l] = r23[0]; v[3].l[l] = r23[1];
    }
#pragma unroll
    for (int l = 0; l < 9; l++) {                                       // register bit 1 <-> lane bit 5
        auto r02 = __builtin_amdgcn_permlane32_swap(v[0].l[l], v[2].l[l], false, false);
        auto r13 = __builtin_amdgcn_permlane32_swap(v[1].l[l], v[3].l[l], false, false);
        v[0].l[l] = r02[0]; v[2].l[l] = r02[1]; v[1].l[l] = r13[0]; v[3].l[l] = r13[1];
    }
    const uint32_t lo = we & 3u, hi = we >> 2;
    ntt_butterflies<LOG_R, 2, false, SHOUP>(v, 2, lo, tw_lds, tw_shoup, fp);
    const uint32_t a0 = (hi << 4) | lo, fa = sw_fold<SWZ>(a0);
#pragma unroll
    for (int k = 0; k < 4; k++) tile.put(((a0 + k * 4) ^ fa ^ sw_fold<SWZ>(k * 4)) * pitch + t, v[k]);
}
#endif

template <int LOG_R, int EPT_REQ, bool SWZ = false, bool SHOUP = false>
__global__ void __launch_bounds__(EPT_REQ <= 4 ? 1024 : 512) ntt_pass_kernel(const NttPassParams P) {
    constexpr int R = 1 << LOG_R;
    constexpr int EPT = (R >= EPT_REQ) ? EPT_REQ : R;
    constexpr int KMAX = (EPT == 8) ? 3 : (EPT == 4) ? 2 : (EPT == 2 ? 1 : 1);
    constexpr int TWN = (R / 2 > 0) ? R / 2 : 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // The swizzled instantiation is only dispatched for 8-column tiles with pitch 8 (ntt_engine.hip: launch_one): with the tile shape a
    // compile-time constant the column t = e & 7 of a lane is the same for its EPT elements (the lane count is a multiple of 8), and
    // everything that depends on t alone — array / class / plane / row-table / output base addresses, 64-bit products — is computed
    // once per lane instead of once per element.
    const uint32_t LOG_T = SWZ ? 3u : P.log_t;
    const uint32_t T = 1u << LOG_T;
    const uint32_t pitch = SWZ ? 8u : P.tile_pitch;
    const uint32_t plane = R * pitch;
    const uint32_t nthreads = (R * T) / EPT;
    const uint32_t u = threadIdx.x;
    LdsTile tile{smem, plane};
    uint32_t* tw_lds = smem + 9 * plane;            // limb-major: word(l, e) = l*TWN + e

    // ---- tile decode (scalar)
    const uint64_t ti = blockIdx.x;
    const uint64_t x0 = ti % P.n0, x12 = ti / P.n0, x1 = x12 % P.n1, x2 = x12 / P.n1;
    const uint64_t lbase = x0 * P.ls0 + x1 * P.ls1 + x2 * P.ls2;
    const uint64_t b0 = x0 * P.bs0 + x1 * P.bs1;
    const uint64_t q0 = x0 * P.qs0 + x1 * P.qs1 + x2 * P.qs2;

    // ---- small twiddle table -> LDS : w_R^e , e < R/2   (the SHOUP instantiation reads its prepared twiddles through L1 instead)
    if constexpr (!SHOUP) {
        for (uint32_t e = u; e < TWN; e += nthreads) {
            const F29 tw = load_f29(P.tw_small + ((uint64_t)e << (NTT_LOG_RMAX - LOG_R)));
#pragma unroll
            for (int l = 0; l < 9; l++) tw_lds[l * TWN + e] = tw.l[l];
        }
    }

    // ---- load tile (element a of the column goes to LDS row brev(a): DIT input order)
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const uint32_t e = u + i * nthreads;
        uint32_t a, t;
        if (P.load_a_fast) { a = e & (R - 1); t = e >> LOG_R; }
        else               { t = e & (T - 1); a = e >> LOG_T; }
        F29 v;
        const uint64_t arr = q0 + (uint64_t)t * P.tq;                       // array index; its class selects the first-pass tables
        const uint64_t cls = arr & (((uint64_t)1 << P.cls_log) - 1);
        if (P.in_len != 0) {
            const uint64_t pos = (uint64_t)a * P.pa + x0 * P.ps0 + x1 * P.ps1 + (uint64_t)t * P.pt;
            const Fr* src = P.in + (arr >> P.cls_log) * P.in_row_pitch;
            if (pos < P.in_len) {
                v = f29_from_sat(load_fr(src + pos));
            } else {
#pragma unroll
                for (int l = 0; l < 9; l++) v.l[l] = 0;
            }
            if (P.nfold > 1 && pos + P.fold_m < P.in_len) {       // a handful of elements per transform (n + 3 coefficients on n points)
                const F29* fc = P.fold_c + NTT_MAX_FOLD * cls;
                for (uint32_t uu = 1; uu < P.nfold; uu++) {
                    const uint64_t j = pos + (uint64_t)uu * P.fold_m;
                    if (j < P.in_len) v = f29_add(v, f29_mul(f29_from_sat(load_fr(src + j)), load_f29(fc + uu), P.fp));
                    if (uu == 3) f29_norm(v);                     // limbs: at most four normalised values summed between normalisations
                }
                f29_norm(v);                                      // < p + 7 * 1.36 p = 10.6 p: inside the row product's operand range (2^261 > 70 p)
            }
        } else {
            v = f29_from_sat(load_fr(P.in + lbase + (uint64_t)a * P.l_astride + (uint64_t)t * P.l_tstride));
        }
        if (P.pro_rowtab != nullptr) {
            v = f29_mul(v, load_f29(P.pro_rowtab + cls * P.rowtab_qstride + a), P.fp);
        } else if (P.pro.enabled) {
            const uint64_t pos = (uint64_t)a * P.pa + x0 * P.ps0 + x1 * P.ps1 + (uint64_t)t * P.pt;
            const F29 s = two_level(P.pro, pos, arr, P.fp);
            v = f29_mul(v, s, P.fp);
        }
        const uint32_t row = brev(a, LOG_R);
        tile.put((row ^ sw_fold<SWZ>(row)) * pitch + t, v);
    }
    __syncthreads();

    // ---- in-LDS DIT transform, stages grouped KMAX at a time from stage 0 (the first group enjoys the
    //      trivial twiddles), the LOG_R % KMAX remainder last
    {
        const uint32_t t = u & (T - 1), w = u >> LOG_T;
        constexpr int KREM = LOG_R % KMAX;
        constexpr int SFULL = LOG_R - KREM;          // stages covered by full groups
        if constexpr (R <= EPT) {
            ntt_step<LOG_R, LOG_R, EPT, true, SWZ, SHOUP>(tile, tw_lds, P.tw_shoup, 0, w, t, pitch, P.fp);
            __syncthreads();
        } else {
            if constexpr (SFULL > 0) {
#ifdef NTT_XLANE
                constexpr bool XL = SWZ && EPT == 4 && LOG_R >= 7;       // one group per lane, 8-column tiles: w = u >> 3
#else
                constexpr bool XL = false;
#endif
                if constexpr (XL) {
#ifdef NTT_XLANE
                    ntt_steps0123_xlane<LOG_R, SWZ, SHOUP>(tile, tw_lds, P.tw_shoup, w, t, pitch, P.fp);
#endif
                } else {
                    ntt_step<LOG_R, KMAX, EPT, true, SWZ, SHOUP>(tile, tw_lds, P.tw_shoup, 0, w, t, pitch, P.fp);
                }
                __syncthreads();
#pragma unroll 1
                for (int s = XL ? 2 * KMAX : KMAX; s < SFULL; s += KMAX) {
                    ntt_step<LOG_R, KMAX, EPT, false, SWZ, SHOUP>(tile, tw_lds, P.tw_shoup, s, w, t, pitch, P.fp);
                    __syncthreads();
                }
            }
            if constexpr (KREM != 0) {
                ntt_step<LOG_R, KREM, EPT, SFULL == 0, SWZ, SHOUP>(tile, tw_lds, P.tw_shoup, SFULL, w, t, pitch, P.fp);
                __syncthreads();
            }
        }
    }

    // ---- store: LDS row i holds output index i
    uint64_t m0 = 0;
    if (P.is_last) {
        m0 = x0 * P.ms0;
        uint32_t total = 0;
        for (uint32_t d = 0; d < P.rev_ndig; d++) total += P.rev_w[d];
        uint32_t sh = P.rev_shift0, consumed = 0;
        for (uint32_t d = 0; d < P.rev_ndig; d++) {      // x1 holds rev_ndig digits, most significant first
            consumed += P.rev_w[d];
            const uint64_t dig = (x1 >> (total - consumed)) & (((uint64_t)1 << P.rev_w[d]) - 1);
            m0 |= dig << sh;
            sh += P.rev_w[d];
        }
    }
    const uint64_t sbase = x0 * P.ss0 + x1 * P.ss1 + x2 * P.ss2;
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const uint32_t e = u + i * nthreads;
        const uint32_t t = e & (T - 1), idx = e >> LOG_T;
        F29 v = tile.get((idx ^ sw_fold<SWZ>(idx)) * pitch + t);
        if (!P.is_last) {
            const uint64_t b = b0 + (uint64_t)t * P.tb;
            if (P.tw_plane != nullptr) {
                // streamed factor plane: one 256-bit-limb load replaces two table gathers and a product
                const uint64_t pcls = (q0 + (uint64_t)t * P.tq) & (((uint64_t)1 << P.cls_log) - 1);
                v = f29_mul(v, f29_from_sat(load_fr(P.tw_plane + pcls * P.plane_qstride + (uint64_t)idx * P.plane_rp + b)), P.fp);
            } else {
                const uint64_t ex = (b * idx) << P.tw_shift;
                const uint64_t mask = ((uint64_t)1 << P.tw_lt) - 1;
                const F29 tl = load_f29(P.tw_lo + (ex & mask));
                const F29 th = load_f29(P.tw_hi + ((ex >> P.tw_lt) & mask));
                v = f29_mul(v, f29_mul(tl, th, P.fp), P.fp);
            }
            // < 1.36 p < 2^256: stored re-packed but not canonicalised (the next pass does not care)
            store_fr(P.out + sbase + (uint64_t)idx * P.s_istride + (uint64_t)t * P.s_tstride, f29_to_sat(v));
        } else {
            const uint64_t arr = q0 + (uint64_t)t * P.tq;
            const uint64_t q = arr >> P.cls_log;                                        // output array (row)
            const uint64_t k = ((m0 + (uint64_t)t * P.tk + (uint64_t)idx * P.kstride) << P.cls_log) | (arr & (((uint64_t)1 << P.cls_log) - 1));
            bool lazy = false;
            if (P.epi_plane != nullptr) {
                v = f29_mul(v, f29_from_sat(load_fr(P.epi_plane + q * P.epi_qstride + k)), P.fp);
                if (P.scale_const_enabled) v = f29_mul(v, P.scale_const, P.fp);
            } else if (P.epi.enabled) {
                v = f29_mul(v, two_level(P.epi, k, q, P.fp), P.fp);
                if (P.scale_const_enabled) v = f29_mul(v, P.scale_const, P.fp);
            } else if (P.scale_const_enabled) {
                v = f29_mul(v, P.scale_const, P.fp);
            } else {
                lazy = true;                                     // < 7.4p + 2p per stage beyond the second: < 21.4p after nine (SHOUP: 4p per stage, < 36p)
            }
            v = lazy ? f29_canon_lazy(v, P.fp) : f29_canon(v, P.fp);
            uint64_t addr;
            if (P.split_log >= 0)
                addr = (k >> P.split_log) * P.split_blk + (q << P.split_log) + (k & (((uint64_t)1 << P.split_log) - 1));
            else
                addr = q * P.oq + k * P.ok;
            store_fr(P.out + addr, f29_to_sat(v));
        }
    }
}


// Inter-pass factor plane for pass p of a size-M transform: plane[i*r_p + b] = w_{r_{p-1}}^{b*i} (times 1/M for the
// first pass of an inverse transform via the scaled lo table, times g^b when a forward coset shift is folded in),
// stored canonical in constant form (c*2^261 mod p) as 8 x u32.
static __global__ void __launch_bounds__(256) ntt_gen_plane_kernel(Fr* __restrict__ out, uint64_t r_prev, uint64_t r_p, const F29* __restrict__ tw_lo,
                                                            const F29* __restrict__ tw_hi, uint32_t lt, uint32_t shift,
                                                            const F29* __restrict__ g_lo, const F29* __restrict__ g_hi, uint64_t coset_mult,
                                                            const F29Params fp) {
    const uint64_t pos = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= r_prev) return;
    const uint64_t i = pos / r_p, b = pos % r_p;
    const uint64_t ex = (b * i) << shift, mask = ((uint64_t)1 << lt) - 1;
    F29 v = f29_mul(load_f29(tw_lo + (ex & mask)), load_f29(tw_hi + ((ex >> lt) & mask)), fp);
    if (g_lo != nullptr) {
        const uint64_t eg = b * coset_mult;           // g^(b * coset_mult): the column half of the coset shift
        const F29 g = f29_mul(load_f29(g_lo + (eg & mask)), load_f29(g_hi + ((eg >> lt) & mask)), fp);
        v = f29_mul(v, g, fp);
    }
    store_fr(out + pos, f29_to_sat(f29_canon(v, fp)));
}


// First-pass plane of a transform on the coset h * <w_M>:  plane[i*r_p + b] = w_M^(b*i) * h^b, with h^b = h_lo[b & 1023] * h_hi[b >> 10]
// (host-built power tables of the arbitrary shift h).  Same storage form as ntt_gen_plane_kernel.
static __global__ void __launch_bounds__(256) ntt_gen_shift_plane_kernel(Fr* __restrict__ out, uint64_t r_prev, uint64_t r_p, const F29* __restrict__ tw_lo,
                                                                  const F29* __restrict__ tw_hi, uint32_t lt, uint32_t shift,
                                                                  const F29* __restrict__ h_lo, const F29* __restrict__ h_hi, const F29Params fp) {
    const uint64_t pos = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= r_prev) return;
    const uint64_t i = pos / r_p, b = pos % r_p;
    const uint64_t ex = (b * i) << shift, mask = ((uint64_t)1 << lt) - 1;
    F29 v = f29_mul(load_f29(tw_lo + (ex & mask)), load_f29(tw_hi + ((ex >> lt) & mask)), fp);
    v = f29_mul(v, load_f29(h_lo + (b & 1023)), fp);
    if (r_p > 1024) v = f29_mul(v, load_f29(h_hi + (b >> 10)), fp);
    store_fr(out + pos, f29_to_sat(f29_canon(v, fp)));
}


// Output-factor plane of the distributed row pass (fft1_helper, worker.rs:86-93): plane[q*M + k] =
// w_N^(+-(q+q0)*k) (direction in the tables), times g^(q+q0) when the forward coset shift's row constant is folded in.
static __global__ void __launch_bounds__(256) ntt_gen_epi_plane_kernel(Fr* __restrict__ out, uint64_t M, uint64_t batch, uint64_t q0,
                                                                const F29* __restrict__ tw_lo, const F29* __restrict__ tw_hi, uint32_t lt,
                                                                uint32_t shift, const F29* __restrict__ g_lo, const F29* __restrict__ g_hi,
                                                                const F29Params fp) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= M * batch) return;
    const uint64_t q = id / M + q0, kk = id % M;
    const uint64_t ex = (q * kk) << shift, mask = ((uint64_t)1 << lt) - 1;
    F29 v = f29_mul(load_f29(tw_lo + (ex & mask)), load_f29(tw_hi + ((ex >> lt) & mask)), fp);
    if (g_lo != nullptr) {
        const F29 g = f29_mul(load_f29(g_lo + (q & mask)), load_f29(g_hi + ((q >> lt) & mask)), fp);
        v = f29_mul(v, g, fp);
    }
    store_fr(out + id, f29_to_sat(f29_canon(v, fp)));
}
