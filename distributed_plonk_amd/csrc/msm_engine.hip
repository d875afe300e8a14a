// msm_engine.hip — Pippenger multi-scalar multiplication sum_i s_i * P_i on gfx950.
//
// What it replaces: ark-ec 0.3.0 VariableBaseMSM::multi_scalar_mul as called by the reference at
// src/worker.rs:179-182 (varMsm), :117-123,405 (commit_polynomial / round1), src/dispatcher.rs:1052
// and src/dispatcher2.rs:835-893.  Only the group element is contractual
// (Appendix A.1), so the GPU algorithm is free to differ from the CPU one:
//
//   1. digits      every scalar is cut into W = ceil((bits+1)/c) SIGNED c-bit digits, stored planar
//                                                                              (msm_digits_kernel)
//   2/3. sort      two-level LDS-staged counting sort of the (window, |digit|) keys: per-slice partition
//                  histograms + one exclusive scan, a coalesced level-1 scatter, then one workgroup per
//                  partition orders its L2-resident entries and emits the bucket offsets
//                                                           (sort_hist / scan_* / sort_scatter / sort_partition)
//   3b. schedule   bucket ids counting-sorted by size, so a wave's lanes own equally loaded buckets
//   4. accumulate  one lane per bucket walks its segment and adds the resident limb-form bases (negated for
//                  negative digits) into an XYZZ accumulator held in VGPRs   (msm_accumulate_kernel;
//                  same-x exceptional additions are redone by msm_accumulate_redo_kernel)
//   5. reduce      sum_d d*B_d per window as a pyramid of running-sum passes over chunks of K = 4 entries
//                  (msm_reduce_level_kernel, fast path + redo), then per-level sums (msm_points_sum_kernel)
//   6. fold        V_w = Sigma + sum_l 4^l A_l per window and the W window sums -> one point (W*c doublings), on the
//                  host; the result is returned as a normalised Jacobian triple (x, y, 1) / (1, 1, 0).
//   (opt.)         a fixed-base window table built at `init` (planes 2^(c*G*t) * P_i) lets a scalar's W windows share G bucket sets
//                  (msm_table_kernel); measured a wash on MI355X and off by default — see the comment there.
//
// Zero scalars and digit 0 never touch a bucket (the reference filters zeros too); scalar 1 needs no
// special case (digit 1 in window 0).  Infinity bases are skipped, P+P and P+(-P) are handled in
// ec_lazy.hpp (device) / ec.hpp (host fold).  No MFMA: ~10 Fq Montgomery products per (point, window) dominate everything — the
// kernel is v_mad_u64_u32 bound; algorithmic HBM bytes are n*(sizeof(affine)+32).
#include <algorithm>
#include <cstring>

#include "constants.h"
#include "ec.hpp"
#include "ec_lazy.hpp"
#include "plonk_internal.hpp"

#define BASES_BAD_CURVE 1u
#define BASES_BAD_RANGE 2u
#define BASES_BAD_FLAG 4u
template <int NQ> static const FpParams<NQ>& fq_params(int curve);
template <> const FpParams<8>& fq_params<8>(int) { return BN254_FQ_PARAMS; }
template <> const FpParams<12>& fq_params<12>(int) { return BLS12_381_FQ_PARAMS; }

// Wave priority of everything in an MSM that is NOT the bucket accumulation (build flag -DMSM_SIDE_PRIO=1..3; default 0 = no instruction): a SIMD's
// arbiter issues the highest-priority ready wave first and the oldest among equals, so beside another stream's accumulation (four long-lived
// VALU-bound waves per SIMD) the short memory- / latency-bound phases of the next MSM otherwise queue behind it for every instruction.
#ifndef MSM_SIDE_PRIO
#define MSM_SIDE_PRIO 0
#endif
#if MSM_SIDE_PRIO > 0
#define MSM_SIDE_PRIO_ENTER() __builtin_amdgcn_s_setprio(MSM_SIDE_PRIO)
#else
#define MSM_SIDE_PRIO_ENTER() ((void)0)
#endif

// ---------------------------------------------------------------------------------------------- helpers
template <typename T> __device__ __forceinline__ T load16(const T* p) {
    static_assert(sizeof(T) % 16 == 0, "16-byte multiple");
    T r;
    const uint4* s = reinterpret_cast<const uint4*>(p);
    uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
    return r;
}
template <typename T> __device__ __forceinline__ void store16(T* p, const T& v) {
    uint4* d = reinterpret_cast<uint4*>(p);
    const uint4* s = reinterpret_cast<const uint4*>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
}

// Signed c-bit digits: d_w in [-2^(c-1), 2^(c-1)], sum_w d_w 2^(wc) = s.  Bucket index |d|-1 in [0, 2^(c-1)),
// the sign travels in bit 31 of the sorted point index (a negative digit adds -P: y -> -y).  Halves the
// bucket count (and the reduction work) of the unsigned decomposition arkworks uses; the group element
// is the same.  W*c >= bits+1 guarantees the last carry is absorbed.
__device__ __forceinline__ uint32_t scalar_raw_digit(const uint32_t* s, int bit0, int c) {
    const int limb = bit0 >> 5, off = bit0 & 31;
    if (limb >= 8) return 0;
    uint64_t v = s[limb];
    if (limb + 1 < 8) v |= (uint64_t)s[limb + 1] << 32;
    return (uint32_t)(v >> off) & ((1u << c) - 1);
}

// ---- 1: digits, planar: dig[w*n + i] = magnitude | sign << 31   (magnitude 0 = no contribution)
// scalars_mont != 0: the scalars are Fr in Montgomery form and `into_repr` (commit_polynomial, worker.rs:117-123) is taken here,
// on the fly, instead of in a separate pass that wrote 32 B per scalar to HBM and read them back.
// `len` <= n scalars are valid; positions len .. n-1 get zero digits (a batch of commitments pads its shorter polynomials).
__global__ void __launch_bounds__(256) msm_digits_kernel(const uint32_t* __restrict__ scalars, uint64_t n, uint64_t len, int c, int W,
                                                         uint32_t* __restrict__ dig, int scalars_mont, const FpParams<8> FR) {
    MSM_SIDE_PRIO_ENTER();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i >= len) {
        for (int w = 0; w < W; w++) dig[(uint64_t)w * n + i] = 0;
        return;
    }
    uint32_t s[8];
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + 8 * i);
    const uint4 lo = sp[0], hi = sp[1];
    s[0] = lo.x; s[1] = lo.y; s[2] = lo.z; s[3] = lo.w; s[4] = hi.x; s[5] = hi.y; s[6] = hi.z; s[7] = hi.w;
    if (scalars_mont) {
        Fp<8> v;
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = s[k];
        v = fp_from_mont(v, FR);
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = v.l[k];
    }
    const uint32_t half = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
        // static limb selection keeps s[] in registers
        const int bit0 = w * c, limb = bit0 >> 5, off = bit0 & 31;
        uint64_t v = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k == limb) v |= s[k];
            if (k == limb + 1) v |= (uint64_t)s[k] << 32;
        }
        uint32_t raw = ((uint32_t)(v >> off) & ((1u << c) - 1)) + carry;
        carry = raw > half;
        const uint32_t mag = carry ? (1u << c) - raw : raw;
        dig[(uint64_t)w * n + i] = mag | (carry << 31);
    }
}

// ---- 3: two-level LDS-staged counting sort of the (window, bucket) keys.
// Level 1 splits every window into 2^lp partitions by the top bits of the bucket index (all traffic in
// contiguous pieces, no global atomics: per-slice histograms + one exclusive scan give every (slice, partition)
// its exact range).  Level 2 sorts one partition per workgroup with LDS counters; its ~64 KiB of entries stay
// in L2.  Replaces a histogram + scatter pair that issued one HBM atomic and one 4-byte random store per
// (point, window) — 16x write amplification once n*W*4 B outgrows the 256 MiB Infinity Cache.
#define SORT_SLICE_LOG 14
#define SORT_SLICE (1 << SORT_SLICE_LOG)      // entries per level-1 workgroup
// A level-1 entry is  bucket_low << 15 | sign << 14 | (point index - slice start): the slice number is not stored, the
// level-2 workgroup recovers it from the entry's position (its partition is the concatenation of one run per slice,
// whose boundaries are the scanned histogram).  This keeps 2^10 level-1 partitions — 16-entry contiguous runs per
// (slice, partition) — for every window width; packing the full point index used to force 2^12 partitions (4-entry runs,
// sort 5.7 -> 10 ms) once c reached 20.
struct SortGeom {
    uint64_t n;
    int W, cb, lp, low_bits;
    uint32_t nblk;               // slices per window
    uint32_t nreal;              // W << lp : real partitions; the W "digit == 0" partitions follow them
    uint32_t idx_bits;           // > 0: level-1 entries carry the FULL point index in idx_bits bits (low_bits + 1 + idx_bits <= 32), so the
                                 // level-2 workgroup needs no search for the slice; 0: index within the slice only (wide windows / huge n)
    uint32_t stage_cap;          // entries the staged level-2 kernel orders in LDS at a time (what its LDS budget leaves beside the counters and slice starts)
};

__global__ void __launch_bounds__(256) sort_hist_kernel(const uint32_t* __restrict__ dig, SortGeom g, uint32_t* __restrict__ blk_hist) {
    MSM_SIDE_PRIO_ENTER();
    extern __shared__ uint32_t h[];
    const uint32_t np = (1u << g.lp) + 1;
    for (uint32_t k = threadIdx.x; k < np; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const uint32_t w = blockIdx.y, blk = blockIdx.x;
    const uint64_t beg = (uint64_t)blk * SORT_SLICE, end = beg + SORT_SLICE < g.n ? beg + SORT_SLICE : g.n;
    const uint32_t* d = dig + (uint64_t)w * g.n;
    for (uint64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t mag = d[i] & 0x7fffffffu;
        const uint32_t part = mag ? ((mag - 1) >> g.low_bits) : (np - 1);
        atomicAdd(&h[part], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < np; k += blockDim.x) {
        const uint64_t pid = (k == np - 1) ? (uint64_t)g.nreal + w : ((uint64_t)w << g.lp) + k;
        blk_hist[pid * g.nblk + blk] = h[k];
    }
}

// Level-1 scatter.  A wave's 64 entries go to 64 different partitions, so storing them straight to HBM costs one
// write transaction per 4-byte entry (measured WRITE_SIZE 7x the payload).  Instead the slice is first ordered by
// partition in LDS (counts -> exclusive scan -> ranks), then copied out with consecutive lanes writing consecutive
// addresses of each partition's run.
#define SCATTER_THREADS 1024     // 16 waves on a ~74 KiB LDS footprint: two workgroups = eight waves per SIMD hide the LDS-atomic and HBM latency
__global__ void __launch_bounds__(SCATTER_THREADS) sort_scatter_kernel(const uint32_t* __restrict__ dig, SortGeom g, const uint32_t* __restrict__ blk_off,
                                                                       uint32_t* __restrict__ tmp) {
    MSM_SIDE_PRIO_ENTER();
    extern __shared__ uint32_t sm[];
    const uint32_t np = (1u << g.lp) + 1;                 // last bin: zero digits (dropped)
    uint32_t* cnt = sm;                                   // [np]   counts, then running cursors
    uint32_t* loc = sm + np;                              // [np+1] exclusive local offsets
    uint32_t* buf = sm + 2 * np + 1;                      // [SORT_SLICE] entries ordered by partition
    const uint32_t w = blockIdx.y, blk = blockIdx.x;
    for (uint32_t k = threadIdx.x; k < np; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    const uint64_t beg = (uint64_t)blk * SORT_SLICE, end = beg + SORT_SLICE < g.n ? beg + SORT_SLICE : g.n;
    const uint32_t* d = dig + (uint64_t)w * g.n;
    for (uint64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t mag = d[i] & 0x7fffffffu;
        atomicAdd(&cnt[mag ? ((mag - 1) >> g.low_bits) : (np - 1)], 1u);
    }
    __syncthreads();
    // exclusive scan of the np counts (np <= 8193): every lane a contiguous strip, then a strip-sum scan
    __shared__ uint32_t strip[SCATTER_THREADS];
    const uint32_t per = (np + blockDim.x - 1) / blockDim.x;
    const uint32_t s0 = threadIdx.x * per < np ? threadIdx.x * per : np, s1 = s0 + per < np ? s0 + per : np;
    uint32_t acc = 0;
    for (uint32_t k = s0; k < s1; k++) acc += cnt[k];
    strip[threadIdx.x] = acc;
    __syncthreads();
    for (int dd = 1; dd < SCATTER_THREADS; dd <<= 1) {
        const uint32_t t = (int)threadIdx.x >= dd ? strip[threadIdx.x - dd] : 0;
        __syncthreads();
        strip[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = strip[threadIdx.x] - acc;
    for (uint32_t k = s0; k < s1; k++) { const uint32_t v = cnt[k]; loc[k] = run; cnt[k] = run; run += v; }
    if (threadIdx.x == blockDim.x - 1) loc[np] = strip[blockDim.x - 1];
    __syncthreads();
    // rank into LDS.  While the packed entry leaves room (low_bits + 15 + lp <= 32, i.e. windows up to 18 bits) the partition
    // number rides in the entry's top bits, so the copy-out below needs no search.
    const uint32_t low_mask = (1u << g.low_bits) - 1;
    const uint32_t ebits = (uint32_t)g.low_bits + SORT_SLICE_LOG + 1;
    const bool packed = ebits + (uint32_t)g.lp <= 32;
    for (uint64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t e = d[i], mag = e & 0x7fffffffu;
        if (!mag) continue;
        const uint32_t key = mag - 1, part = key >> g.low_bits;
        const uint32_t pos = atomicAdd(&cnt[part], 1u);
        uint32_t v = ((key & low_mask) << (SORT_SLICE_LOG + 1)) | ((e >> 31) << SORT_SLICE_LOG) | (uint32_t)(i - beg);
        if (packed && g.lp) v |= part << ebits;
        buf[pos] = v;
    }
    __syncthreads();
    // copy out: position j of the ordered slice belongs to the partition whose [loc[k], loc[k+1]) contains it
    const uint32_t nreal_local = loc[np - 1];             // entries with a non-zero digit
    const uint32_t emask = ebits >= 32 ? 0xffffffffu : ((1u << ebits) - 1);
    for (uint32_t j = threadIdx.x; j < nreal_local; j += blockDim.x) {
        const uint32_t v = buf[j];
        uint32_t lo;
        if (packed) {
            lo = g.lp ? (v >> ebits) : 0;
        } else {
            lo = 0;
            uint32_t hi = np - 1;                         // largest k with loc[k] <= j
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (loc[mid] <= j) lo = mid; else hi = mid; }
        }
        const uint64_t pid = ((uint64_t)w << g.lp) + lo;
        uint32_t out = packed ? (v & emask) : v;
        if (g.idx_bits) {                                 // re-pack with the full point index: low | sign | (slice start + index in slice)
            const uint32_t in = out & (SORT_SLICE - 1), sg = (out >> SORT_SLICE_LOG) & 1u, lowb = out >> (SORT_SLICE_LOG + 1);
            out = (lowb << (g.idx_bits + 1)) | (sg << g.idx_bits) | (uint32_t)(beg + in);
        }
        tmp[blk_off[pid * g.nblk + blk] + (j - loc[lo])] = out;
    }
}

// one workgroup per real partition: final order + bucket offsets
// The bucket-size histogram of step 3b (SIZE_BINS bins, bin 0 = largest) is taken HERE when `ghist` is given (round 5: plain bucket sets, where a
// bucket is one sorted segment and its size is the count this kernel has in its hands): one launch and one pass over the offsets less per MSM.
#define SIZE_BINS 256
struct SizeHist {
    uint32_t* ghist;          // nullptr: step 3b counts the sizes itself (bucket sets merged over windows: the fixed-base table)
    uint32_t bin_shift;
    uint64_t nbuckets;
};
__device__ __forceinline__ uint32_t size_bin_of(uint32_t entries, uint32_t bin_shift) {
    const uint32_t sz = entries >> bin_shift;
    return (SIZE_BINS - 1) - (sz < SIZE_BINS - 1 ? sz : SIZE_BINS - 1);      // bin 0 = largest
}
__global__ void __launch_bounds__(256) sort_partition_kernel(const uint32_t* __restrict__ tmp, SortGeom g, const uint32_t* __restrict__ blk_off,
                                                             uint32_t* __restrict__ sorted, uint32_t* __restrict__ offsets, const SizeHist sz) {
    MSM_SIDE_PRIO_ENTER();
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t shist[SIZE_BINS];
    if (sz.ghist) shist[threadIdx.x] = 0;
    const uint32_t nlow = 1u << g.low_bits;
    uint32_t* cnt = lds;                       // [2^low_bits] counts -> cursors
    uint32_t* run0 = lds + nlow;               // [nblk] start of every slice's run inside this partition
    const uint64_t pid = blockIdx.x;
    const uint32_t* po = blk_off + pid * g.nblk;
    const uint32_t pbeg = po[0], pend = po[g.nblk];
    for (uint32_t k = threadIdx.x; k < nlow; k += blockDim.x) cnt[k] = 0;
    for (uint32_t k = threadIdx.x; k < g.nblk; k += blockDim.x) run0[k] = po[k];
    __syncthreads();
    const int sh = g.idx_bits ? (int)g.idx_bits + 1 : SORT_SLICE_LOG + 1;
    for (uint32_t j = pbeg + threadIdx.x; j < pend; j += blockDim.x) atomicAdd(&cnt[tmp[j] >> sh], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {                    // nlow <= 2048: a serial exclusive scan is negligible
        uint32_t run = pbeg;
        for (uint32_t k = 0; k < nlow; k++) {
            const uint32_t v = cnt[k];
            cnt[k] = run;
            run += v;
            if (sz.ghist && (pid << g.low_bits) + k < sz.nbuckets) shist[size_bin_of(v, sz.bin_shift)]++;
        }
    }
    __syncthreads();
    if (sz.ghist && shist[threadIdx.x]) atomicAdd(&sz.ghist[threadIdx.x], shist[threadIdx.x]);
    for (uint32_t k = threadIdx.x; k < nlow; k += blockDim.x) offsets[(pid << g.low_bits) + k] = cnt[k];
    if (pid == g.nreal - 1 && threadIdx.x == 0) offsets[(uint64_t)g.nreal << g.low_bits] = pend;      // end sentinel
    __syncthreads();
    const uint32_t in_mask = SORT_SLICE - 1;
    // Final placement: direct 4-byte scatter into the partition's 2^low_bits bucket runs.  Fine while the open runs of the
    // workgroups in flight fit the write-combining reach of L1/L2 (c <= 17 at 2^24: 1.5 ms); at c = 20 (512 runs per
    // partition) it costs 6.3 ms.  Ordering the partition in LDS first was measured and rejected: 64 KiB of staging per
    // workgroup leaves one workgroup per CU (7.3 ms), and 4096-entry partitions that would stage in 16 KiB double the
    // level-1 kernels instead (profiles/r01_msm_table_experiment.txt).
    if (g.idx_bits) {                          // entries carry the full point index: no slice search
        const uint32_t imask = (1u << g.idx_bits) - 1;
        for (uint32_t j = pbeg + threadIdx.x; j < pend; j += blockDim.x) {
            const uint32_t e = tmp[j];
            const uint32_t pos = atomicAdd(&cnt[e >> sh], 1u);
            sorted[pos] = (e & imask) | (((e >> g.idx_bits) & 1u) << 31);
        }
        return;
    }
    for (uint32_t j = pbeg + threadIdx.x; j < pend; j += blockDim.x) {
        const uint32_t e = tmp[j];
        uint32_t lo = 0, hi = g.nblk;          // slice = largest b with run0[b] <= j
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (run0[mid] <= j) lo = mid; else hi = mid; }
        const uint32_t pos = atomicAdd(&cnt[e >> sh], 1u);
        sorted[pos] = ((lo << SORT_SLICE_LOG) + (e & in_mask)) | (((e >> SORT_SLICE_LOG) & 1u) << 31);
    }
}

// Level 2 for WIDE windows (2^low_bits >= 256 buckets per partition, i.e. c >= 19 at 2^24 points): the direct scatter above keeps
// 512 one-cache-line runs open per workgroup for the workgroup's whole life, and with ~2000 workgroups in flight those partial lines
// fall out of L2 before they are full (6.2 ms at c = 20 against 1.5 ms at c = 17 for the same bytes).  Here the partition is ordered in
// LDS and leaves in consecutive addresses: counts -> exclusive scan -> ranks into the LDS buffer -> coalesced copy-out.  1024 lanes on
// ~78 KiB of LDS: two workgroups = eight waves per SIMD, like the level-1 scatter.  A partition larger than the buffer (g.stage_cap
// entries: above 2^24 points a partition holds 2^15 and more, and skewed scalars can fill one at any size) is ordered in several CHUNKS
// of consecutive buckets: every chunk streams the partition again (its 128 - 256 KiB are L2 / Infinity-Cache resident) and stages only
// the entries of its own buckets; a single chunk that still does not fit (one huge bucket) takes the direct path.  Late round 3: before,
// such partitions — all of them at 2^25 and 2^26 points — fell back to the direct kernel: sort 17.3 / 35.6 ms beside 30.6 / 59.9 ms of accumulation.
#define STAGE_THREADS 1024
#define STAGE_MAX_CHUNKS 8u
__global__ void __launch_bounds__(STAGE_THREADS) sort_partition_staged_kernel(const uint32_t* __restrict__ tmp, SortGeom g, const uint32_t* __restrict__ blk_off,
                                                                              uint32_t* __restrict__ sorted, uint32_t* __restrict__ offsets, const SizeHist sz) {
    MSM_SIDE_PRIO_ENTER();
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t shist[SIZE_BINS];
    if (sz.ghist && threadIdx.x < SIZE_BINS) shist[threadIdx.x] = 0;
    const uint32_t nlow = 1u << g.low_bits;
    uint32_t* cnt = lds;                       // [2^low_bits] counts -> cursors (relative to the partition start)
    uint32_t* run0 = lds + nlow;               // [nblk] start of every slice's run inside this partition (idx_bits == 0 only)
    uint32_t* buf = run0 + (g.idx_bits ? 0 : g.nblk);      // [stage_cap] final entries in bucket order
    uint32_t* strip = buf;                     // [STAGE_THREADS] scratch of the scan, dead before buf is filled (stage_cap >= STAGE_THREADS)
    const uint64_t pid = blockIdx.x;
    const uint32_t* po = blk_off + pid * g.nblk;
    const uint32_t pbeg = po[0], pend = po[g.nblk], len = pend - pbeg;
    const int sh = g.idx_bits ? (int)g.idx_bits + 1 : SORT_SLICE_LOG + 1;
    for (uint32_t k = threadIdx.x; k < nlow; k += blockDim.x) cnt[k] = 0;
    if (!g.idx_bits)
        for (uint32_t k = threadIdx.x; k < g.nblk; k += blockDim.x) run0[k] = po[k];
    __syncthreads();
    for (uint32_t j = pbeg + threadIdx.x; j < pend; j += blockDim.x) atomicAdd(&cnt[tmp[j] >> sh], 1u);
    __syncthreads();
    // exclusive scan of the nlow (<= 2048) counts: two entries per lane at most, then a strip-sum scan
    const uint32_t per = (nlow + blockDim.x - 1) / blockDim.x;
    const uint32_t s0 = threadIdx.x * per < nlow ? threadIdx.x * per : nlow, s1 = s0 + per < nlow ? s0 + per : nlow;
    uint32_t acc = 0;
    for (uint32_t k = s0; k < s1; k++) acc += cnt[k];
    strip[threadIdx.x] = acc;
    __syncthreads();
    for (int dd = 1; dd < STAGE_THREADS; dd <<= 1) {
        const uint32_t t = (int)threadIdx.x >= dd ? strip[threadIdx.x - dd] : 0;
        __syncthreads();
        strip[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = strip[threadIdx.x] - acc;
    __syncthreads();                           // every lane has read its strip value: the scratch may be overwritten from here on
    for (uint32_t k = s0; k < s1; k++) {
        const uint32_t v = cnt[k];
        cnt[k] = run;
        offsets[(pid << g.low_bits) + k] = pbeg + run;
        run += v;
        if (sz.ghist && (pid << g.low_bits) + k < sz.nbuckets) atomicAdd(&shist[size_bin_of(v, sz.bin_shift)], 1u);      // (zeroed before the first barrier above)
    }
    if (pid == g.nreal - 1 && threadIdx.x == 0) offsets[(uint64_t)g.nreal << g.low_bits] = pend;      // end sentinel
    __syncthreads();
    if (sz.ghist && threadIdx.x < SIZE_BINS && shist[threadIdx.x]) atomicAdd(&sz.ghist[threadIdx.x], shist[threadIdx.x]);
    const uint32_t in_mask = SORT_SLICE - 1;
    const uint32_t imask = g.idx_bits ? (1u << g.idx_bits) - 1 : 0;
    // chunks of consecutive buckets: one when the partition fits the buffer, else sized for 3/4 of it (bucket sizes fluctuate)
    const uint32_t cap = g.stage_cap;
    // (a partition far beyond that — the 2^14-bucket top window of a c = 20 decomposition puts 2^19 entries in each of its partitions — would stream
    //  itself dozens of times from one workgroup: it takes the direct path in one pass, as before)
    uint32_t nch = len <= cap ? 1u : min(nlow, (len + (cap - cap / 4) - 1) / (cap - cap / 4));
    const bool direct = nch > STAGE_MAX_CHUNKS;
    if (direct) nch = 1;
    const uint32_t bper = (nlow + nch - 1) / nch;
    for (uint32_t k0 = 0; k0 < nlow; k0 += bper) {
        const uint32_t k1 = k0 + bper < nlow ? k0 + bper : nlow;
        const uint32_t base = cnt[k0], cend = k1 < nlow ? cnt[k1] : len;       // cursors of this chunk's buckets are still at their starts
        const bool staged = !direct && cend - base <= cap;
        __syncthreads();                       // all lanes hold base / cend before the cursors move
        for (uint32_t j = pbeg + threadIdx.x; j < pend; j += blockDim.x) {
            const uint32_t e = tmp[j], bk = e >> sh;
            if (bk < k0 || bk >= k1) continue;
            uint32_t val;
            if (g.idx_bits) {
                val = (e & imask) | (((e >> g.idx_bits) & 1u) << 31);
            } else {
                uint32_t lo = 0, hi = g.nblk;  // slice = largest b with run0[b] <= j
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (run0[mid] <= j) lo = mid; else hi = mid; }
                val = ((lo << SORT_SLICE_LOG) + (e & in_mask)) | (((e >> SORT_SLICE_LOG) & 1u) << 31);
            }
            const uint32_t pos = atomicAdd(&cnt[bk], 1u);
            if (staged) buf[pos - base] = val; else sorted[pbeg + pos] = val;
        }
        __syncthreads();
        if (staged)
            for (uint32_t j = threadIdx.x; j < cend - base; j += blockDim.x) sorted[pbeg + base + j] = buf[j];
        __syncthreads();                       // the buffer is free for the next chunk
    }
}

// ---------------------------------------------------------------------------------------------- 3b: schedule buckets by size
// One lane owns one bucket, so a wave runs as long as its fullest bucket.  A counting sort of the bucket
// ids by (clamped) size, largest first, puts equally loaded buckets in the same wave.
// Bucket sets.  Plain: one set per window.  With the fixed-base window table (msm_table_kernel): G sets per scalar vector — set s = k*G + g of
// vector k collects the windows w = g, g + G, g + 2G, ... < W1 of that vector; the entries of (window w, bucket b) are the sorted segment
// ((k*W1 + w) << cb) + b and their bases come from plane t = w / G of the table (2^(c*G*t) * P_i).  Plain mode is G = W1 (one segment, plane 0).
struct SetGeom {
    uint32_t cb;          // log2 buckets per set
    uint32_t G;           // sets per scalar vector
    uint32_t W1;          // windows per scalar vector
    uint32_t bin_shift;   // size-bin resolution (entries >> bin_shift)
    uint64_t tab_stride;  // points between the planes of the table (0: plain)
};
// first segment of set-bucket sb; the following ones are seg_step(g) apart
__device__ __forceinline__ uint64_t seg_first(const SetGeom& g, uint32_t sb, uint32_t* w0) {
    const uint32_t s = sb >> g.cb, b = sb & ((1u << g.cb) - 1u);
    const uint32_t k = s / g.G, gi = s - k * g.G;
    *w0 = gi;
    return (((uint64_t)k * g.W1 + gi) << g.cb) + b;
}
__device__ __forceinline__ uint64_t seg_step(const SetGeom& g) { return (uint64_t)g.G << g.cb; }
__device__ __forceinline__ uint32_t bucket_entries(const uint32_t* offsets, uint32_t sb, const SetGeom& g) {
    uint32_t w;
    uint64_t seg = seg_first(g, sb, &w);
    uint32_t sz = 0;
    for (; w < g.W1; w += g.G, seg += seg_step(g)) sz += offsets[seg + 1] - offsets[seg];
    return sz;
}
__device__ __forceinline__ uint32_t size_bin(const uint32_t* offsets, uint32_t sb, const SetGeom& g) {
    return size_bin_of(bucket_entries(offsets, sb, g), g.bin_shift);     // merged buckets hold more entries: the host scales them into the 256 bins
}
__global__ void __launch_bounds__(256) bucket_size_hist_kernel(const uint32_t* __restrict__ offsets, uint64_t nbuckets, SetGeom g, uint32_t* __restrict__ ghist) {
    MSM_SIDE_PRIO_ENTER();
    __shared__ uint32_t h[SIZE_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nbuckets) atomicAdd(&h[size_bin(offsets, (uint32_t)b, g)], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(SIZE_BINS) bucket_size_scan_kernel(const uint32_t* __restrict__ ghist, uint32_t* __restrict__ bin_cursor) {
    MSM_SIDE_PRIO_ENTER();
    __shared__ uint32_t buf[SIZE_BINS];
    const uint32_t v = ghist[threadIdx.x];
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < SIZE_BINS; d <<= 1) {
        uint32_t t = (int)threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    bin_cursor[threadIdx.x] = buf[threadIdx.x] - v;
}
// ghist != nullptr (round 5): bin_cursor arrives ZEROED and every workgroup scans the finished histogram itself (256 values: eight LDS steps) — the
// one-workgroup scan launch in between is gone; ghist == nullptr: bin_cursor holds the scanned starts (bucket_size_scan_kernel).
__global__ void __launch_bounds__(256) bucket_size_place_kernel(const uint32_t* __restrict__ offsets, uint64_t nbuckets, SetGeom g,
                                                                uint32_t* __restrict__ bin_cursor, uint32_t* __restrict__ order, const uint32_t* __restrict__ ghist) {
    MSM_SIDE_PRIO_ENTER();
    __shared__ uint32_t h[SIZE_BINS];
    __shared__ uint32_t base[SIZE_BINS];
    __shared__ uint32_t pre[SIZE_BINS];
    h[threadIdx.x] = 0;
    uint32_t start = 0;
    if (ghist != nullptr) {
        const uint32_t v = ghist[threadIdx.x];
        pre[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < SIZE_BINS; d <<= 1) {
            const uint32_t t = (int)threadIdx.x >= d ? pre[threadIdx.x - d] : 0;
            __syncthreads();
            pre[threadIdx.x] += t;
            __syncthreads();
        }
        start = pre[threadIdx.x] - v;
    }
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bin = 0, rank = 0;
    if (b < nbuckets) { bin = size_bin(offsets, (uint32_t)b, g); rank = atomicAdd(&h[bin], 1u); }
    __syncthreads();
    if (h[threadIdx.x]) base[threadIdx.x] = start + atomicAdd(&bin_cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (b < nbuckets) order[base[bin] + rank] = (uint32_t)b;
}

// ---------------------------------------------------------------------------------------------- 2: exclusive scan (u32)
#define SCAN_ITEMS 16
#define SCAN_THREADS 256
#define SCAN_CHUNK (SCAN_ITEMS * SCAN_THREADS)

__global__ void __launch_bounds__(SCAN_THREADS) scan_block_sums_kernel(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ block_sums) {
    MSM_SIDE_PRIO_ENTER();
    __shared__ uint32_t red[SCAN_THREADS];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += in[base + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = SCAN_THREADS / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(1024) scan_top_kernel(uint32_t* __restrict__ block_sums, uint64_t nblocks, uint32_t* __restrict__ total_out) {
    MSM_SIDE_PRIO_ENTER();
    __shared__ uint32_t buf[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nblocks; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            uint32_t t = (int)threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        const uint32_t incl = buf[threadIdx.x];
        if (i < nblocks) block_sums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const uint32_t* __restrict__ in, uint64_t n, const uint32_t* __restrict__ block_offsets,
                                                                  uint32_t* __restrict__ out) {
    MSM_SIDE_PRIO_ENTER();
    __shared__ uint32_t buf[SCAN_THREADS];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    buf[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < SCAN_THREADS; d <<= 1) {
        uint32_t t = (int)threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = block_offsets[blockIdx.x] + buf[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// ---------------------------------------------------------------------------------------------- 4: bucket accumulation
// Limb geometry of the lazy base-field arithmetic per curve (flimb.hpp)
template <int NQ> struct LimbGeom;
template <> struct LimbGeom<8> { static constexpr int NL = 9, B = 29; };      // BN254 Fq, R' = 2^261
template <> struct LimbGeom<12> { static constexpr int NL = 14, B = 28; };    // BLS12-381 Fq, R' = 2^392

template <typename T> __device__ __forceinline__ T load8(const T* p) {
    static_assert(sizeof(T) % 8 == 0, "8-byte multiple");
    T r;
    const uint2* s = reinterpret_cast<const uint2*>(p);
    uint2* d = reinterpret_cast<uint2*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 8; i++) d[i] = s[i];
    return r;
}
template <typename T> __device__ __forceinline__ void store8(T* p, const T& v) {
    uint2* d = reinterpret_cast<uint2*>(p);
    const uint2* s = reinterpret_cast<const uint2*>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 8; i++) d[i] = s[i];
}

// Resident form of a base point (round 4).  Rounds 1-3 kept bases as unsaturated limbs (9 x 29 bits per coordinate: 72 B per BN254 point, 112 B on
// BLS12-381) so that a gather needed no re-limbing — but a 72-byte record straddles 64-byte sectors, and the accumulation gathers every base once
// per window: PMC showed 127 GB of fetches per batched launch for 3 GB of algorithmic bytes.  Bases now rest as the same canonical R'-Montgomery
// residues in SATURATED 32-bit words, x || y = 64 B (BN254) / 96 B (BLS12-381): a BN254 record is exactly one aligned 64-byte sector, fetched with
// four dwordx4 loads and re-limbed in registers (+30 of ~2500 VALU instructions per addition).  Infinity stays (0, 0).  Same box, alternating
// builds (profiles/r04_packed_bases_experiment.txt): accumulate 16.23 -> 15.6 ms at 2^24, commitments of the 2^24 step 275 -> 271 ms, of the 2^20
// step 23.9 -> 23.5 ms, BLS12-381 unchanged; 11 % less resident memory.
template <int NQ> using BaseRec = AffPt<NQ>;
template <int NQ> __device__ __forceinline__ AffL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> load_base(const BaseRec<NQ>* p) {
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    const AffPt<NQ> a = load16(p);
    AffL<NL, B> r;
    r.x = fl_from_sat<NL, B, NQ>(a.x);
    r.y = fl_from_sat<NL, B, NQ>(a.y);
    return r;
}
template <int NQ> __device__ __forceinline__ void store_base(BaseRec<NQ>* p, const AffL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>& q) {   // q canonical (< p), normalised
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    AffPt<NQ> a;
    a.x = fl_to_sat<NL, B, NQ>(q.x);
    a.y = fl_to_sat<NL, B, NQ>(q.y);
    store16(p, a);
}

// SRS bases: reference layout (x||y, R = 2^(32N) Montgomery, canonical) -> resident limb form (R' Montgomery)
template <int NQ>
__global__ void __launch_bounds__(256) bases_to_limbs_kernel(const AffPt<NQ>* __restrict__ in, uint64_t n,
                                                             BaseRec<NQ>* __restrict__ out,
                                                             const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P) {
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AffPt<NQ> a = load16(in + i);
    const FL<NL, B> fix = fl_load_const<NL, B>(P.r2fix);
    AffL<NL, B> o;
    o.x = fl_canon_lt2p(fl_mul(fl_from_sat<NL, B, NQ>(a.x), fix, P), P);
    o.y = fl_canon_lt2p(fl_mul(fl_from_sat<NL, B, NQ>(a.y), fix, P), P);
    store_base<NQ>(out + i, o);
}

template <int NQ>
__device__ __forceinline__ void store_std(XyzzPt<NQ>* dst, const XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>& acc,
                                          const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>& P) {
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    // back to the canonical R = 2^(32N) Montgomery form the rest of the pipeline (and the reference) uses
    const FL<NL, B> rs = fl_load_const<NL, B>(P.r_std);
    XyzzPt<NQ> o;
    o.x = fl_to_sat<NL, B, NQ>(fl_canon_lt2p(fl_mul(acc.x, rs, P), P));
    o.y = fl_to_sat<NL, B, NQ>(fl_canon_lt2p(fl_mul(acc.y, rs, P), P));
    o.zz = fl_to_sat<NL, B, NQ>(fl_canon_lt2p(fl_mul(acc.zz, rs, P), P));
    o.zzz = fl_to_sat<NL, B, NQ>(fl_canon_lt2p(fl_mul(acc.zzz, rs, P), P));
    store16(dst, o);
}

#define HEAVY_BUCKET 2048u      // entries above which a bucket is shared by HEAVY_SEGS workgroups
#define HEAVY_SEGS 8
// Hot kernel: lane i owns bucket order[i] (size-sorted).  Exceptional additions (same x) abort the
// bucket, which is queued for msm_accumulate_redo_kernel — keeps calls, scratch and the doubling
// formula out of this kernel.
template <int NQ, bool FUSED_Y3>
__global__ void __launch_bounds__(256) msm_accumulate_kernel(const BaseRec<NQ>* __restrict__ bases,
                                                             const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ order, uint64_t nbuckets, SetGeom geom,
                                                             uint32_t heavy_thresh, XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ buckets, uint32_t* __restrict__ redo_count,
                                                             uint32_t* __restrict__ redo_list, uint32_t* __restrict__ heavy_count,
                                                             uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ work,
                                                             const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P) {
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    // work == nullptr: one lane per bucket, grid = nbuckets / 256.  work != nullptr: PERSISTENT waves — the grid is a fixed number of workgroups
    // (msm_acc_persist per CU, default 4 = every wave slot the registers allow) and every wave takes the next 64 buckets of the size-ordered
    // list from a global counter until the list is exhausted: no rounds of workgroups, so the 2^14 thousand-entry buckets of a c = 20 top window
    // (the launch's 13 ms critical path, started first) never hold a half-empty round open behind them, and the tail is one 64-bucket group per
    // wave: 16.6 -> 16.1 ms alone at 2^24, commitments -1.4 ... -2.4 % per step (profiles/r03_commit_overlap_experiment.txt).
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (;;) {
        if (work) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(work, 64u);
            base = (uint32_t)__shfl((int)base, 0);
            if (base >= nbuckets) break;
            i = (uint64_t)base + lane;
        }
        if (i < nbuckets) {
            const uint32_t b = order[i];
            const uint32_t total = bucket_entries(offsets, b, geom);
            if (total > heavy_thresh) {   // skewed scalars / a nearly empty top window: one lane must not walk it alone
                heavy_list[atomicAdd(heavy_count, 1u)] = b;
            } else {
                // ONE loop over the bucket's entries, across its segments (one per table plane; plain mode: a single segment).  A loop per segment
                // would run every segment as long as the wave's LONGEST one: the lanes of a wave are matched by total size, not per segment —
                // measured +38 % (2^24 points) to +80 % (2^22) on the merged accumulation, which rounds 1-2 mistook for a gather penalty.
                uint32_t w;
                uint64_t seg = seg_first(geom, b, &w);
                const uint64_t step = seg_step(geom);
                uint32_t j = offsets[seg], end = offsets[seg + 1];
                const BaseRec<NQ>* tb = bases;                               // plane t: 2^(c*G*t) * P_i
                XyzzL<NL, B> acc = xyzzl_inf<NL, B>();
                bool ok = true;
                for (uint32_t it = 0; it < total; it++) {
                    while (j == end) { seg += step; tb += geom.tab_stride; j = offsets[seg]; end = offsets[seg + 1]; }   // `total` guarantees a next entry
                    const uint32_t e = sorted[j++];
                    AffL<NL, B> q = load_base<NQ>(tb + (e & 0x7fffffffu));
                    if (affl_is_inf(q)) continue;
                    if (e >> 31) q = affl_neg(q, P);
                    if (!xyzzl_madd_fast<NL, B, FUSED_Y3>(acc, q, P)) { ok = false; break; }
                }
                if (ok) store8(buckets + b, acc);
                else redo_list[atomicAdd(redo_count, 1u)] = b;
            }
        }
        if (!work) break;
    }
}

// Heavy buckets: HEAVY_SEGS workgroups share one bucket (strided slices), every lane accumulates a strided subset
// with the complete addition, an LDS tree folds the workgroup, msm_heavy_finish_kernel folds the segments.
template <int NQ>
__global__ void __launch_bounds__(256) msm_heavy_kernel(const BaseRec<NQ>* __restrict__ bases,
                                                        const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
                                                        SetGeom geom,
                                                        const uint32_t* __restrict__ heavy_count, const uint32_t* __restrict__ heavy_list,
                                                        XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ partial,
                                                        const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P) {
    MSM_SIDE_PRIO_ENTER();
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XyzzL<NL, B>* sh = reinterpret_cast<XyzzL<NL, B>*>(smem_raw);
    const uint32_t total = *heavy_count, seg = blockIdx.y;
    for (uint32_t h = blockIdx.x; h < total; h += gridDim.x) {
        const uint32_t b = heavy_list[h];
        XyzzL<NL, B> acc = xyzzl_inf<NL, B>();
        uint32_t w;
        uint64_t sg = seg_first(geom, b, &w);
        const BaseRec<NQ>* tb = bases;
        for (; w < geom.W1; w += geom.G, sg += seg_step(geom), tb += geom.tab_stride) {
            const uint32_t beg = offsets[sg], end = offsets[sg + 1];
            for (uint32_t j = beg + seg * blockDim.x + threadIdx.x; j < end; j += HEAVY_SEGS * blockDim.x) {
                const uint32_t e = sorted[j];
                AffL<NL, B> q = load_base<NQ>(tb + (e & 0x7fffffffu));
                if (affl_is_inf(q)) continue;            // before negating: affl_neg would turn (0,0) into (0, 2p)
                if (e >> 31) q = affl_neg(q, P);
                acc = xyzzl_madd(acc, q, P);
            }
        }
        __syncthreads();
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int d = blockDim.x / 2; d > 0; d >>= 1) {
            if ((int)threadIdx.x < d) {
                const XyzzL<NL, B> a = sh[threadIdx.x], b2 = sh[threadIdx.x + d];
                sh[threadIdx.x] = xyzzl_add(a, b2, P);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) store8(partial + (uint64_t)h * HEAVY_SEGS + seg, sh[0]);
    }
}

template <int NQ>
__global__ void __launch_bounds__(64) msm_heavy_finish_kernel(const uint32_t* __restrict__ heavy_count, const uint32_t* __restrict__ heavy_list,
                                                              const XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ partial,
                                                              XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ buckets,
                                                              const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P) {
    MSM_SIDE_PRIO_ENTER();
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    const uint32_t total = *heavy_count;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < total; h += gridDim.x * blockDim.x) {
        XyzzL<NL, B> acc = xyzzl_inf<NL, B>();
        for (int s = 0; s < HEAVY_SEGS; s++) acc = xyzzl_add(acc, load8(partial + (uint64_t)h * HEAVY_SEGS + s), P);
        store8(buckets + heavy_list[h], acc);
    }
}

template <int NQ>
__global__ void __launch_bounds__(64) msm_accumulate_redo_kernel(const BaseRec<NQ>* __restrict__ bases,
                                                                 const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offsets,
                                                                 SetGeom geom,
                                                                 XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ buckets, const uint32_t* __restrict__ redo_count,
                                                                 const uint32_t* __restrict__ redo_list,
                                                                 const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P) {
    MSM_SIDE_PRIO_ENTER();
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    const uint32_t total = *redo_count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t b = redo_list[i];
        XyzzL<NL, B> acc = xyzzl_inf<NL, B>();
        uint32_t w;
        uint64_t sg = seg_first(geom, b, &w);
        const BaseRec<NQ>* tb = bases;
        for (; w < geom.W1; w += geom.G, sg += seg_step(geom), tb += geom.tab_stride) {
            const uint32_t beg = offsets[sg], end = offsets[sg + 1];
            for (uint32_t j = beg; j < end; j++) {
                const uint32_t e = sorted[j];
                AffL<NL, B> q = load_base<NQ>(tb + (e & 0x7fffffffu));
                if (affl_is_inf(q)) continue;
                if (e >> 31) q = affl_neg(q, P);
                acc = xyzzl_madd(acc, q, P);
            }
        }
        store8(buckets + b, acc);
    }
}

#ifndef REDUCE_LOGK
#define REDUCE_LOGK 2
#endif
#define REDUCE_K (1u << REDUCE_LOGK)
// ---------------------------------------------------------------------------------------------- 5: window reduction
// V_w = sum_j (j+1) * B_j over the 2^cb buckets of a window, as a short pyramid of running-sum passes:
//   level l holds n_l = 2^cb / K^l entries E_j with weights (j+1) (K = REDUCE_K = 4: the serial depth 2K*log_K(2^cb) is what
//   bounds this phase, not its work); one lane takes the STRIDED chunk {ch + t * nch_l : t < K_l} (nch_l = n_l / K_l chunks; K_l =
//   min(K, n_l)) and emits  acc = sum_t t * E_(ch + t nch)  and  S_ch = sum_t E_(ch + t nch)  (2K point additions, no scalar
//   multiplications).  Since (j+1) = (ch+1) + t * nch_l:  sum_j (j+1) E_j = sum_ch (ch+1) S_ch + nch_l * sum_ch acc_ch — the S values
//   are the next level's entries with the same kind of weights.  With A_l = sum of level l's acc values and Sigma = the single entry
//   left at the top:  V_w = Sigma + sum_l nch_l * A_l  (host: Horner, log2 K_l doublings per level).  Strided rather than contiguous
//   chunks (round 3): consecutive lanes read consecutive 144-byte entries, a wave's working set per step is 9 KiB instead of 36 KiB.
// All in lazy limb arithmetic (ec_lazy.hpp); only the W*(L+1) results are converted to the standard form.
template <int NQ>
__global__ void __launch_bounds__(256) msm_reduce_level_kernel(const XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ in, uint64_t n_in,
                                                               uint32_t K, uint64_t nch, uint64_t total,
                                                               XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ out_acc,
                                                               XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ out_s,
                                                               uint32_t* __restrict__ redo_count, uint32_t* __restrict__ redo_list,
                                                               const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P) {
    MSM_SIDE_PRIO_ENTER();
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const uint64_t w = id / nch, ch = id % nch;
    const XyzzL<NL, B>* E = in + w * n_in + ch;           // STRIDED chunk: entries ch, ch + nch, ..., ch + (K-1)*nch
    XyzzL<NL, B> running = xyzzl_inf<NL, B>(), acc = xyzzl_inf<NL, B>();
    bool ok = true;
    for (uint32_t d = K; d-- > 0;) {
        if (!xyzzl_add_fast(acc, running, P)) { ok = false; break; }
        if (!xyzzl_add_fast(running, load8(E + (uint64_t)d * nch), P)) { ok = false; break; }
    }
    if (ok) {
        store8(out_acc + id, acc);
        store8(out_s + id, running);
    } else {
        redo_list[atomicAdd(redo_count, 1u)] = (uint32_t)id;      // same-x additions: redone by the complete kernel
    }
}

template <int NQ>
__global__ void __launch_bounds__(64) msm_reduce_level_redo_kernel(const XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ in, uint64_t n_in,
                                                                   uint32_t K, uint64_t nch,
                                                                   XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ out_acc,
                                                                   XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ out_s,
                                                                   const uint32_t* __restrict__ redo_count, const uint32_t* __restrict__ redo_list,
                                                                   const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P) {
    MSM_SIDE_PRIO_ENTER();
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    const uint32_t total = *redo_count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint64_t id = redo_list[i];
        const uint64_t w = id / nch, ch = id % nch;
        const XyzzL<NL, B>* E = in + w * n_in + ch;
        XyzzL<NL, B> running = xyzzl_inf<NL, B>(), acc = xyzzl_inf<NL, B>();
        for (uint32_t d = K; d-- > 0;) {
            acc = xyzzl_add(acc, running, P);
            running = xyzzl_add(running, load8(E + (uint64_t)d * nch), P);
        }
        store8(out_acc + id, acc);
        store8(out_s + id, running);
    }
}

// block (l, w): sum of `count[l]` consecutive limb-form points starting at src[l] + w*count[l]  ->  standard form
struct SumJobs {
    const void* src[16];
    uint64_t count[16];
    uint64_t wstride[16];      // points between the inputs of consecutive windows (0: = count, the dense layout of the pyramid arrays)
};
// RAW: the block's sum stays in the limb form (input of a second, combining launch) instead of being converted to the standard form.
// Large pyramids (c = 20: 2^17 level-0 values per window) are summed by `nsplit` workgroups per (level, window) and a second launch
// folds the nsplit partials — one workgroup per (level, window) used to walk 512 points per lane: 6.6 of the 9 ms a c = 20 reduction took.
template <int NQ, bool RAW = false>
__global__ void __launch_bounds__(256) msm_points_sum_kernel(SumJobs jobs, XyzzPt<NQ>* __restrict__ out, uint32_t nlevels, uint32_t nsplit,
                                                             const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P,
                                                             XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ out_raw = nullptr) {
    MSM_SIDE_PRIO_ENTER();
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XyzzL<NL, B>* sh = reinterpret_cast<XyzzL<NL, B>*>(smem_raw);
    const uint32_t l = blockIdx.x, w = blockIdx.y, z = blockIdx.z;
    const uint64_t cnt = jobs.count[l];
    const XyzzL<NL, B>* src = reinterpret_cast<const XyzzL<NL, B>*>(jobs.src[l]) + (uint64_t)w * (jobs.wstride[l] ? jobs.wstride[l] : cnt);
    XyzzL<NL, B> acc = xyzzl_inf<NL, B>();
    for (uint64_t i = (uint64_t)z * blockDim.x + threadIdx.x; i < cnt; i += (uint64_t)nsplit * blockDim.x) acc = xyzzl_add(acc, load8(src + i), P);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int d = blockDim.x / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            const XyzzL<NL, B> a = sh[threadIdx.x], b2 = sh[threadIdx.x + d];
            sh[threadIdx.x] = xyzzl_add(a, b2, P);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (RAW) store8(out_raw + ((uint64_t)w * nlevels + l) * nsplit + z, sh[0]);
        else store_std<NQ>(out + ((uint64_t)w * nlevels + l) * nsplit + z, sh[0], P);
    }
}

// ---------------------------------------------------------------------------------------------- 5b: window reduction as a grid
// "msm_reduce_grid" (experiment, off by default): V_w = sum_j (j+1) B_j with the bucket index split as j = hi * L + lo
// (L = 2^ceil(cb/2) columns, H = 2^cb / L rows):
//     V_w = sum_lo (lo+1) R_lo + L * sum_hi hi * C_hi,      R_lo = sum_hi B[hi][lo] (column sums),  C_hi = sum_lo B[hi][lo] (row sums),
// and a weighted sum of <= 1024 points is taken bit by bit: sum_i k_i X_i = sum_b 2^b * (sum of the X_i whose weight k_i has bit b) — the
// host's Horner supplies the 2^b.  Same two additions per bucket as the pyramid, but every sum is a TREE: the serial depth of a window's
// reduction is T + log2(SEG) additions for the row / column sums (one launch, both kinds side by side) plus <= 2 + 8 for the bit sums,
// against 5 additions for each of the pyramid's (c-1)/2 dependent launches and a 16-deep per-level sum.  What it targets is the latency-
// bound reduction of SMALL problems (2^20 points: 8 + 2 launches, 1.7 ms beside 2.4 ms of accumulation); at 2^24 points the work is the same.
//
// One workgroup of 256 lanes = CW outputs x SEG interleaved segments; a lane adds T = Y / SEG inputs, then the SEG partial sums of an
// output meet in an LDS tree.  kind 0 (blockIdx.x < colblocks): column sums — lane (seg, cx), consecutive lanes read consecutive columns
// of one row; kind 1: row sums — lane (cx, seg), consecutive lanes read consecutive entries of one row.
template <int NQ>
__global__ void __launch_bounds__(256) msm_grid_sums_kernel(const XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ in, uint32_t logL, uint32_t logH,
                                                            uint32_t log_segc, uint32_t log_segr, uint32_t colblocks,
                                                            XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ out_r,
                                                            XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ out_c,
                                                            const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P) {
    MSM_SIDE_PRIO_ENTER();
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XyzzL<NL, B>* sh = reinterpret_cast<XyzzL<NL, B>*>(smem_raw);
    const uint32_t tid = threadIdx.x, w = blockIdx.y;
    const bool col = blockIdx.x < colblocks;
    const uint32_t blk = col ? blockIdx.x : blockIdx.x - colblocks;
    const uint32_t log_seg = col ? log_segc : log_segr;
    const uint32_t SEG = 1u << log_seg, CW = 256u >> log_seg;
    const uint32_t seg = col ? tid / CW : tid & (SEG - 1);
    const uint32_t cx = col ? tid % CW : tid >> log_seg;
    const uint32_t X = 1u << (col ? logL : logH), Y = 1u << (col ? logH : logL);
    const uint32_t x = blk * CW + cx;
    const XyzzL<NL, B>* E = in + ((uint64_t)w << (logL + logH));
    XyzzL<NL, B> acc = xyzzl_inf<NL, B>();
    if (x < X) {
        for (uint32_t y = seg; y < Y; y += SEG) {
            const uint64_t j = col ? ((uint64_t)y << logL) + x : ((uint64_t)x << logL) + y;
            acc = xyzzl_add(acc, load8(E + j), P);
        }
    }
    sh[tid] = acc;
    __syncthreads();
    const uint32_t stride = col ? CW : 1u;
    for (uint32_t d = SEG >> 1; d > 0; d >>= 1) {
        if (seg < d) {
            const XyzzL<NL, B> a = sh[tid], b2 = sh[tid + d * stride];
            sh[tid] = xyzzl_add(a, b2, P);
        }
        __syncthreads();
    }
    if (seg == 0 && x < X) store8((col ? out_r : out_c) + (uint64_t)w * X + x, sh[tid]);
}

// block (b, w): b <= logL: the column sums R_lo whose weight (lo + 1) has bit b;  b > logL: the row sums C_hi whose weight hi has bit b - logL - 1.
// The indices with the bit set are enumerated (no lane walks an entry it skips); standard-form result at out[w * nbits + b].
template <int NQ>
__global__ void __launch_bounds__(256) msm_bit_sums_kernel(const XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ r_sums,
                                                           const XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>* __restrict__ c_sums, uint32_t logL, uint32_t logH,
                                                           XyzzPt<NQ>* __restrict__ out, const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P) {
    MSM_SIDE_PRIO_ENTER();
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XyzzL<NL, B>* sh = reinterpret_cast<XyzzL<NL, B>*>(smem_raw);
    const uint32_t b = blockIdx.x, w = blockIdx.y, nbits = logL + 1 + logH;
    const bool cols = b <= logL;
    const uint32_t bit = cols ? b : b - logL - 1;
    const uint32_t n = 1u << (cols ? logL : logH), add = cols ? 1u : 0u;
    const XyzzL<NL, B>* src = (cols ? r_sums : c_sums) + (uint64_t)w * n;
    XyzzL<NL, B> acc = xyzzl_inf<NL, B>();
    const uint32_t half = n > 1 ? n >> 1 : 1;
    for (uint32_t u = threadIdx.x; u < half; u += blockDim.x) {
        const uint32_t k = ((u >> bit) << (bit + 1)) | (1u << bit) | (u & ((1u << bit) - 1));      // u-th weight with bit `bit` set
        if (k >= add && k - add < n) acc = xyzzl_add(acc, load8(src + (k - add)), P);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int d = blockDim.x / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            const XyzzL<NL, B> a = sh[threadIdx.x], b2 = sh[threadIdx.x + d];
            sh[threadIdx.x] = xyzzl_add(a, b2, P);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) store_std<NQ>(out + (uint64_t)w * nbits + b, sh[0], P);
}

// ---------------------------------------------------------------------------------------------- ark layout -> compact
// arkworks GroupAffine { x, y, infinity: bool } padded to 8 bytes: stride 16*Q64 + 8 bytes.
template <int NQ>
__global__ void __launch_bounds__(256) bases_convert_kernel(const uint32_t* __restrict__ raw, uint64_t n, AffPt<NQ>* __restrict__ out,
                                                            unsigned long long* __restrict__ bad /* [0]: first offending index, [1]: reasons */) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* s = raw + i * (2 * NQ + 2);
    AffPt<NQ> p;
    const bool inf = (s[2 * NQ] & 0xff) != 0;
    if (bad != nullptr && (s[2 * NQ] & 0xff) > 1) {            // `infinity: bool` is 0 or 1 in Rust: anything else is not a GroupAffine at this stride
        atomicMin(bad, (unsigned long long)i);
        atomicOr(bad + 1, (unsigned long long)BASES_BAD_FLAG);
    }
#pragma unroll
    for (int k = 0; k < NQ; k++) { p.x.l[k] = inf ? 0 : s[k]; p.y.l[k] = inf ? 0 : s[NQ + k]; }
    store16(out + i, p);
}

int bases_convert_ark(int curve, const void* d_raw, size_t n, void* d_compact, unsigned long long* d_bad, hipStream_t stream) {
    if (n == 0) return PLONK_OK;
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    if (curve == PLONK_BN254)
        hipLaunchKernelGGL(bases_convert_kernel<8>, dim3(grid), dim3(256), 0, stream, (const uint32_t*)d_raw, (uint64_t)n, (AffPt<8>*)d_compact, d_bad);
    else
        hipLaunchKernelGGL(bases_convert_kernel<12>, dim3(grid), dim3(256), 0, stream, (const uint32_t*)d_raw, (uint64_t)n, (AffPt<12>*)d_compact, d_bad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "bases_convert launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- the SRS, checked at the boundary
// Every base must be a point of the curve: coordinates reduced, y^2 = x^3 + b, or the (0, 0) this library reads as infinity.  The Rust
// side hands over `[G1Affine]` by reinterpreting memory (utils.rs:27-43, worker.rs:136-141) and `GroupAffine {x, y, infinity}` has no
// guaranteed field order: swapped coordinates, a shifted stride or a different padding would otherwise give commitments that are
// garbage with PLONK_OK.  One lane per point, 3 saturated Montgomery products: ~1 ms at 2^24 points, once per init.
template <int NQ>
__global__ void __launch_bounds__(256) bases_check_kernel(const AffPt<NQ>* __restrict__ pts, uint64_t n, const FpParams<NQ> P, const Fp<NQ> b_mont,
                                                          unsigned long long* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AffPt<NQ> a = load16(pts + i);
    if (aff_is_inf(a)) return;
    uint32_t why = 0;
    bool xr = false, yr = false;                               // < p ?
    for (int k = NQ - 1; k >= 0; k--) { if (a.x.l[k] != P.p[k]) { xr = a.x.l[k] < P.p[k]; break; } }
    for (int k = NQ - 1; k >= 0; k--) { if (a.y.l[k] != P.p[k]) { yr = a.y.l[k] < P.p[k]; break; } }
    if (!xr || !yr) why = BASES_BAD_RANGE;
    else if (!fp_eq(fp_sqr(a.y, P), fp_add(fp_mul(fp_sqr(a.x, P), a.x, P), b_mont, P))) why = BASES_BAD_CURVE;
    if (why) {
        atomicMin(bad, (unsigned long long)i);
        atomicOr(bad + 1, (unsigned long long)why);
    }
}

// d_bad: two device words, zeroed to {~0, 0} here unless `keep` (the ark conversion has already written its findings)
template <int NQ> static int bases_check_t(int curve, const void* d_xy, size_t n, unsigned long long* d_bad, bool keep, const char* who, hipStream_t stream) {
    const FpParams<NQ>& P = fq_params<NQ>(curve);
    if (!keep) {
        static const unsigned long long init[2] = {~0ull, 0ull};
        HIP_TRY(hipMemcpyAsync(d_bad, init, sizeof init, hipMemcpyHostToDevice, stream));
    }
    Fp<NQ> b = fp_zero<NQ>(), one;
    for (int k = 0; k < NQ; k++) one.l[k] = P.one[k];
    for (int k = 0; k < (curve == PLONK_BN254 ? 3 : 4); k++) b = fp_add(b, one, P);          // y^2 = x^3 + 3 (BN254) / + 4 (BLS12-381), Montgomery form
    if (n) hipLaunchKernelGGL(bases_check_kernel<NQ>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, (const AffPt<NQ>*)d_xy, (uint64_t)n, P, b, d_bad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "bases_check launch: %s", hipGetErrorString(e));
    unsigned long long h[2] = {~0ull, 0ull};
    HIP_TRY(hipMemcpyAsync(h, d_bad, sizeof h, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (h[1] == 0) return PLONK_OK;
    return plonk_fail(PLONK_ERR_ARG, "%s: base %llu of %zu is not a curve point (%s%s%s) — wrong layout, field order or stride?  (PLONK_BASES_XY: x||y, "
                      "PLONK_BASES_ARK: x, y, infinity byte; Montgomery limbs; plonk_set_option(\"check_bases\", 0) skips this check)", who, h[0], n,
                      (h[1] & BASES_BAD_FLAG) ? "infinity flag byte is neither 0 nor 1; " : "", (h[1] & BASES_BAD_RANGE) ? "coordinate not below the modulus; " : "",
                      (h[1] & BASES_BAD_CURVE) ? "y^2 != x^3 + b" : "");
}
int bases_check(int curve, const void* d_xy, size_t n, unsigned long long* d_bad, bool keep, const char* who, hipStream_t stream) {
    return curve == PLONK_BN254 ? bases_check_t<8>(curve, d_xy, n, d_bad, keep, who, stream) : bases_check_t<12>(curve, d_xy, n, d_bad, keep, who, stream);
}

template <int NQ> static const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B>& fl_params(int curve) {
    static const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P = fl_make_params<LimbGeom<NQ>::NL, LimbGeom<NQ>::B, NQ>(fq_params<NQ>(curve));
    return P;
}

size_t msm_limb_base_bytes(int curve) { return curve == PLONK_BN254 ? sizeof(BaseRec<8>) : sizeof(BaseRec<12>); }

// XY (reference layout) -> resident limb form; d_out holds n * msm_limb_base_bytes(curve) bytes
int bases_to_limbs(int curve, const void* d_xy, size_t n, void* d_out, hipStream_t stream) {
    if (n == 0) return PLONK_OK;
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    if (curve == PLONK_BN254)
        hipLaunchKernelGGL(bases_to_limbs_kernel<8>, dim3(grid), dim3(256), 0, stream, (const AffPt<8>*)d_xy, (uint64_t)n, (BaseRec<8>*)d_out, fl_params<8>(curve));
    else
        hipLaunchKernelGGL(bases_to_limbs_kernel<12>, dim3(grid), dim3(256), 0, stream, (const AffPt<12>*)d_xy, (uint64_t)n, (BaseRec<12>*)d_out, fl_params<12>(curve));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "bases_to_limbs launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- fixed-base window table
// The SRS is fixed between `init` calls (worker.rs:141), so shifted copies of it can be built once and kept resident: plane t holds
// 2^(c*G*t) * P_i (T planes x 64 / 96 B per point; 288 GB of HBM make this cheap).  Window w = g + t*G of a scalar then reads its bases from
// plane t and adds into the bucket set of window g: a scalar vector needs G bucket sets instead of W — the reduction pyramid, the
// bucket ordering and the host fold shrink W/G-fold (G = 1: one set for all windows), and with fewer, fuller buckets a wider window pays.
// Lane i walks its point through the T-1 shifts: c*G doublings (lazy XYZZ), one Fermat inversion, back to the canonical affine limb form
// the accumulate kernel reads.
// HISTORY: rounds 1-2 measured the merged accumulation 34 % slower per addition (23.8 vs 17.8 ms) and blamed the gathers from a 15.7 GB
// table.  Round 3 found the cause elsewhere (profiles/r03_msm_table_experiment.txt): the kernel ran one loop PER SEGMENT, and a wave's lanes
// are matched by their buckets' TOTAL size, so every segment ran as long as the wave's longest — +80 % at 2^22 points (3.9 GB table), +38 % at 2^24.
// The accumulate kernel now runs one loop over all entries of a bucket, and bases gathered from 4.8 GB (2^26 points) cost what 1.2 GB costs.
// What is left (+8 % per addition at 4 planes, +15 % at 13: the segment boundaries inside the loop) cancels what the smaller pyramid saves: OFF by
// default (`msm_precompute` = 0), kept as an option with its tests (tests/test_gpu_msm_table.py).
template <int NQ>
__global__ void __launch_bounds__(64) msm_table_kernel(BaseRec<NQ>* __restrict__ table, uint64_t n, uint64_t stride, int shift, int T,
                                                       const FLParams<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> P,
                                                       const FL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> pm2 /* p - 2, limb form */) {
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    AffL<NL, B> q = load_base<NQ>(table + i);
    for (int w = 1; w < T; w++) {
        if (!affl_is_inf(q)) {
            XyzzL<NL, B> a = xyzzl_dbl_affine(q, P);
            for (int k = 1; k < shift; k++) a = xyzzl_dbl(a, P);
            // 1 / ZZZ by Fermat; 1/Z = ZZ / ZZZ; x = X / Z^2, y = Y / ZZZ   (an infinity gives 0 -> (0, 0) = infinity)
            FL<NL, B> inv = fl_load_const<NL, B>(P.one);
            for (int bit = NL * B - 1; bit >= 0; bit--) {
                inv = fl_sqr(inv, P);
                if ((pm2.l[bit / B] >> (bit % B)) & 1) inv = fl_mul(inv, a.zzz, P);
            }
            const FL<NL, B> u = fl_mul(a.zz, inv, P);
            q.x = fl_canon_lt2p(fl_mul(a.x, fl_sqr(u, P), P), P);
            q.y = fl_canon_lt2p(fl_mul(a.y, inv, P), P);
        }
        store_base<NQ>(table + (uint64_t)w * stride + i, q);
    }
}

// ---------------------------------------------------------------------------------------------- host orchestration
// level-1 partition bits of the sort for 2^cb buckets per window: 2^10 partitions (16-entry runs per 16384-entry slice, two
// level-1 workgroups per CU), more only when a level-2 workgroup would otherwise own more than 2^11 buckets.
static bool sort_geometry(int cb, size_t n, int* lp_out) {
    const int lp = std::max(std::min(cb, 10), cb - 11);
    if (lp > 13 || n > ((size_t)1 << 27)) return false;        // 2^13 + 1 level-1 bins; <= 8192 slice offsets in the level-2 LDS
    *lp_out = lp;
    return true;
}

#define MSM_TABLE_MAX_C 21                           // widest window the table plan considers (the level-2 sort is staged in LDS up to 2^10 buckets per partition ...)
static double g_reduce_cost = 2.7 * 3300.0;           // VALU instructions per bucket in the reduction pyramid (round 3: the x2 for its low occupancy went
                                                      // with the split per-level sums and the LDS-staged level-2 sort: c = 20 now wins at 2^24 points)
// measured issue cost (profiles/r01_pmc_sq_2p24.json): ~2480 VALU instructions per mixed addition, ~3300 per full
// addition; the reduction pyramid does ~2.7 full additions per bucket
static bool window_usable(size_t n, int bits, int c) {
    int lp;
    if (!sort_geometry(c - 1, n, &lp)) return false;
    const int W = (bits + 1 + c - 1) / c;
    const int top_bits = bits - (W - 1) * c;                   // entropy of the last window's digit
    return !(W > 1 && top_bits < std::min(c - 3, 8));          // a near-empty top window puts every point in a few buckets
}
// G = bucket sets per scalar vector (plain: G = W)
static double window_cost(size_t n, int bits, int c, int G) {
    const int W = (bits + 1 + c - 1) / c;
    const double sets = (double)std::min(std::max(G, 1), W);
    // one lane per bucket: below 4 waves per SIMD (262144 lanes) the additions are latency-bound and the SIMDs idle in
    // proportion (measured: 2^16 points at c = 8 -> 4096 lanes, 10 ms; the same points at c = 15 -> 0.3 ms)
    const double lanes = sets * (double)((size_t)1 << (c - 1));
    const double par = std::min(1.0, lanes / 262144.0);
    return (double)W * (double)n * 2480.0 / par + sets * (double)((size_t)1 << (c - 1)) * g_reduce_cost;
}
static int choose_window(size_t n, int bits, double* cost_out = nullptr) {
    double best = 1e300;
    int bc = 4;
    for (int c = 4; c <= 20; c++) {
        if (!window_usable(n, bits, c)) continue;
        const double cost = window_cost(n, bits, c, (bits + 1 + c - 1) / c);
        if (cost < best) { best = cost; bc = c; }
    }
    if (cost_out) *cost_out = best;
    return bc;
}

// Plan of the fixed-base table for an SRS of n points: window width c (0 = no table: too small to pay off, or over budget), W windows per scalar,
// G bucket sets per scalar, T = ceil(W / G) planes.  mode 1: only when the cost model sees a gain; mode 2: always (tests).  The widest table the
// budget allows wins ties; force_c / force_sets (options "msm_table_c" / "msm_table_sets") pin a shape (tests, experiments).
int msm_table_plan(int curve, size_t n, int mode, size_t budget_bytes, int force_c, int force_sets, int* W_out, int* G_out, int* T_out) {
    *W_out = 1; *G_out = 1; *T_out = 1;
    if (mode == 0 || n == 0) return 0;
    const int bits = fr_params(curve).bits;
    const double pb = (double)msm_limb_base_bytes(curve);
    double plain = 0;
    choose_window(n, bits, &plain);
    const int fc = force_c > 0 ? force_c : (getenv("PLONK_MSM_TAB_C") ? atoi(getenv("PLONK_MSM_TAB_C")) : 0);       // "msm_table_c" / "msm_table_sets" options;
    const int fg = force_sets > 0 ? force_sets : (getenv("PLONK_MSM_TAB_G") ? atoi(getenv("PLONK_MSM_TAB_G")) : 0);   // the variables serve tools that cannot reach the context
    double best = 1e300;
    int bc = 0, bG = 1;
    for (int c = 4; c <= MSM_TABLE_MAX_C; c++) {
        if (fc > 0 && c != fc) continue;
        if (!window_usable(n, bits, c)) continue;
        const int W = (bits + 1 + c - 1) / c;
        if (W < 2) continue;
        for (int G = 1; G < W; G++) {                            // G = W would be the plain plan
            if (fg > 0 && G != fg) continue;
            const int T = (W + G - 1) / G;
            if (fg <= 0 && G != (W + T - 1) / T) continue;       // skip shapes that a table of the same size serves with fewer sets
            if ((double)T * (double)n * pb > (double)budget_bytes) continue;
            const double cost = window_cost(n, bits, c, G) * (1.0 + 1e-4 * T);      // ties: the smaller table
            if (cost < best) { best = cost; bc = c; bG = G; }
        }
    }
    if (!bc) return 0;
    if (mode == 1 && (n < ((size_t)1 << 16) || best > 0.97 * plain)) return 0;
    const int W = (bits + 1 + bc - 1) / bc;
    *W_out = W; *G_out = bG; *T_out = (W + bG - 1) / bG;
    return bc;
}

template <int NQ> static int msm_table_build_t(int curve, void* d_table, size_t n, size_t stride, int shift, int T, hipStream_t stream) {
    constexpr int NL = LimbGeom<NQ>::NL, B = LimbGeom<NQ>::B;
    const FpParams<NQ>& P = fq_params<NQ>(curve);
    Fp<NQ> pm2;
    uint64_t br = 2;
    for (int i = 0; i < NQ; i++) { uint64_t t = (uint64_t)P.p[i] - br; pm2.l[i] = (uint32_t)t; br = (t >> 32) & 1; }
    hipLaunchKernelGGL(msm_table_kernel<NQ>, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, stream, (BaseRec<NQ>*)d_table, (uint64_t)n, (uint64_t)stride, shift, T,
                       fl_params<NQ>(curve), fl_from_sat<NL, B, NQ>(pm2));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "msm_table launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}
// d_table: T planes of `stride` points, plane 0 already holds the bases (bases_to_limbs); plane t = 2^(shift * t) * plane 0, shift = c * G
int msm_table_build(int curve, void* d_table, size_t n, size_t stride, int shift, int T, hipStream_t stream) {
    if (n == 0 || T < 2) return PLONK_OK;
    if (curve == PLONK_BN254) return msm_table_build_t<8>(curve, d_table, n, stride, shift, T, stream);
    return msm_table_build_t<12>(curve, d_table, n, stride, shift, T, stream);
}

void msm_ws_release(MsmWorkspace& ws) {
    if (ws.d_buf) (void)hipFree(ws.d_buf);
    ws.d_buf = nullptr; ws.bytes = 0;
}

static int ensure_ws(MsmWorkspace& ws, size_t bytes) {
    if (ws.bytes >= bytes) return PLONK_OK;
    if (ws.d_buf) (void)hipFree(ws.d_buf);
    ws.d_buf = nullptr; ws.bytes = 0;
    HIP_TRY(hipMalloc(&ws.d_buf, bytes));
    ws.bytes = bytes;
    return PLONK_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }


// Window width and bucket sets per scalar vector for an MSM over n points: the table's shape when one is resident and beats the best plain plan
// for THIS n (small sub-ranges and forced windows do not use it), else the plain plan (G = W1).
static int plan_window(size_t n, int bits, int window_bits, const MsmTable& tab, int* G_out) {
    if (window_bits <= 0 && tab.c > 0 && window_usable(n, bits, tab.c)) {
        double plain = 0;
        choose_window(n, bits, &plain);
        if (tab.force || window_cost(n, bits, tab.c, tab.G) < plain) { *G_out = tab.G; return tab.c; }
    }
    const int c = window_bits > 0 ? std::min(std::max(window_bits, 2), 20) : choose_window(n, bits);
    *G_out = (bits + 1 + c - 1) / c;
    return c;
}

// K >= 1 scalar vectors against the SAME bases in one set of launches (the independent commitments of a prover round): vector k
// supplies the windows k*W1 .. (k+1)*W1 - 1 of one big (window, bucket) problem, so the sort, the bucket accumulation and the
// reduction pyramid each run once over K times the work — no per-MSM launch gaps, wave tails or host round trips, which is what
// bounds small MSMs (2^20 - 2^21 points per rank / per configs[1]).  lens[k] <= n valid scalars in vector k.
template <int NQ>
static int msm_slice(int curve, const BaseRec<NQ>* d_bases, const uint32_t* const* d_scalars, const size_t* lens, int K, bool scalars_mont,
                     size_t n, XyzzPt<NQ>* h_result, MsmWorkspace& ws, int window_bits, const MsmTable& tab, hipStream_t stream) {
    const FpParams<NQ>& P = fq_params<NQ>(curve);
    const int bits = fr_params(curve).bits;
    int G = 1;
    const int c = plan_window(n, bits, window_bits, tab, &G);
    const int W1 = (bits + 1 + c - 1) / c;             // windows per scalar vector (signed digits: one spare bit for the last carry)
    const bool merged = G < W1;                        // fixed-base table in use: G bucket sets per vector, window g + t*G reads plane t
    if (merged && (W1 != tab.W || G != tab.G)) return plonk_fail(PLONK_ERR_STATE, "msm: table built for %d windows in %d sets, plan has %d in %d", tab.W, tab.G, W1, G);
    const int W = K * W1;                              // windows of the whole batch
    const int cb = c - 1;                              // 2^(c-1) buckets per window
    const uint64_t nb = (uint64_t)1 << cb;
    const int Wr = K * G;                              // bucket sets to accumulate and reduce
    const uint64_t nsub = (uint64_t)W * nb;            // (window, bucket) segments the sort produces
    const uint64_t nbuckets = (uint64_t)Wr * nb;       // accumulators
    if (nbuckets >= 0xffffffffull) return plonk_fail(PLONK_ERR_ARG, "msm: %llu buckets", (unsigned long long)nbuckets);
    const uint64_t avg = ((uint64_t)n * W) / nbuckets + 1;
    const uint32_t heavy_thresh = (uint32_t)std::max<uint64_t>(HEAVY_BUCKET, 4 * avg);
    SetGeom sg;
    sg.cb = (uint32_t)cb; sg.G = (uint32_t)G; sg.W1 = (uint32_t)W1; sg.tab_stride = merged ? tab.stride : 0;
    sg.bin_shift = 0;
    while ((avg >> sg.bin_shift) > 96) sg.bin_shift++;        // keep the average bucket inside the 256 size bins
    if ((uint64_t)n * W >= 0xffffffffull) return plonk_fail(PLONK_ERR_ARG, "msm slice too large");
    SortGeom g;
    g.n = n; g.W = W; g.cb = cb; g.stage_cap = 0;
    if (!sort_geometry(cb, n, &g.lp)) return plonk_fail(PLONK_ERR_ARG, "msm: window %d too wide for %zu points", c, n);
    g.low_bits = cb - g.lp;
    g.nblk = (uint32_t)((n + SORT_SLICE - 1) / SORT_SLICE);
    g.nreal = (uint32_t)W << g.lp;
    {
        uint32_t ln = 1;
        while (((uint64_t)1 << ln) < (uint64_t)n) ln++;
        g.idx_bits = ((uint32_t)g.low_bits + 1 + ln <= 32) ? ln : 0;
    }
    const uint64_t nhist = ((uint64_t)g.nreal + W) * g.nblk;
    const uint64_t nscan_blocks = (nhist + SCAN_CHUNK - 1) / SCAN_CHUNK;

    size_t off = 0;
    const size_t o_dig = off; off = align_up(off + (size_t)n * W * 4, 256);
    const size_t o_tmp = off; off = align_up(off + (size_t)n * W * 4, 256);
    const size_t o_hist = off; off = align_up(off + (nhist + 1) * 4, 256);
    const size_t o_hoff = off; off = align_up(off + (nhist + 1) * 4, 256);
    const size_t o_offsets = off; off = align_up(off + (nsub + 1) * 4, 256);
    const size_t o_bsums = off; off = align_up(off + (nscan_blocks + 1) * 4, 256);
    const size_t o_sorted = off; off = align_up(off + (size_t)n * W * 4, 256);
    const size_t o_order = off; off = align_up(off + nbuckets * 4, 256);
    const size_t o_redo = off; off = align_up(off + (nbuckets + 1) * 4, 256);
    const size_t o_szh = off; off = align_up(off + 2 * SIZE_BINS * 4, 256);
    const size_t o_work = off; off = align_up(off + 4, 256);       // work counter of the persistent accumulation
    const uint64_t max_heavy = ((uint64_t)n * W) / heavy_thresh + 1;     // a bucket is heavy only above heavy_thresh entries
    const size_t o_heavy = off; off = align_up(off + (max_heavy + 1) * 4, 256);
    typedef XyzzL<LimbGeom<NQ>::NL, LimbGeom<NQ>::B> BucketL;
    // reduction pyramid geometry
    int nlev = 0;
    uint64_t lev_n[16], lev_k[16], lev_nch[16];
    for (uint64_t cur = nb; cur > 1 && nlev < 15; nlev++) {
        lev_n[nlev] = cur;
        lev_k[nlev] = std::min<uint64_t>(REDUCE_K, cur);
        lev_nch[nlev] = cur / lev_k[nlev];
        cur = lev_nch[nlev];
    }
    uint64_t pyr = 0;
    for (int l = 0; l < nlev; l++) pyr += (uint64_t)Wr * lev_nch[l];
    const size_t o_buckets = off; off = align_up(off + nbuckets * sizeof(BucketL), 256);
    const size_t o_acc = off; off = align_up(off + pyr * sizeof(BucketL), 256);
    const size_t o_sarr = off; off = align_up(off + pyr * sizeof(BucketL), 256);
    const uint32_t nsplit = 1;
    // plain mode: the per-(level, window) sums of large pyramids are split over `gsplit` workgroups and folded by a second launch
    // (one workgroup per 2048 level-0 values: at 2^13 values per window — c = 16, the 2^20-point plan — a single workgroup walked 32 points per lane
    //  before its 8-step tree, the deepest dependent chain of a small MSM's reduction)
    const uint32_t gsplit = (uint32_t)std::min<uint64_t>(32, std::max<uint64_t>(1, (nlev ? lev_nch[0] : 1) / 2048));
    const size_t o_wsum = off; off = align_up(off + (size_t)Wr * (nlev + 1) * nsplit * sizeof(XyzzPt<NQ>), 256);
    const size_t o_wpart = off; off = align_up(off + (gsplit > 1 ? (size_t)Wr * (nlev + 1) * gsplit * sizeof(BucketL) : 0), 256);
    const size_t o_hpart = off; off = align_up(off + max_heavy * HEAVY_SEGS * sizeof(BucketL), 256);
    // "msm_reduce_grid": buckets of a set as an H x L grid (5b); row / column sums in the limb form, then logL + 1 + logH bit sums per set
    const bool grid_mode = ws.reduce_grid != 0 && cb >= 2;
    const uint32_t g_logL = (uint32_t)(cb + 1) / 2, g_logH = (uint32_t)cb - g_logL, g_nbits = g_logL + 1 + g_logH;
    const size_t o_grc = off; off = align_up(off + (grid_mode ? (size_t)Wr * (((size_t)1 << g_logL) + ((size_t)1 << g_logH)) * sizeof(BucketL) : 0), 256);
    const size_t o_gbits = off; off = align_up(off + (grid_mode ? (size_t)Wr * g_nbits * sizeof(XyzzPt<NQ>) : 0), 256);
    int rc = ensure_ws(ws, off);
    if (rc) return rc;
    char* base = (char*)ws.d_buf;
    uint32_t* dig = (uint32_t*)(base + o_dig);
    uint32_t* tmp = (uint32_t*)(base + o_tmp);
    uint32_t* blk_hist = (uint32_t*)(base + o_hist);
    uint32_t* blk_off = (uint32_t*)(base + o_hoff);
    uint32_t* offsets = (uint32_t*)(base + o_offsets);
    uint32_t* bsums = (uint32_t*)(base + o_bsums);
    uint32_t* sorted = (uint32_t*)(base + o_sorted);
    uint32_t* order = (uint32_t*)(base + o_order);
    uint32_t* redo = (uint32_t*)(base + o_redo);            // [0] = count, [1..] = list
    uint32_t* heavy = (uint32_t*)(base + o_heavy);          // [0] = count, [1..] = list
    uint32_t* ghist = (uint32_t*)(base + o_szh);
    uint32_t* bin_cursor = ghist + SIZE_BINS;
    BucketL* buckets = (BucketL*)(base + o_buckets);
    BucketL* acc_arr = (BucketL*)(base + o_acc);
    BucketL* s_arr = (BucketL*)(base + o_sarr);
    XyzzPt<NQ>* wsum = (XyzzPt<NQ>*)(base + o_wsum);
    BucketL* hpart = (BucketL*)(base + o_hpart);
    BucketL* wpart = (BucketL*)(base + o_wpart);
    BucketL* grid_r = (BucketL*)(base + o_grc);
    BucketL* grid_c = grid_r + ((size_t)Wr << g_logL);
    XyzzPt<NQ>* gbits = (XyzzPt<NQ>*)(base + o_gbits);

    const size_t lds1 = ((size_t)(1u << g.lp) + 1) * 4;
    { ProfScope ps("msm_digits_kernel", stream);
    for (int k = 0; k < K; k++)
        hipLaunchKernelGGL(msm_digits_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_scalars[k], (uint64_t)n, (uint64_t)lens[k], c, W1,
                           dig + (size_t)k * W1 * n, scalars_mont ? 1 : 0, fr_params(curve)); }
    // the bucket-size histogram rides in the level-2 sort when a bucket is one sorted segment (no fixed-base table).  "msm_fused_order": 1 (default) = for
    // launches of >= 2^23 points in total, where it takes 3 % off the commitment phase of a 2^24 step; below that the level-2 sort loses more to the
    // histogram's atomics than the two saved launches give back (+0.7 % per 2^20 step; profiles/r05_msm_fused_order.txt); 2 = always (tests); 0 = never
    const bool fused_order = (ws.fused_order == 2 || (ws.fused_order == 1 && (uint64_t)n * (uint64_t)K >= ((uint64_t)1 << 23))) && sg.tab_stride == 0 && sg.G == sg.W1;
    const SizeHist szh{fused_order ? ghist : nullptr, sg.bin_shift, (uint64_t)nbuckets};
    if (fused_order) HIP_TRY(hipMemsetAsync(ghist, 0, 2 * SIZE_BINS * 4, stream));
    { ProfScope ps("msm_sort", stream);
    hipLaunchKernelGGL(sort_hist_kernel, dim3(g.nblk, W), dim3(256), lds1, stream, dig, g, blk_hist);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((uint32_t)nscan_blocks), dim3(SCAN_THREADS), 0, stream, blk_hist, nhist, bsums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, stream, bsums, nscan_blocks, blk_off + nhist);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((uint32_t)nscan_blocks), dim3(SCAN_THREADS), 0, stream, blk_hist, nhist, bsums, blk_off);
    {
        static DeviceOnce attr;
        HIP_TRY(attr.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }));
    }
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(g.nblk, W), dim3(SCATTER_THREADS), (2 * ((size_t)(1u << g.lp) + 1) + 1 + SORT_SLICE) * 4, stream, dig, g,
                       blk_off, tmp);
    // the staged level-2 kernel: 78 KiB of LDS per workgroup (two per CU); what the counters and the slice starts leave is the staging buffer
    const size_t lds_fixed = (((size_t)1 << g.low_bits) + (g.idx_bits ? 0 : g.nblk)) * 4;
    const size_t lds_staged = 78 * 1024;
    g.stage_cap = lds_fixed + 4096 * 4 <= lds_staged ? (uint32_t)((lds_staged - lds_fixed) / 4) : 0;
    if (ws.sort_stage_cap > 0 && g.stage_cap) g.stage_cap = std::min<uint32_t>(g.stage_cap, std::max<uint32_t>((uint32_t)ws.sort_stage_cap, STAGE_THREADS));   // tests: force the chunked path
    if (g.low_bits >= 8 && g.stage_cap >= STAGE_THREADS && !getenv("PLONK_MSM_NO_STAGED_SORT")) {
        static DeviceOnce attr2;
        HIP_TRY(attr2.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_partition_staged_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }));
        hipLaunchKernelGGL(sort_partition_staged_kernel, dim3(g.nreal), dim3(STAGE_THREADS), lds_staged, stream, tmp, g, blk_off, sorted, offsets, szh);
    } else {
        hipLaunchKernelGGL(sort_partition_kernel, dim3(g.nreal), dim3(256), (((size_t)1 << g.low_bits) + g.nblk) * 4, stream, tmp, g, blk_off, sorted, offsets, szh);
    } }
    { ProfScope ps("msm_bucket_order", stream);
    HIP_TRY(hipMemsetAsync(redo, 0, 4, stream));
    HIP_TRY(hipMemsetAsync(heavy, 0, 4, stream));
    const uint32_t bgrid = (uint32_t)((nbuckets + 255) / 256);
    if (fused_order) {
        hipLaunchKernelGGL(bucket_size_place_kernel, dim3(bgrid), dim3(256), 0, stream, offsets, nbuckets, sg, bin_cursor, order, (const uint32_t*)ghist);
    } else {
        HIP_TRY(hipMemsetAsync(ghist, 0, 2 * SIZE_BINS * 4, stream));
        hipLaunchKernelGGL(bucket_size_hist_kernel, dim3(bgrid), dim3(256), 0, stream, offsets, nbuckets, sg, ghist);
        hipLaunchKernelGGL(bucket_size_scan_kernel, dim3(1), dim3(SIZE_BINS), 0, stream, ghist, bin_cursor);
        hipLaunchKernelGGL(bucket_size_place_kernel, dim3(bgrid), dim3(256), 0, stream, offsets, nbuckets, sg, bin_cursor, order, (const uint32_t*)nullptr);
    } }
    // "msm_acc_persist" (default 4): the accumulation as that many workgroups per CU of persistent waves; 0 = one lane per bucket over the whole grid
    uint32_t* work = nullptr;
    uint32_t acc_grid = (uint32_t)((nbuckets + 255) / 256);
    if (ws.acc_persist != 0) {
        // CU count of the CURRENT device, cached per device (a process may drive several GPUs from several threads)
        static int cu_of_dev[64];
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        int n_cu = dev >= 0 && dev < 64 ? __atomic_load_n(&cu_of_dev[dev], __ATOMIC_ACQUIRE) : 0;
        if (!n_cu) {
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            n_cu = std::max(prop.multiProcessorCount, 1);
            if (dev >= 0 && dev < 64) __atomic_store_n(&cu_of_dev[dev], n_cu, __ATOMIC_RELEASE);
        }
        const uint32_t pgrid = ws.acc_persist > 0 ? (uint32_t)n_cu * (uint32_t)ws.acc_persist : (uint32_t)(-ws.acc_persist);   // < 0: absolute grid (tests)
        if (pgrid < acc_grid) {                                   // small problems keep the plain grid
            work = (uint32_t*)(base + o_work);
            HIP_TRY(hipMemsetAsync(work, 0, 4, stream));
            acc_grid = pgrid;
        }
    }
    { ProfScope ps("msm_accumulate_kernel", stream);
    if (ws.fused_y3)
        hipLaunchKernelGGL((msm_accumulate_kernel<NQ, true>), dim3(acc_grid), dim3(256), 0, stream, d_bases, sorted, offsets, order,
                           nbuckets, sg, heavy_thresh, buckets, redo, redo + 1, heavy, heavy + 1, work, fl_params<NQ>(curve));
    else
        hipLaunchKernelGGL((msm_accumulate_kernel<NQ, false>), dim3(acc_grid), dim3(256), 0, stream, d_bases, sorted, offsets, order,
                           nbuckets, sg, heavy_thresh, buckets, redo, redo + 1, heavy, heavy + 1, work, fl_params<NQ>(curve)); }
    { ProfScope ps("msm_heavy", stream);
    hipLaunchKernelGGL(msm_heavy_kernel<NQ>, dim3(128, HEAVY_SEGS), dim3(256), 256 * sizeof(BucketL), stream, d_bases, sorted, offsets, sg,
                       heavy, heavy + 1, hpart, fl_params<NQ>(curve));
    hipLaunchKernelGGL(msm_heavy_finish_kernel<NQ>, dim3(16), dim3(64), 0, stream, heavy, heavy + 1, hpart, buckets, fl_params<NQ>(curve)); }
    { ProfScope ps("msm_accumulate_redo_kernel", stream);
    hipLaunchKernelGGL(msm_accumulate_redo_kernel<NQ>, dim3(256), dim3(64), 0, stream, d_bases, sorted, offsets, sg, buckets, redo,
                       redo + 1, fl_params<NQ>(curve)); }
    if (grid_mode) {
        ProfScope ps("msm_reduce", stream);
        // segments per output: 64 interleaved segments keep the serial part shortest; problems with lanes to spare trade a tree level for a longer
        // serial part (T up to 8 inputs per lane) while the launch still has >= 2^18 lanes — fewer LDS tree additions per bucket
        auto pick_seg = [&](uint32_t logY) {
            uint32_t ls = std::min<uint32_t>(logY, 6);
            while (ls > 3 && (logY - ls) < 3 && (((uint64_t)Wr << cb) >> (logY - ls + 1)) >= ((uint64_t)1 << 18)) ls--;
            return ls;
        };
        const uint32_t log_segc = pick_seg(g_logH), log_segr = pick_seg(g_logL);
        const uint32_t cwc = 256u >> log_segc, cwr = 256u >> log_segr;
        const uint32_t colblocks = (uint32_t)((((uint64_t)1 << g_logL) + cwc - 1) / cwc), rowblocks = (uint32_t)((((uint64_t)1 << g_logH) + cwr - 1) / cwr);
        hipLaunchKernelGGL(msm_grid_sums_kernel<NQ>, dim3(colblocks + rowblocks, Wr), dim3(256), 256 * sizeof(BucketL), stream, buckets, g_logL, g_logH, log_segc, log_segr,
                           colblocks, grid_r, grid_c, fl_params<NQ>(curve));
        hipLaunchKernelGGL(msm_bit_sums_kernel<NQ>, dim3(g_nbits, Wr), dim3(256), 256 * sizeof(BucketL), stream, grid_r, grid_c, g_logL, g_logH, gbits, fl_params<NQ>(curve));
    } else {
        ProfScope ps("msm_reduce", stream);
        SumJobs jobs;
        memset(&jobs, 0, sizeof jobs);
        const BucketL* in = buckets;
        uint64_t at = 0;
        for (int l = 0; l < nlev; l++) {
            const uint64_t total_chunks = (uint64_t)Wr * lev_nch[l];
            HIP_TRY(hipMemsetAsync(redo, 0, 4, stream));          // the accumulate redo list is dead by now: reuse it per level
            hipLaunchKernelGGL(msm_reduce_level_kernel<NQ>, dim3((uint32_t)((total_chunks + 255) / 256)), dim3(256), 0, stream, in, lev_n[l],
                               (uint32_t)lev_k[l], lev_nch[l], total_chunks, acc_arr + at, s_arr + at, redo, redo + 1, fl_params<NQ>(curve));
            hipLaunchKernelGGL(msm_reduce_level_redo_kernel<NQ>, dim3(64), dim3(64), 0, stream, in, lev_n[l], (uint32_t)lev_k[l], lev_nch[l],
                               acc_arr + at, s_arr + at, redo, redo + 1, fl_params<NQ>(curve));
            jobs.src[l] = acc_arr + at;
            jobs.count[l] = lev_nch[l];
            in = s_arr + at;
            at += total_chunks;
        }
        jobs.src[nlev] = in;            // the single entry left per window: Sigma (nb == 1: the bucket itself)
        jobs.count[nlev] = 1;
        if (gsplit > 1) {
            hipLaunchKernelGGL((msm_points_sum_kernel<NQ, true>), dim3(nlev + 1, Wr, gsplit), dim3(256), 256 * sizeof(BucketL), stream, jobs, (XyzzPt<NQ>*)nullptr,
                               (uint32_t)(nlev + 1), gsplit, fl_params<NQ>(curve), wpart);
            SumJobs fold;
            memset(&fold, 0, sizeof fold);
            for (int l = 0; l <= nlev; l++) {
                fold.src[l] = wpart + (size_t)l * gsplit;
                fold.count[l] = gsplit;
                fold.wstride[l] = (uint64_t)(nlev + 1) * gsplit;
            }
            hipLaunchKernelGGL((msm_points_sum_kernel<NQ, false>), dim3(nlev + 1, Wr, 1), dim3(256), 256 * sizeof(BucketL), stream, fold, wsum, (uint32_t)(nlev + 1), 1u,
                               fl_params<NQ>(curve), (BucketL*)nullptr);
        } else {
            hipLaunchKernelGGL((msm_points_sum_kernel<NQ, false>), dim3(nlev + 1, Wr, nsplit), dim3(256), 256 * sizeof(BucketL), stream, jobs, wsum, (uint32_t)(nlev + 1), nsplit,
                               fl_params<NQ>(curve), (BucketL*)nullptr);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "msm launch: %s", hipGetErrorString(e));

    std::vector<XyzzPt<NQ>> h(grid_mode ? (size_t)Wr * g_nbits : (size_t)Wr * (nlev + 1) * nsplit);
    HIP_TRY(hipMemcpyAsync(h.data(), grid_mode ? gbits : wsum, h.size() * sizeof(XyzzPt<NQ>), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const int Wk = G;                                  // bucket sets per scalar vector: total = sum_g 2^(c*g) * V_(k, g)
    for (int kk = 0; kk < K; kk++) {
    XyzzPt<NQ> total = xyzz_inf<NQ>();
    for (int w = (kk + 1) * Wk - 1; w >= kk * Wk; w--) {
        if (grid_mode) {
            // V_w = sum_(b <= logL) 2^b TR_b + 2^logL * sum_(b < logH) 2^b TC_b: one Horner over the merged coefficients, from the top bit
            const XyzzPt<NQ>* tr = h.data() + (size_t)w * g_nbits;
            const XyzzPt<NQ>* tc = tr + g_logL + 1;
            XyzzPt<NQ> v = xyzz_inf<NQ>();
            for (int b = (int)(g_logL + (g_logH ? g_logH - 1 : 0)); b >= 0; b--) {
                if (!xyzz_is_inf(v)) v = xyzz_dbl(v, P);
                if (b <= (int)g_logL) v = xyzz_add(v, tr[b], P);
                if (b >= (int)g_logL && b - (int)g_logL < (int)g_logH) v = xyzz_add(v, tc[b - g_logL], P);
            }
            if (!xyzz_is_inf(total))
                for (int k = 0; k < c; k++) total = xyzz_dbl(total, P);
            total = xyzz_add(total, v, P);
            continue;
        }
        // V_w = Sigma + sum_l K^l A_l  (Horner from the top level)
        const XyzzPt<NQ>* hw = h.data() + (size_t)w * (nlev + 1) * nsplit;
        auto part = [&](int l) {                     // the partial sums of (level l, window w)
            XyzzPt<NQ> t = hw[(size_t)l * nsplit];
            for (uint32_t z = 1; z < nsplit; z++) t = xyzz_add(t, hw[(size_t)l * nsplit + z], P);
            return t;
        };
        // V_w = Sigma + sum_l nch_l A_l,  nch_l = nch_(l+1) * K_(l+1),  nch_(nlev-1) = 1:  Horner from level 0
        XyzzPt<NQ> v = xyzz_inf<NQ>();
        for (int l = 0; l < nlev; l++) {
            if (!xyzz_is_inf(v))
                for (uint64_t k = lev_k[l]; k > 1; k >>= 1) v = xyzz_dbl(v, P);
            v = xyzz_add(v, part(l), P);
        }
        v = xyzz_add(v, part(nlev), P);
        if (!xyzz_is_inf(total))
            for (int k = 0; k < c; k++) total = xyzz_dbl(total, P);
        total = xyzz_add(total, v, P);
    }
    h_result[kk] = total;
    }
    return PLONK_OK;
}


// K scalar vectors (lens[k] valid scalars each) against bases[0 .. n): out = K Jacobian points.  Vectors are batched as far as the
// 32-bit entry indices of the sort allow (n * W * K < 2^32), points beyond the slice size are sliced as before.
template <int NQ>
static int msm_run_t(int curve, const void* d_bases, const uint32_t* const* d_scalars, const size_t* lens, int K, bool scalars_mont, size_t n, uint32_t* h_out_jac,
                     MsmWorkspace& ws, int window_bits, const MsmTable& tab, hipStream_t stream) {
    const FpParams<NQ>& P = fq_params<NQ>(curve);
    std::vector<XyzzPt<NQ>> total((size_t)K, xyzz_inf<NQ>());
    const size_t SLICE = (size_t)1 << std::min(std::max(ws.slice_log, 8), 26);      // default 2^26 points per slice (workspace sizing); "msm_slice_log" option for tests
    for (size_t s = 0; s < n; s += SLICE) {
        const size_t m = std::min(SLICE, n - s);
        // windows per vector for this slice size -> how many vectors fit one launch set
        const int bits = fr_params(curve).bits;
        int G_ = 1;
        const int c = plan_window(m, bits, window_bits, tab, &G_);
        const int W1 = (bits + 1 + c - 1) / c;
        const uint64_t per = (uint64_t)m * W1;
        int group = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)K, (0xfffffff0ull / per)));
        if ((uint64_t)group * W1 > 65535) group = 65535 / W1;                    // grid.y of the per-window launches
        group = std::min(group, std::min(std::max(ws.batch_max, 1), 64));
        for (int k0 = 0; k0 < K; k0 += group) {
            const int kn = std::min(group, K - k0);
            std::vector<const uint32_t*> ptrs(kn);
            std::vector<size_t> ln(kn);
            for (int k = 0; k < kn; k++) {
                ptrs[k] = d_scalars[k0 + k] + 8 * s;
                ln[k] = lens[k0 + k] > s ? std::min(lens[k0 + k] - s, m) : 0;
            }
            std::vector<XyzzPt<NQ>> part(kn);
            int rc = msm_slice<NQ>(curve, (const BaseRec<NQ>*)d_bases + s, ptrs.data(), ln.data(), kn, scalars_mont, m, part.data(), ws,
                                   window_bits, tab, stream);
            if (rc) return rc;
            for (int k = 0; k < kn; k++) total[k0 + k] = xyzz_add(total[k0 + k], part[k], P);
        }
    }
    for (int k = 0; k < K; k++) {
        JacPt<NQ> j = jac_from_xyzz_normalised(total[k], P);
        memcpy((char*)h_out_jac + (size_t)k * sizeof j, &j, sizeof j);
    }
    return PLONK_OK;
}

int msm_run_many(int curve, const void* d_bases, const uint32_t* const* d_scalars, const size_t* lens, int K, bool scalars_mont, size_t n, uint32_t* h_out_jac,
                 MsmWorkspace& ws, int window_bits, const MsmTable& tab, hipStream_t stream) {
    if (K <= 0) return PLONK_OK;
    if (curve == PLONK_BN254) return msm_run_t<8>(curve, d_bases, d_scalars, lens, K, scalars_mont, n, h_out_jac, ws, window_bits, tab, stream);
    return msm_run_t<12>(curve, d_bases, d_scalars, lens, K, scalars_mont, n, h_out_jac, ws, window_bits, tab, stream);
}

int msm_run(int curve, const void* d_bases, const uint32_t* d_scalars, bool scalars_mont, size_t n, uint32_t* h_out_jac, MsmWorkspace& ws, int window_bits,
            const MsmTable& tab, hipStream_t stream) {
    const size_t len = n;
    return msm_run_many(curve, d_bases, &d_scalars, &len, 1, scalars_mont, n, h_out_jac, ws, window_bits, tab, stream);
}

template <int NQ> static void jac_add_host_t(int curve, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    const FpParams<NQ>& P = fq_params<NQ>(curve);
    JacPt<NQ> ja, jb;
    memcpy(&ja, a, sizeof ja); memcpy(&jb, b, sizeof jb);
    XyzzPt<NQ> s = xyzz_add(xyzz_from_jac(ja, P), xyzz_from_jac(jb, P), P);
    JacPt<NQ> r = jac_from_xyzz_normalised(s, P);
    memcpy(out, &r, sizeof r);
}
int msm_jac_add_host(int curve, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    if (curve == PLONK_BN254) jac_add_host_t<8>(curve, a, b, out); else jac_add_host_t<12>(curve, a, b, out);
    return PLONK_OK;
}
template <int NQ> static void jac_to_affine_host_t(int curve, const uint32_t* jac, uint32_t* out_xy, int* is_inf) {
    const FpParams<NQ>& P = fq_params<NQ>(curve);
    JacPt<NQ> j;
    memcpy(&j, jac, sizeof j);
    XyzzPt<NQ> x = xyzz_from_jac(j, P);
    *is_inf = xyzz_is_inf(x) ? 1 : 0;
    AffPt<NQ> a = xyzz_to_affine(x, P);
    if (*is_inf) a.y = fp_one(P);          // arkworks' affine zero is (0, 1, infinity = true)
    memcpy(out_xy, &a, sizeof a);
}
int msm_jac_to_affine_host(int curve, const uint32_t* jac, uint32_t* out_xy, int* is_inf) {
    if (curve == PLONK_BN254) jac_to_affine_host_t<8>(curve, jac, out_xy, is_inf); else jac_to_affine_host_t<12>(curve, jac, out_xy, is_inf);
    return PLONK_OK;
}
