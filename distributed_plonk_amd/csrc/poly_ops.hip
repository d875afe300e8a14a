// poly_ops.hip — the O(n) prover steps either side of the MSM/NTT hot path, device-resident:
//
//   rank 2  permutation grand product                      /root/reference/src/dispatcher2.rs:329-344
//           (per-gate numerator/denominator, division, running product — a serial host loop with one field
//           inversion per gate in the reference; here: one streaming kernel + two multiplicative scans and a
//           single inversion:  z[j] = prod_{k<j} num_k * prod_{k>=j} den_k / prod_all den_k)
//   rank 3  DensePolynomial::evaluate (Horner at zeta)     dispatcher2.rs:545-555
//           scalar * polynomial sums (lin_poly, batch_poly) dispatcher2.rs:566-633,646-649
//           synthetic division by (X - z)                   dispatcher2.rs:651-666,672-688
//           (q_{i-1} = z^-i * sum_{k>=i} c_k z^k : an additive suffix scan of scaled coefficients)
//           blinding (rand(k-1) * Z_H + poly)               dispatcher2.rs:311-312,347-348
//
// Arithmetic: HBM keeps the reference's R = 2^256 Montgomery form, fully reduced.  Products run on 9 x 29-bit limbs
// (fp29.hpp, mont(a, b) = a*b/2^261).  rep(x) = x*2^261 is the form in which products of DATA close under mont();
// an R-form datum is rep(x * 2^-5) and the stray powers of two cancel where noted.  Every value stored is
// canonical, so results are bit-identical to ark-ff's.
//
// Scans: block tile = 256 lanes x 8 consecutive elements per lane; phase 1 tile totals, phase 2 one workgroup scans
// the totals, phase 3 re-reads the tile, scans it through LDS and applies an epilogue.  All streaming; none of this
// is on the critical path of the proof-equivalent mix (a few ms at n = 2^24 against ~1 s of NTT + MSM).
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "constants.h"
#include "ntt_kernels.hpp"
#include "plonk_internal.hpp"

#define PO_LANES 256
#define PO_CH 8
#define PO_TILE (PO_LANES * PO_CH)
#define PO_TOP 1024          // lanes of the single-workgroup phase-2 kernel

struct PoCtx {
    F29Params f29;
    FrParams fp;
};

// ---------------------------------------------------------------------------------------------- scan operators
struct OpMul {               // values: rep-form field elements as F29, normalised, < 1.4 p
    typedef F29 V;
    static constexpr int WORDS = 9;
    static __device__ __forceinline__ V load(const Fr* p, const PoCtx&) { return f29_from_sat(load_fr(p)); }
    static __device__ __forceinline__ V from_raw(const Fr& r) { return f29_from_sat(r); }
    static __device__ __forceinline__ void store(Fr* p, const V& v, const PoCtx& c) { store_fr(p, f29_to_sat(f29_canon(v, c.f29))); }
    static __device__ __forceinline__ V comb(const V& a, const V& b, const PoCtx& c) { return f29_mul(a, b, c.f29); }
    static __device__ __forceinline__ V ident(const PoCtx& c) { return params_one(c.f29); }
    static __device__ __forceinline__ void to_words(uint32_t* w, const V& v) {
#pragma unroll
        for (int i = 0; i < 9; i++) w[i] = v.l[i];
    }
    static __device__ __forceinline__ V from_words(const uint32_t* w) {
        V v;
#pragma unroll
        for (int i = 0; i < 9; i++) v.l[i] = w[i];
        return v;
    }
};
struct OpAdd {               // values: canonical Fr
    typedef Fr V;
    static constexpr int WORDS = 8;
    static __device__ __forceinline__ V load(const Fr* p, const PoCtx&) { return load_fr(p); }
    static __device__ __forceinline__ V from_raw(const Fr& r) { return r; }
    static __device__ __forceinline__ void store(Fr* p, const V& v, const PoCtx&) { store_fr(p, v); }
    static __device__ __forceinline__ V comb(const V& a, const V& b, const PoCtx& c) { return fp_add(a, b, c.fp); }
    static __device__ __forceinline__ V ident(const PoCtx&) { return fp_zero<8>(); }
    static __device__ __forceinline__ void to_words(uint32_t* w, const V& v) {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = v.l[i];
    }
    static __device__ __forceinline__ V from_words(const uint32_t* w) {
        V v;
#pragma unroll
        for (int i = 0; i < 8; i++) v.l[i] = w[i];
        return v;
    }
};

// LDS rows are 9 words (odd pitch: consecutive lanes fall on distinct banks for both operators)
template <class Op, int LANES>
__device__ __forceinline__ typename Op::V block_total(const typename Op::V& mine, uint32_t (*sh)[9], const PoCtx& c) {
    const int t = threadIdx.x;
    typename Op::V v = mine;
    Op::to_words(sh[t], v);
    __syncthreads();
    for (int s = LANES / 2; s > 0; s >>= 1) {
        if (t < s) {
            v = Op::comb(v, Op::from_words(sh[t + s]), c);
            Op::to_words(sh[t], v);
        }
        __syncthreads();
    }
    v = Op::from_words(sh[0]);
    __syncthreads();
    return v;
}

// inclusive Hillis-Steele scan over the lanes of a workgroup; returns the exclusive value of this lane
template <class Op, int LANES>
__device__ __forceinline__ typename Op::V block_exclusive(const typename Op::V& mine, uint32_t (*sh)[9], const PoCtx& c, typename Op::V* total) {
    const int t = threadIdx.x;
    typename Op::V v = mine;
    Op::to_words(sh[t], v);
    __syncthreads();
    for (int d = 1; d < LANES; d <<= 1) {
        typename Op::V o;
        const bool on = t >= d;
        if (on) o = Op::from_words(sh[t - d]);
        __syncthreads();
        if (on) {
            v = Op::comb(o, v, c);
            Op::to_words(sh[t], v);
        }
        __syncthreads();
    }
    typename Op::V ex = t > 0 ? Op::from_words(sh[t - 1]) : Op::ident(c);
    if (total) *total = Op::from_words(sh[LANES - 1]);
    __syncthreads();
    return ex;
}

// logical position q of a scan <-> physical index (suffix scans run over the reversed array)
__device__ __forceinline__ uint64_t phys_index(uint64_t q, uint64_t n, bool reverse) { return reverse ? n - 1 - q : q; }

// A lane of the scan kernels owns PO_CH CONSECUTIVE elements (the scan is serial in the index).  Their 2 * PO_CH 16-byte loads are issued UP FRONT, into
// registers, through a compile-time unrolled loop: one after the other with a ~200-instruction product in between, every 64-byte sector was fetched once
// per 16 bytes used (PMC, round 5: 4x the bytes these kernels need); and an array indexed by a run-time loop counter lives in scratch memory (the
// pinned multiplier keeps hipcc from unrolling `#pragma unroll` loops around it), so the indices here are template constants.
template <int K> struct StaticFor {
    template <class F> static __device__ __forceinline__ void run(F&& f) { StaticFor<K - 1>::run(f); f(std::integral_constant<int, K - 1>()); }
};
template <> struct StaticFor<0> {
    template <class F> static __device__ __forceinline__ void run(F&&) {}
};
struct LaneChunk {
    Fr raw[PO_CH];
    __device__ __forceinline__ void load(const Fr* in, uint64_t q0, uint64_t n, int reverse) {
        StaticFor<PO_CH>::run([&](auto k) {
            constexpr int K = decltype(k)::value;
            const uint64_t q = q0 + K;
            raw[K] = q < n ? load_fr(in + phys_index(q, n, reverse)) : fp_zero<8>();
        });
    }
};

template <class Op>
__global__ void __launch_bounds__(PO_LANES) scan_totals_kernel(const Fr* __restrict__ in, Fr* __restrict__ btot, uint64_t n, int reverse, const PoCtx c) {
    __shared__ uint32_t sh[PO_LANES][9];
    const uint64_t q0 = (uint64_t)blockIdx.x * PO_TILE + (uint64_t)threadIdx.x * PO_CH;
    LaneChunk ch;
    ch.load(in, q0, n, reverse);
    typename Op::V acc = Op::ident(c);
    StaticFor<PO_CH>::run([&](auto k) {
        constexpr int K = decltype(k)::value;
        if (q0 + K < n) {
            const typename Op::V x = Op::from_raw(ch.raw[K]);
            acc = K == 0 ? x : Op::comb(acc, x, c);              // q0 < n whenever any element of the lane is valid: element 0 comes first
        }
    });
    const typename Op::V tot = block_total<Op, PO_LANES>(acc, sh, c);
    if (threadIdx.x == 0) Op::store(btot + blockIdx.x, tot, c);
}

// one workgroup: exclusive scan of the nb tile totals -> boff[b]; grand total -> *total
template <class Op>
__global__ void __launch_bounds__(PO_TOP) scan_top_kernel(const Fr* __restrict__ btot, Fr* __restrict__ boff, Fr* __restrict__ total, uint64_t nb,
                                                          const PoCtx c) {
    __shared__ uint32_t sh[PO_TOP][9];
    const uint64_t per = (nb + PO_TOP - 1) / PO_TOP;
    const uint64_t lo = (uint64_t)threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    typename Op::V acc = Op::ident(c);
    for (uint64_t b = lo; b < hi; b++) acc = (b == lo) ? Op::load(btot + b, c) : Op::comb(acc, Op::load(btot + b, c), c);
    typename Op::V tot;
    typename Op::V run = block_exclusive<Op, PO_TOP>(acc, sh, c, &tot);
    for (uint64_t b = lo; b < hi; b++) {
        Op::store(boff + b, run, c);
        run = Op::comb(run, Op::load(btot + b, c), c);
    }
    if (threadIdx.x == 0 && total) Op::store(total, tot, c);
}

// phase 3: out(q) = boff[tile] o (elements before q in the tile) [o x_q if inclusive], handed to the epilogue
template <class Op, class Epi>
__global__ void __launch_bounds__(PO_LANES) scan_apply_kernel(const Fr* __restrict__ in, const Fr* __restrict__ boff, uint64_t n, int reverse, int inclusive,
                                                              const PoCtx c, const Epi epi) {
    __shared__ uint32_t sh[PO_LANES][9];
    const uint64_t q0 = (uint64_t)blockIdx.x * PO_TILE + (uint64_t)threadIdx.x * PO_CH;
    // the lane's PO_CH elements: loaded once, up front, kept in registers across both loops (LaneChunk above; profiles/r05_poly_rows.txt has the
    // PMC history: a run-time indexed array in scratch memory, then elements read twice 16 bytes at a time — 4-5x the algorithmic traffic either way)
    LaneChunk ch;
    ch.load(in, q0, n, reverse);
    typename Op::V acc = Op::ident(c);
    StaticFor<PO_CH>::run([&](auto k) {
        constexpr int K = decltype(k)::value;
        const typename Op::V xk = q0 + K < n ? Op::from_raw(ch.raw[K]) : Op::ident(c);
        acc = K == 0 ? xk : Op::comb(acc, xk, c);
    });
    const typename Op::V ex = block_exclusive<Op, PO_LANES>(acc, sh, c, nullptr);
    typename Op::V run = Op::comb(Op::load(boff + blockIdx.x, c), ex, c);
    StaticFor<PO_CH>::run([&](auto k) {
        constexpr int K = decltype(k)::value;
        const uint64_t q = q0 + K;
        const typename Op::V xk = q < n ? Op::from_raw(ch.raw[K]) : Op::ident(c);
        if (inclusive) run = Op::comb(run, xk, c);
        if (q < n) epi(run, phys_index(q, n, reverse), c);
        if (!inclusive) run = Op::comb(run, xk, c);
    });
}

template <class Op>
struct EpiStore {
    Fr* out;
    __device__ __forceinline__ void operator()(const typename Op::V& v, uint64_t i, const PoCtx& c) const { Op::store(out + i, v, c); }
};

template <class Op, class Epi>
static int scan_run(const Fr* in, uint64_t n, bool reverse, bool inclusive, Fr* btot, Fr* boff, Fr* total, const PoCtx& c, const Epi& epi,
                    const char* name, hipStream_t stream) {
    if (n == 0) return PLONK_OK;
    const uint64_t nb = (n + PO_TILE - 1) / PO_TILE;
    ProfScope ps(name, stream);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_totals_kernel<Op>), dim3((uint32_t)nb), dim3(PO_LANES), 0, stream, in, btot, n, (int)reverse, c);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_top_kernel<Op>), dim3(1), dim3(PO_TOP), 0, stream, (const Fr*)btot, boff, total, nb, c);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_apply_kernel<Op, Epi>), dim3((uint32_t)nb), dim3(PO_LANES), 0, stream, in, (const Fr*)boff, n, (int)reverse,
                       (int)inclusive, c, epi);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "%s launch: %s", name, hipGetErrorString(e));
    return PLONK_OK;
}

static PoCtx make_ctx(const NttTables& T) {
    PoCtx c;
    c.f29 = T.fp29;
    c.fp = T.fp;
    return c;
}
static F29 host_rep(const Fr& v_mont, const FrParams& P) { return f29_const_from_mont256(v_mont, P); }   // rep(v) = v * 2^261
static Fr fr_arg(const uint64_t* p) { return fp_from_limbs<8>((const uint32_t*)p); }
static bool fr_arg_ok(const uint64_t* p, const FrParams& P) {     // canonical (< p)?
    const uint32_t* l = (const uint32_t*)p;
    for (int i = 7; i >= 0; i--) {
        if (l[i] < P.p[i]) return true;
        if (l[i] > P.p[i]) return false;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------- powers of a point
// z^i = t0[i & 1023] * t1[(i >> 10) & 1023] * t2[i >> 20]   (rep form; i < 2^30)
struct PowTab {
    const F29* t0;
    const F29* t1;
    const F29* t2;
    int levels;
};
__device__ __forceinline__ F29 pow_at(const PowTab& T, uint64_t i, const F29Params& fp) {
    F29 r = load_f29(T.t0 + (i & 1023));
    if (T.levels > 1) r = f29_mul(r, load_f29(T.t1 + ((i >> 10) & 1023)), fp);
    if (T.levels > 2) r = f29_mul(r, load_f29(T.t2 + (i >> 20)), fp);
    return r;
}
#define POWTAB_BYTES (3 * 1024 * sizeof(F29))
// Power tables are cached per context, keyed by (point, levels, scale): the prover evaluates 25 polynomials on the same
// coset shift and divides by the same (X - zeta) repeatedly; building a table costs ~3000 host products + an upload + a sync.
#define POWTAB_CACHE_MAX 64
static int build_pow_tab(NttTables& T, const Fr& z_mont, uint64_t len, PowTab* out, hipStream_t stream, const Fr* scale_mont = nullptr) {
    const FrParams& P = T.fp;
    const int levels = len <= 1024 ? 1 : (len <= (1u << 20) ? 2 : 3);
    std::string key((const char*)z_mont.l, 32);
    key.push_back((char)levels);
    if (scale_mont) key.append((const char*)scale_mont->l, 32);
    F29* d_tab = nullptr;
    auto it = T.pow_tabs.find(key);
    if (it != T.pow_tabs.end()) {
        d_tab = it->second;
        // LRU: a hit moves the entry to the back, so the tables a call has already been handed (poly_div_linear fetches z, then
        // 1/z) are the LAST to be evicted — with FIFO order, a miss on the second could free the first while it is still in use
        auto pos = std::find(T.pow_order.begin(), T.pow_order.end(), key);
        if (pos != T.pow_order.end()) { T.pow_order.erase(pos); T.pow_order.push_back(key); }
    } else {
        std::vector<F29> h((size_t)levels * 1024);
        Fr base = z_mont;
        for (int l = 0; l < levels; l++) {
            const Fr mult = (l == 0 && scale_mont) ? *scale_mont : fp_one(P);      // level 0 carries an optional constant factor
            Fr acc = fp_one(P);
            for (int i = 0; i < 1024; i++) { h[(size_t)l * 1024 + i] = host_rep(fp_mul(acc, mult, P), P); acc = fp_mul(acc, base, P); }
            base = acc;                       // base^1024
        }
        if (T.pow_order.size() >= POWTAB_CACHE_MAX) {          // evict the least recently used; it may still be in flight on the stream
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(T.pow_tabs[T.pow_order.front()]);
            T.pow_tabs.erase(T.pow_order.front());
            T.pow_order.erase(T.pow_order.begin());
        }
        HIP_TRY(hipMalloc((void**)&d_tab, POWTAB_BYTES));
        HIP_TRY(hipMemcpyAsync(d_tab, h.data(), h.size() * sizeof(F29), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        T.pow_tabs[key] = d_tab;
        T.pow_order.push_back(key);
    }
    out->t0 = d_tab; out->t1 = d_tab + 1024; out->t2 = d_tab + 2048; out->levels = levels;
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- rank 2: permutation product
struct PermParams {
    const Fr* wire[5];
    const Fr* id;             // extended_id_permutation, 5n
    const uint64_t* idx;      // perm_i * n + perm_j, 5n
    Fr* num;                  // rep(A_j * 2^-25)
    Fr* den;                  // rep(B_j * 2^-25); den[n-1] = rep(1)
    uint64_t n;
    uint64_t j0, cnt;         // the gates [j0, j0 + cnt) of this launch (a rank's slice; the whole product: 0, n); num / den are indexed by j - j0
    F29 gamma_r;              // gamma * 2^256 (plain limbs of the Montgomery form)
    F29 beta_c;               // rep(beta)
    uint32_t* flag;           // bit 0: zero denominator, bit 1: permutation index out of range
    PoCtx c;
};

__global__ void __launch_bounds__(256) perm_terms_kernel(const PermParams P) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.cnt) return;
    const uint64_t j = P.j0 + t;
    const F29Params& fp = P.c.f29;
    if (j == P.n - 1) {                       // the reference's loop stops at n-2 (dispatcher2.rs:331)
        const Fr one = f29_to_sat(f29_canon(params_one(fp), fp));
        store_fr(P.num + t, one);
        store_fr(P.den + t, one);
        return;
    }
    F29 a, b;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint64_t e = (uint64_t)i * P.n + j;
        uint64_t pe = P.idx[e];
        if (pe >= 5 * P.n) { atomicOr(P.flag, 2u); pe = 0; }
        const F29 t = f29_add(f29_from_sat(load_fr(P.wire[i] + j)), P.gamma_r);                       // w + gamma, limbs < 2^30
        F29 s = f29_add(t, f29_mul(f29_from_sat(load_fr(P.id + e)), P.beta_c, fp));                   // + beta * id      (:337)
        F29 d = f29_add(t, f29_mul(f29_from_sat(load_fr(P.id + pe)), P.beta_c, fp));                  // + beta * id[perm] (:339-340)
        f29_norm(s);
        f29_norm(d);
        a = i == 0 ? s : f29_mul(s, a, fp);
        b = i == 0 ? d : f29_mul(d, b, fp);
    }
    const Fr ac = f29_to_sat(f29_canon(a, fp)), bc = f29_to_sat(f29_canon(b, fp));
    if (fp_is_zero(bc)) atomicOr(P.flag, 1u);
    store_fr(P.num + t, ac);
    store_fr(P.den + t, bc);
}

// kconst = (1 / D) * 2^256 as plain limbs, D given in rep form: one lane, Fermat
__global__ void fr_inv_kernel(const Fr* __restrict__ total, Fr* __restrict__ kconst, uint32_t* flag, const PoCtx c, const F29 pm2_bits, const F29 r256) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const F29Params& fp = c.f29;
    const Fr tv = load_fr(total);
    if (fp_is_zero(tv)) atomicOr(flag, 1u);
    const F29 x = f29_from_sat(tv);
    F29 inv = params_one(fp);
    for (int bit = 9 * 29 - 1; bit >= 0; bit--) {
        inv = f29_mul(inv, inv, fp);
        if ((pm2_bits.l[bit / 29] >> (bit % 29)) & 1) inv = f29_mul(inv, x, fp);
    }
    store_fr(kconst, f29_to_sat(f29_canon(f29_mul(inv, r256, fp), fp)));
}

struct EpiPermFinal {        // z[i] = PN[i] * SD[i] * kconst  -> R form, canonical
    const Fr* pn;
    const Fr* kconst;
    Fr* out;
    __device__ __forceinline__ void operator()(const F29& sd, uint64_t i, const PoCtx& c) const {
        const F29 t = f29_mul(sd, f29_from_sat(load_fr(pn + i)), c.f29);
        const F29 r = f29_mul(t, f29_from_sat(load_fr(kconst)), c.f29);
        store_fr(out + i, f29_to_sat(f29_canon(r, c.f29)));
    }
};

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
static uint64_t tiles_of(uint64_t n) { return (n + PO_TILE - 1) / PO_TILE; }

size_t perm_product_scratch_bytes(size_t n) { return 3 * align256(n * 32) + 2 * align256(tiles_of(n) * 32) + 1024; }

// d_out[t] = prod_{j0 <= k < j0 + t} num_k / den_k, t < cnt: the product vector of dispatcher2.rs:329-344 for (j0, cnt) = (0, n); a rank's
// slice of it up to the factor prod_{k < j0} for any other range (class_prover.py: cnt = slice + 1, so that the last value is the slice's total)
int perm_product_run(NttTables& T, const void* const* wires, const void* id_perm, const void* perm_idx, const uint64_t* beta, const uint64_t* gamma,
                     size_t n_all, size_t j0, size_t cnt, void* d_out, void* scratch, hipStream_t stream) {
    const FrParams& P = T.fp;
    if (n_all < 2) return plonk_fail(PLONK_ERR_ARG, "perm_product: n = %zu", n_all);
    if (cnt == 0 || j0 + cnt > n_all) return plonk_fail(PLONK_ERR_ARG, "perm_product: gates [%zu, %zu) of %zu", j0, j0 + cnt, n_all);
    const size_t n = cnt;                     // everything below the terms kernel works on the slice
    if (!fr_arg_ok(beta, P) || !fr_arg_ok(gamma, P)) return plonk_fail(PLONK_ERR_ARG, "perm_product: challenge not reduced");
    char* s = (char*)scratch;
    Fr* A = (Fr*)s; s += align256(n * 32);
    Fr* B = (Fr*)s; s += align256(n * 32);
    Fr* PN = (Fr*)s; s += align256(n * 32);
    Fr* btot = (Fr*)s; s += align256(tiles_of(n) * 32);
    Fr* boff = (Fr*)s; s += align256(tiles_of(n) * 32);
    Fr* total = (Fr*)s; s += 64;
    Fr* kconst = (Fr*)s; s += 64;
    uint32_t* flag = (uint32_t*)s;
    HIP_TRY(hipMemsetAsync(flag, 0, 4, stream));
    bool zero_den = false;                    // the product of all denominators vanished (found by the host-side inversion)
    const PoCtx c = make_ctx(T);
    PermParams q;
    memset(&q, 0, sizeof q);
    for (int i = 0; i < 5; i++) q.wire[i] = (const Fr*)wires[i];
    q.id = (const Fr*)id_perm; q.idx = (const uint64_t*)perm_idx;
    q.num = A; q.den = B; q.n = n_all; q.j0 = j0; q.cnt = cnt;
    q.gamma_r = f29_from_sat(fr_arg(gamma));
    q.beta_c = host_rep(fr_arg(beta), P);
    q.flag = flag; q.c = c;
    {
        ProfScope ps("perm_terms_kernel", stream);
        hipLaunchKernelGGL(perm_terms_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, q);
    }
    int rc = scan_run<OpMul>(A, n, false, false, btot, boff, (Fr*)nullptr, c, EpiStore<OpMul>{PN}, "perm_scan_num", stream);       // PN[j] = prod_{k<j} num
    if (rc) return rc;
    // total of the denominators, its inverse, then the suffix scan fused with the final product
    {
        const uint64_t nb = tiles_of(n);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_totals_kernel<OpMul>), dim3((uint32_t)nb), dim3(PO_LANES), 0, stream, (const Fr*)B, btot, (uint64_t)n, 1, c);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_top_kernel<OpMul>), dim3(1), dim3(PO_TOP), 0, stream, (const Fr*)btot, boff, total, nb, c);
        // 1 / (product of all denominators): ONE inversion per call.  On the host (round 5): 32 bytes down, fp_inv, 32 bytes up — ~40 us of round trip
        // against 0.41 ms of a single lane walking 520 dependent products (fr_inv_kernel, kept for reference).  `total` holds D * 2^261 as plain limbs;
        // kconst = 2^256 / D = 2^517 / total: fp_inv reads `total` as the Montgomery form of total / 2^256 and returns 2^512 / total, then five doublings.
        Fr h_total;
        HIP_TRY(hipMemcpyAsync(&h_total, total, sizeof(Fr), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        bool zero_total = fp_is_zero(h_total);
        Fr h_k = zero_total ? fp_zero<8>() : fp_inv(h_total, P);
        for (int i = 0; i < 5; i++) h_k = fp_add(h_k, h_k, P);
        HIP_TRY(hipMemcpyAsync(kconst, &h_k, sizeof(Fr), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));                                 // h_k is a stack object
        zero_den = zero_total;
        ProfScope ps("perm_scan_den_final", stream);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_apply_kernel<OpMul, EpiPermFinal>), dim3((uint32_t)nb), dim3(PO_LANES), 0, stream, (const Fr*)B, (const Fr*)boff,
                           (uint64_t)n, 1, 1, c, EpiPermFinal{PN, kconst, (Fr*)d_out});
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "perm_product launch: %s", hipGetErrorString(e));
    uint32_t h_flag = 0;
    HIP_TRY(hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (zero_den) h_flag |= 1;
    if (h_flag & 2) return plonk_fail(PLONK_ERR_ARG, "perm_product: permutation index out of range (>= 5n)");
    if (h_flag & 1) return plonk_fail(PLONK_ERR_ARG, "perm_product: zero denominator (the reference panics on this division, dispatcher2.rs:343)");
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- rank 3: evaluate
// Round 5: Horner per lane over elements PO_LANES apart (coalesced 256-bit loads), ONE product per coefficient:
//   poly(z) = sum_L z^(base + L) * H_L,   H_L = sum_k c[base + L + 256 k] * (z^256)^k   (Horner in z^256, a constant shared by all lanes)
// EV_K coefficients per lane, then one product by z^(base + L) from the power table and the block's additive tree.  (Rounds 2-4 multiplied
// every coefficient by z^i from the three-level table: 3 products per coefficient, 12.7 % of the HBM peak.)
#define EV_K 32
#define EV_TILE (PO_LANES * EV_K)
__global__ void __launch_bounds__(PO_LANES) poly_eval_kernel(const Fr* __restrict__ poly, uint64_t len, const F29 z256 /* rep(z^256) */, const PowTab pw,
                                                             Fr* __restrict__ partial, const PoCtx c) {
    __shared__ uint32_t sh[PO_LANES][9];
    const F29Params& fp = c.f29;
    const uint64_t base = (uint64_t)blockIdx.x * EV_TILE + threadIdx.x;
    F29 acc;
#pragma unroll
    for (int l = 0; l < 9; l++) acc.l[l] = 0;
    bool any = false;
#pragma unroll 4
    for (int k = EV_K - 1; k >= 0; k--) {
        const uint64_t i = base + (uint64_t)k * PO_LANES;
        if (i < len) {
            const F29 ci = f29_from_sat(load_fr(poly + i));
            acc = any ? f29_add(f29_mul(acc, z256, fp), ci) : ci;       // < 1.36 p + p, limbs < 2^30: inside f29_mul's operand range
            any = true;
        }
    }
    Fr mine = fp_zero<8>();
    if (any) mine = f29_to_sat(f29_canon(f29_mul(acc, pow_at(pw, base, fp), fp), fp));        // * z^(base + L), R form, canonical
    const Fr tot = block_total<OpAdd, PO_LANES>(mine, sh, c);
    if (threadIdx.x == 0) store_fr(partial + blockIdx.x, tot);
}
__global__ void __launch_bounds__(PO_LANES) fr_sum_kernel(const Fr* __restrict__ v, uint64_t n, Fr* __restrict__ out, const PoCtx c) {
    __shared__ uint32_t sh[PO_LANES][9];
    Fr acc = fp_zero<8>();
    for (uint64_t i = threadIdx.x; i < n; i += PO_LANES) acc = fp_add(acc, load_fr(v + i), c.fp);
    const Fr tot = block_total<OpAdd, PO_LANES>(acc, sh, c);
    if (threadIdx.x == 0) store_fr(out, tot);
}

size_t poly_scratch_bytes(size_t len) { return 3 * align256((len / 1024 + 2) * 32) + 1024; }      // tile aggregates / carries of the division, partial sums of the evaluation

// rep(b^(2^k))
static F29 host_rep_pow2k(Fr b, int k, const FrParams& P) {
    for (int i = 0; i < k; i++) b = fp_mul(b, b, P);
    return host_rep(b, P);
}

int poly_eval_run(NttTables& T, const void* d_poly, size_t len, const uint64_t* point, uint64_t* out_host, void* scratch, hipStream_t stream) {
    const FrParams& P = T.fp;
    if (len >= ((size_t)1 << 30)) return plonk_fail(PLONK_ERR_ARG, "poly_eval: %zu coefficients (limit 2^30)", len);
    if (!fr_arg_ok(point, P)) return plonk_fail(PLONK_ERR_ARG, "poly_eval: point not reduced");
    if (len == 0) { memset(out_host, 0, 32); return PLONK_OK; }
    char* s = (char*)scratch;
    Fr* partial = (Fr*)s; s += align256(tiles_of(len) * 32);
    Fr* res = (Fr*)s;
    PowTab pw;
    int rc = build_pow_tab(T, fr_arg(point), len, &pw, stream);
    if (rc) return rc;
    const PoCtx c = make_ctx(T);
    const uint64_t nb = (len + EV_TILE - 1) / EV_TILE;
    {
        ProfScope ps("poly_eval_kernel", stream);
        hipLaunchKernelGGL(poly_eval_kernel, dim3((uint32_t)nb), dim3(PO_LANES), 0, stream, (const Fr*)d_poly, (uint64_t)len, host_rep_pow2k(fr_arg(point), 8, P), pw,
                           partial, c);
        hipLaunchKernelGGL(fr_sum_kernel, dim3(1), dim3(PO_LANES), 0, stream, (const Fr*)partial, nb, res, c);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "poly_eval launch: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(out_host, res, 32, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- rank 3: linear combination
#define LINCOMB_MAX 32
struct LincombParams {
    const Fr* poly[LINCOMB_MAX];
    uint64_t len[LINCOMB_MAX];
    F29 coef[LINCOMB_MAX];     // rep(c_k)
    Fr* out;
    uint64_t out_len;
    int k;
    PoCtx c;
};
__global__ void __launch_bounds__(256) poly_lincomb_kernel(const LincombParams P) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.out_len) return;
    const F29Params& fp = P.c.f29;
    F29 sum;
#pragma unroll
    for (int l = 0; l < 9; l++) sum.l[l] = 0;
    int pending = 0;
    for (int t = 0; t < P.k; t++) {
        if (i < P.len[t]) {
            sum = f29_add(sum, f29_mul(f29_from_sat(load_fr(P.poly[t] + i)), P.coef[t], fp));
            if (++pending == 3) { f29_norm(sum); pending = 0; }
        }
    }
    f29_norm(sum);                                                // < 44 p < 2^259.4
    store_fr(P.out + i, f29_to_sat(f29_canon(f29_mul(sum, params_one(fp), fp), fp)));
}

int poly_lincomb_run(NttTables& T, size_t k, const void* const* polys, const size_t* lens, const uint64_t* coeffs, void* d_out, size_t out_len,
                     hipStream_t stream) {
    const FrParams& P = T.fp;
    if (k == 0 || k > LINCOMB_MAX) return plonk_fail(PLONK_ERR_ARG, "poly_lincomb: %zu terms (1..%d)", k, LINCOMB_MAX);
    if (out_len == 0) return PLONK_OK;
    LincombParams q;
    memset(&q, 0, sizeof q);
    for (size_t t = 0; t < k; t++) {
        if (!polys[t] && lens[t]) return plonk_fail(PLONK_ERR_ARG, "poly_lincomb: null polynomial %zu", t);
        if (!fr_arg_ok(coeffs + 4 * t, P)) return plonk_fail(PLONK_ERR_ARG, "poly_lincomb: coefficient %zu not reduced", t);
        if (polys[t] == d_out) return plonk_fail(PLONK_ERR_ARG, "poly_lincomb: output aliases input %zu", t);
        q.poly[t] = (const Fr*)polys[t];
        q.len[t] = lens[t];
        q.coef[t] = host_rep(fr_arg(coeffs + 4 * t), P);
    }
    q.out = (Fr*)d_out; q.out_len = out_len; q.k = (int)k; q.c = make_ctx(T);
    {
        ProfScope ps("poly_lincomb_kernel", stream);
        hipLaunchKernelGGL(poly_lincomb_kernel, dim3((uint32_t)((out_len + 255) / 256)), dim3(256), 0, stream, q);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "poly_lincomb launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- rank 3: division by (X - z)
// q_j = sum_{k > j} c_k z^(k-j-1)  (the quotient of dispatcher2.rs:651-666; q_j = c_(j+1) + z q_(j+1) from the top).
// Round 5: reduce-then-scan over tiles of DV_TILE coefficients, the coefficients read twice and the quotient written once
// (96 B per coefficient; rounds 2-4: scale / three-phase additive scan / unscale, 160 B and 6 products per coefficient, 8.3 % of HBM peak):
//   D1  tile t (coefficients k = 1 + t*DV_TILE + e):  A_t = sum_e c_k z^e      — lane-strided Horner as in poly_eval, one product per coefficient
//   D2  one workgroup:  X_t = sum_{u > t} A_u (z^DV_TILE)^(u-t-1) = q at the first index above tile t
//   D3  tile t: the coefficients go through LDS so that lane L owns DV_CH CONSECUTIVE ones (the recurrence is serial in j):
//       lane aggregate (DV_CH products) -> scaled by z^(DV_CH*L) (1) -> ADDITIVE suffix scan over the lanes (no products) -> unscaled by
//       z^-(DV_CH*(L+1)) with the tile's carry folded in (2) -> the recurrence down the lane's chunk (DV_CH) -> LDS -> coalesced stores.
//       2 + 3/DV_CH products per coefficient.
#define DV_LANES 128
#define DV_CH 8
#define DV_TILE (DV_LANES * DV_CH)
#define DV_TOP 1024
// per-point constants (cached like the power tables): [0, DV_LANES) rep(z^(DV_CH*L)); [DV_LANES, 2*DV_LANES) rep(z^-(DV_CH*(L+1)));
// then rep(z), rep(z^DV_LANES), rep(z^DV_TILE)
#define DVT_Z (2 * DV_LANES)
#define DVT_ZL (2 * DV_LANES + 1)
#define DVT_ZT (2 * DV_LANES + 2)
#define DVT_N (2 * DV_LANES + 3)
// The relations these kernels rely on (the constants DO get changed — the tile-shape A/B of round 5 — and a shape that breaks one of
// them would give wrong openings with PLONK_OK):
static_assert(PO_LANES == (1 << 8), "poly_eval: the lane-strided Horner multiplies by z^PO_LANES = host_rep_pow2k(point, 8)");
static_assert((DV_CH & (DV_CH - 1)) == 0 && (DV_LANES & (DV_LANES - 1)) == 0, "poly_div: z^DV_CH, z^DV_LANES and z^DV_TILE are built by repeated squaring");
static_assert(DV_LANES <= 1024, "poly_div_agg_kernel indexes the first level of the power table (1024 entries) with the lane number");
static_assert(DV_TILE == DV_LANES * DV_CH && DV_TOP >= 2, "poly_div: a tile is DV_LANES lanes x DV_CH consecutive coefficients");
static int build_div_tab(NttTables& T, const Fr& z_mont, const F29** out, hipStream_t stream) {
    const FrParams& P = T.fp;
    std::string key((const char*)z_mont.l, 32);
    key.push_back('D');
    auto it = T.pow_tabs.find(key);
    if (it != T.pow_tabs.end()) {
        auto pos = std::find(T.pow_order.begin(), T.pow_order.end(), key);
        if (pos != T.pow_order.end()) { T.pow_order.erase(pos); T.pow_order.push_back(key); }
        *out = it->second;
        return PLONK_OK;
    }
    std::vector<F29> h(DVT_N);
    Fr zc = z_mont;
    for (int i = 1; i < DV_CH; i <<= 1) zc = fp_mul(zc, zc, P);                  // z^DV_CH (DV_CH is a power of two)
    const Fr zci = fp_inv(zc, P);
    Fr up = fp_one(P), dn = zci;
    for (int L = 0; L < DV_LANES; L++) {
        h[L] = host_rep(up, P);
        h[DV_LANES + L] = host_rep(dn, P);
        up = fp_mul(up, zc, P);
        dn = fp_mul(dn, zci, P);
    }
    Fr zl = z_mont;
    for (int i = 1; i < DV_LANES; i <<= 1) zl = fp_mul(zl, zl, P);
    h[DVT_Z] = host_rep(z_mont, P);
    h[DVT_ZL] = host_rep(zl, P);
    h[DVT_ZT] = host_rep(up, P);                                                  // (z^DV_CH)^DV_LANES
    if (T.pow_order.size() >= POWTAB_CACHE_MAX) {
        HIP_TRY(hipStreamSynchronize(stream));
        (void)hipFree(T.pow_tabs[T.pow_order.front()]);
        T.pow_tabs.erase(T.pow_order.front());
        T.pow_order.erase(T.pow_order.begin());
    }
    F29* d_tab = nullptr;
    HIP_TRY(hipMalloc((void**)&d_tab, DVT_N * sizeof(F29)));
    HIP_TRY(hipMemcpyAsync(d_tab, h.data(), h.size() * sizeof(F29), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    T.pow_tabs[key] = d_tab;
    T.pow_order.push_back(key);
    *out = d_tab;
    return PLONK_OK;
}

// D1: A_t = sum_e c[1 + t*DV_TILE + e] z^e  (canonical Fr, R form)
__global__ void __launch_bounds__(DV_LANES) poly_div_agg_kernel(const Fr* __restrict__ poly, uint64_t len, const F29* __restrict__ tab, const PowTab pw,
                                                                Fr* __restrict__ agg, const PoCtx c) {
    __shared__ uint32_t sh[DV_LANES][9];
    const F29Params& fp = c.f29;
    const F29 zl = load_f29(tab + DVT_ZL);
    const uint64_t base = 1 + (uint64_t)blockIdx.x * DV_TILE + threadIdx.x;
    F29 acc;
#pragma unroll
    for (int l = 0; l < 9; l++) acc.l[l] = 0;
    bool any = false;
#pragma unroll
    for (int k = DV_CH - 1; k >= 0; k--) {
        const uint64_t i = base + (uint64_t)k * DV_LANES;
        if (i < len) {
            const F29 ci = f29_from_sat(load_fr(poly + i));
            acc = any ? f29_add(f29_mul(acc, zl, fp), ci) : ci;
            any = true;
        }
    }
    Fr mine = fp_zero<8>();
    if (any) mine = f29_to_sat(f29_canon(f29_mul(acc, load_f29(pw.t0 + threadIdx.x), fp), fp));       // * z^L (level 0 of the power table: L < 1024)
    const Fr tot = block_total<OpAdd, DV_LANES>(mine, sh, c);
    if (threadIdx.x == 0) store_fr(agg + blockIdx.x, tot);
}

// D2: X_t = A_(t+1) + W X_(t+1), X_(nt-1) = 0, W = z^DV_TILE.  One workgroup; lane l owns the `per` tiles [l*per, (l+1)*per).
struct DivTopParams {
    F29 w;             // rep(W)
    F29 wstep[10];     // rep(W^(per * 2^s)): the lane-level suffix scan's multipliers
};
__global__ void __launch_bounds__(DV_TOP) poly_div_top_kernel(const Fr* __restrict__ agg, Fr* __restrict__ carry, uint64_t nt, uint64_t per, const DivTopParams q,
                                                              const PoCtx c) {
    __shared__ uint32_t sh[DV_TOP][9];
    const F29Params& fp = c.f29;
    const int t = threadIdx.x;
    const uint64_t lo = (uint64_t)t * per, hi = lo + per < nt ? lo + per : nt;
    // G_l = sum_{u in run} A_u W^(u - lo)
    F29 g;
#pragma unroll
    for (int l = 0; l < 9; l++) g.l[l] = 0;
    for (uint64_t u = hi; u > lo; u--) {
        const F29 a = f29_from_sat(load_fr(agg + u - 1));
        g = (u == hi) ? a : f29_add(f29_mul(g, q.w, fp), a);
    }
    f29_norm(g);                                                              // < 2.4 p
    // inclusive suffix windows: V_l <- V_l + W^(per*d) V_(l+d)
    OpMul::to_words(sh[t], g);
    __syncthreads();
    int s = 0;
    for (int d = 1; d < DV_TOP; d <<= 1, s++) {
        F29 o;
        const bool on = t + d < DV_TOP;
        if (on) o = OpMul::from_words(sh[t + d]);
        __syncthreads();
        if (on) {
            g = f29_add(g, f29_mul(o, q.wstep[s], fp));                       // + < 1.36 p per step: < 16 p after the ten steps, inside f29_mul's range
            f29_norm(g);
            OpMul::to_words(sh[t], g);
        }
        __syncthreads();
    }
    // Y_l = V_(l+1): q just above this lane's run; then down the run
    F29 x;
#pragma unroll
    for (int l = 0; l < 9; l++) x.l[l] = 0;
    if (t + 1 < DV_TOP) x = OpMul::from_words(sh[t + 1]);
    for (uint64_t u = hi; u > lo; u--) {                                       // X_(u-1) from X_u ... stored for tile u-1
        store_fr(carry + u - 1, f29_to_sat(f29_canon_lazy(x, fp)));           // x normalised, < 16 p
        x = f29_add(f29_mul(x, q.w, fp), f29_from_sat(load_fr(agg + u - 1)));  // < 2.4 p
        f29_norm(x);
    }
}

// LDS layout of a tile: element e at word e*8 + (e >> 3) — one pad word per DV_CH elements makes both the coalesced (lane = e mod DV_LANES)
// and the chunked (lane = e / DV_CH) accesses 2 lanes per bank
__device__ __forceinline__ uint32_t dv_word(uint32_t e) { return e * 8 + (e >> 3); }
#define DV_LDS_WORDS (DV_TILE * 8 + DV_TILE / 8)
// D3
__global__ void __launch_bounds__(DV_LANES) poly_div_apply_kernel(const Fr* __restrict__ poly, uint64_t len, const F29* __restrict__ tab,
                                                                  const Fr* __restrict__ carry, Fr* __restrict__ out, const PoCtx c) {
    __shared__ uint32_t tile[DV_LDS_WORDS];
    __shared__ uint32_t sh[DV_LANES][9];
    const F29Params& fp = c.f29;
    const uint32_t L = threadIdx.x;
    const uint64_t j0 = (uint64_t)blockIdx.x * DV_TILE;                       // outputs j0 + e, coefficients j0 + 1 + e
    // coalesced loads -> LDS
#pragma unroll
    for (int m = 0; m < DV_CH; m++) {
        const uint32_t e = m * DV_LANES + L;
        const uint64_t k = j0 + 1 + e;
        Fr v = fp_zero<8>();
        if (k < len) v = load_fr(poly + k);
        uint32_t* w = tile + dv_word(e);
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = v.l[i];
    }
    __syncthreads();
    // this lane's DV_CH consecutive coefficients are read from the LDS tile where they are used — twice — and never held in an array: the loops
    // below are not unrolled (the pinned multiplier exceeds the unroll budget) and an indexed array would live in scratch memory (the first
    // form of this kernel: 304 B of scratch per lane, 2.5x the algorithmic traffic in the PMC counters, 0.66 ms)
    auto coeff = [&](int i) {
        const uint32_t* w = tile + dv_word(L * DV_CH + i);
        Fr v;
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = w[k];
        return f29_from_sat(v);
    };
    const F29 z = load_f29(tab + DVT_Z);
    // lane aggregate sum_i x[i] z^i, scaled by z^(DV_CH*L)
    F29 a = coeff(DV_CH - 1);
    for (int i = DV_CH - 2; i >= 0; i--) a = f29_add(f29_mul(a, z, fp), coeff(i));
    const Fr bl = f29_to_sat(f29_canon(f29_mul(a, load_f29(tab + L), fp), fp));
    // exclusive additive suffix scan over the lanes: S_L = sum_{L' > L} B_L'
    {
        const uint32_t r = DV_LANES - 1 - L;                                   // position in the reversed order: a prefix scan there
        Fr v = bl;
        OpAdd::to_words(sh[r], v);
        __syncthreads();
        for (uint32_t d = 1; d < DV_LANES; d <<= 1) {
            Fr o;
            const bool on = r >= d;
            if (on) o = OpAdd::from_words(sh[r - d]);
            __syncthreads();
            if (on) {
                v = fp_add(o, v, c.fp);
                OpAdd::to_words(sh[r], v);
            }
            __syncthreads();
        }
    }
    const uint32_t r = DV_LANES - 1 - L;
    F29 u = f29_from_sat(r > 0 ? OpAdd::from_words(sh[r - 1]) : fp_zero<8>());
    // + z^DV_TILE * X_t, then * z^-(DV_CH*(L+1)):  q at the first index above this lane's chunk
    u = f29_add(u, f29_mul(f29_from_sat(load_fr(carry + blockIdx.x)), load_f29(tab + DVT_ZT), fp));
    F29 q = f29_mul(u, load_f29(tab + DV_LANES + L), fp);
    // the recurrence down the chunk: q_(j0 + L*DV_CH + i) = x[i] + z * q_(that + 1); slot (L, i) of the tile is read, then overwritten, by this lane only
    for (int i = DV_CH - 1; i >= 0; i--) {
        q = f29_add(f29_mul(q, z, fp), coeff(i));                              // < 2.4 p, limbs < 2^30
        F29 qn = q;
        f29_norm(qn);
        const Fr v = f29_to_sat(f29_canon_lazy(qn, fp));                       // canonical without a product
        uint32_t* w = tile + dv_word(L * DV_CH + i);
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = v.l[k];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < DV_CH; m++) {
        const uint32_t e = m * DV_LANES + L;
        const uint64_t j = j0 + e;
        if (j + 1 < len) {
            const uint32_t* w = tile + dv_word(e);
            Fr v;
#pragma unroll
            for (int k = 0; k < 8; k++) v.l[k] = w[k];
            store_fr(out + j, v);
        }
    }
}
__global__ void __launch_bounds__(256) poly_shift_down_kernel(const Fr* __restrict__ poly, uint64_t len, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < len) store_fr(out + i, load_fr(poly + i + 1));
}

int poly_div_linear_run(NttTables& T, const void* d_poly, size_t len, const uint64_t* point, void* d_out, void* scratch, hipStream_t stream) {
    const FrParams& P = T.fp;
    if (len >= ((size_t)1 << 30)) return plonk_fail(PLONK_ERR_ARG, "poly_div_linear: %zu coefficients (limit 2^30)", len);
    if (!fr_arg_ok(point, P)) return plonk_fail(PLONK_ERR_ARG, "poly_div_linear: point not reduced");
    if (len < 2) return PLONK_OK;
    if (d_poly == d_out) return plonk_fail(PLONK_ERR_ARG, "poly_div_linear: in-place not supported");
    const Fr z = fr_arg(point);
    if (fp_is_zero(z)) {                                          // q_{i-1} = c_i
        hipLaunchKernelGGL(poly_shift_down_kernel, dim3((uint32_t)((len + 255) / 256)), dim3(256), 0, stream, (const Fr*)d_poly, (uint64_t)len, (Fr*)d_out);
        return PLONK_OK;
    }
    const uint64_t nt = (len - 1 + DV_TILE - 1) / DV_TILE;       // tiles of the len - 1 quotient coefficients
    char* s = (char*)scratch;
    Fr* agg = (Fr*)s; s += align256(nt * 32);
    Fr* carry = (Fr*)s;
    const F29* tab = nullptr;
    PowTab pz;
    int rc = build_div_tab(T, z, &tab, stream);
    if (!rc) rc = build_pow_tab(T, z, 1024, &pz, stream);        // level 0 only: z^L, L < DV_LANES
    if (rc) return rc;
    const PoCtx c = make_ctx(T);
    DivTopParams q;
    const uint64_t per = (nt + DV_TOP - 1) / DV_TOP;
    Fr w = z;
    for (int i = 1; i < DV_TILE; i <<= 1) w = fp_mul(w, w, P);   // z^DV_TILE
    q.w = host_rep(w, P);
    Fr ws = fp_one(P), b = w;                                     // W^per by square and multiply
    for (uint64_t e = per; e; e >>= 1) { if (e & 1) ws = fp_mul(ws, b, P); b = fp_mul(b, b, P); }
    for (int i = 0; i < 10; i++) { q.wstep[i] = host_rep(ws, P); ws = fp_mul(ws, ws, P); }
    {
        ProfScope ps("poly_div_kernels", stream);
        hipLaunchKernelGGL(poly_div_agg_kernel, dim3((uint32_t)nt), dim3(DV_LANES), 0, stream, (const Fr*)d_poly, (uint64_t)len, tab, pz, agg, c);
        hipLaunchKernelGGL(poly_div_top_kernel, dim3(1), dim3(DV_TOP), 0, stream, (const Fr*)agg, carry, nt, per, q, c);
        hipLaunchKernelGGL(poly_div_apply_kernel, dim3((uint32_t)nt), dim3(DV_LANES), 0, stream, (const Fr*)d_poly, (uint64_t)len, tab, (const Fr*)carry, (Fr*)d_out, c);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "poly_div_linear launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- arbitrary cosets
// Evaluation of a coefficient vector on {shift * w_size^k, k < size} and its inverse, for ANY shift — the building block of
// coset-class parallelism (rank s of G evaluates every polynomial on the points g*w_m^(s+Gk) = (g*w_m^s) * w_(m/G)^k, which
// makes the quotient kernel rank-local).  With shift = g, size = m this is Radix2EvaluationDomain::coset_fft / coset_ifft.
//   eval:    NTT_size of  shift^i * sum_u (shift^size)^u * a[i + u*size]   (coefficients beyond `size` fold back: X^size = shift^size
//            on the coset) — fused into the first pass of the transform (shared-input mode of ntt_run: the shift enters as a row
//            table and the first inter-pass plane), and decomposed into 2^k sub-cosets when fewer than size/2 coefficients are
//            non-zero (see coset_eval_run).
//   interp:  E = iNTT_size(evals);  out[t] = scale * shift^-(i0+t) * E[(i0+t) mod size]   — the contribution of this coset to
//            coefficient i0+t when `scale` = 1/G and G cosets tile a domain of G*size points; G = 1: coset_ifft itself.
__global__ void __launch_bounds__(256) coset_unscale_kernel(const Fr* __restrict__ e, uint64_t size_mask, uint64_t i0, uint64_t count, const PowTab pw,
                                                            Fr* __restrict__ out, const PoCtx c) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint64_t i = i0 + t;
    store_fr(out + t, f29_to_sat(f29_canon(f29_mul(f29_from_sat(load_fr(e + (i & size_mask))), pow_at(pw, i, c.f29), c.f29), c.f29)));
}

size_t coset_scratch_bytes(size_t size) { return align256(size * 32) + align256(POWTAB_BYTES) + 256; }

int coset_eval_run(NttTables& T, const void* d_poly, size_t len, size_t size, const uint64_t* shift, void* d_out, void* scratch, hipStream_t stream) {
    const FrParams& P = T.fp;
    int log_s = 0;
    while (((size_t)1 << log_s) < size) log_s++;
    if (((size_t)1 << log_s) != size || log_s < 1) return plonk_fail(PLONK_ERR_DOMAIN, "coset_eval: size %zu is not a power of two >= 2", size);
    if (log_s > T.two_adicity) return plonk_fail(PLONK_ERR_DOMAIN, "coset_eval: 2^%d exceeds the two-adicity", log_s);
    if (len > NTT_MAX_FOLD * size) return plonk_fail(PLONK_ERR_ARG, "coset_eval: %zu coefficients for a %zu-point coset (limit %dx)", len, size, NTT_MAX_FOLD);
    if (!fr_arg_ok(shift, P)) return plonk_fail(PLONK_ERR_ARG, "coset_eval: shift not reduced");
    if (d_poly == d_out) return plonk_fail(PLONK_ERR_ARG, "coset_eval: d_out must not alias d_poly");
    if (len == 0) { HIP_TRY(hipMemsetAsync(d_out, 0, size * 32, stream)); return PLONK_OK; }      // the zero polynomial
    // Zero-padding awareness (the reference zero-pads n+2 / n+3 coefficients to the 8n-point domain, dispatcher2.rs:746): with
    // B = 2^k classes, the evaluations at the
    // points of index q, q+B, q+2B, ... are a size/B-point transform on the coset (shift * w_size^q) * <w_(size/B)> of the SAME
    // coefficients (folded modulo size/B).  B independent transforms of log(size/B) stages each, no zero is ever loaded, multiplied
    // or stored, and the last pass interleaves the B results into natural order.
    int k = 0;
    // one more halving removes a stage (0.5 products per point) and folds len - size/2B coefficients into each of 2B classes (one
    // product each): worth it while len < 1.5 * size/2B.  n + 3 coefficients on 8n points: B = 8 classes of n points.
    while (k < 4 && log_s - k > 1 && 2 * len < 3 * (size >> (k + 1))) k++;
    NttCall call;
    call.in = (const Fr*)d_poly; call.out = (Fr*)d_out; call.log_m = log_s - k; call.batch = (uint64_t)1 << k; call.inverse = false;
    call.shared_in = true; call.in_len = len; call.shift = fr_arg(shift); call.work = (Fr*)scratch;
    call.out_layout = k ? NTT_INTERLEAVED : NTT_CONTIGUOUS;
    return ntt_run(T, call, stream);
}

int coset_interp_run(NttTables& T, void* d_evals, size_t size, const uint64_t* shift, const uint64_t* scale, size_t i0, size_t count, void* d_out,
                     void* scratch, hipStream_t stream) {
    const FrParams& P = T.fp;
    int log_s = 0;
    while (((size_t)1 << log_s) < size) log_s++;
    if (((size_t)1 << log_s) != size || log_s < 1) return plonk_fail(PLONK_ERR_DOMAIN, "coset_interp: size %zu is not a power of two >= 2", size);
    if (log_s > T.two_adicity) return plonk_fail(PLONK_ERR_DOMAIN, "coset_interp: 2^%d exceeds the two-adicity", log_s);
    if (i0 + count >= ((size_t)1 << 30)) return plonk_fail(PLONK_ERR_ARG, "coset_interp: index range beyond 2^30");
    if (!fr_arg_ok(shift, P) || !fr_arg_ok(scale, P)) return plonk_fail(PLONK_ERR_ARG, "coset_interp: shift/scale not reduced");
    const Fr h = fr_arg(shift);
    if (fp_is_zero(h)) return plonk_fail(PLONK_ERR_ARG, "coset_interp: zero shift");
    if (count == 0) return PLONK_OK;
    Fr* tmp = (Fr*)scratch;
    const Fr sc = fr_arg(scale);
    PowTab pw;
    int rc = build_pow_tab(T, fp_inv(h, P), i0 + count, &pw, stream, &sc);
    if (rc) return rc;
    NttCall call;
    call.in = (const Fr*)d_evals; call.out = tmp; call.log_m = log_s; call.batch = 1; call.inverse = true;     // includes the 1/size factor
    rc = ntt_run(T, call, stream);
    if (rc) return rc;
    const PoCtx c = make_ctx(T);
    {
        ProfScope ps("coset_unscale_kernel", stream);
        hipLaunchKernelGGL(coset_unscale_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, stream, (const Fr*)tmp, (uint64_t)size - 1, (uint64_t)i0,
                           (uint64_t)count, pw, (Fr*)d_out, c);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "coset_unscale launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- classes -> natural order
// out[t*G + s] = scale * in[s*L + (reverse ? (L - t) mod L : t)],  s < G classes of L values each: what the dispatcher does with the
// workers' replies of a distributed transform (dispatcher2.rs:776-787: concatenate, transpose), for the class-decomposed size-n iNTT of
// class_prover.py — rank s evaluates the n evaluations, read as coefficients, at w_n^-s * w_L^t' (plonk_coset_eval_dev, forward roots), so
// coefficient s + G*t of the interpolant is 1/n times value (L - t) mod L of class s.  One lane per t: G coalesced (reversed) loads, one
// contiguous G*32-byte store.
template <int G>
__global__ void __launch_bounds__(256) class_interleave_kernel(const Fr* __restrict__ in, uint64_t L, uint64_t stride, int reverse, int scaled,
                                                               const F29 scale_c, Fr* __restrict__ out, const PoCtx c) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L) return;
    const uint64_t src = reverse ? (t == 0 ? 0 : L - t) : t;
#pragma unroll
    for (int s = 0; s < G; s++) {
        Fr v = load_fr(in + (uint64_t)s * stride + src);
        if (scaled) v = f29_to_sat(f29_canon(f29_mul(f29_from_sat(v), scale_c, c.f29), c.f29));
        store_fr(out + t * G + s, v);
    }
}
int class_interleave_run(NttTables& T, const void* d_in, size_t classes, size_t size, size_t in_stride, int reverse, const uint64_t* scale, void* d_out,
                         hipStream_t stream) {
    const FrParams& P = T.fp;
    if (classes == 0 || (classes & (classes - 1)) || classes > 8) return plonk_fail(PLONK_ERR_ARG, "class_interleave: %zu classes (1, 2, 4 or 8)", classes);
    if (size == 0 || size * classes >= ((size_t)1 << 32)) return plonk_fail(PLONK_ERR_ARG, "class_interleave: %zu values per class", size);
    if (in_stride == 0) in_stride = size;
    if (in_stride < size) return plonk_fail(PLONK_ERR_ARG, "class_interleave: class stride %zu below the class size %zu", in_stride, size);
    if (scale && !fr_arg_ok(scale, P)) return plonk_fail(PLONK_ERR_ARG, "class_interleave: scale not reduced");
    if (d_in == d_out) return plonk_fail(PLONK_ERR_ARG, "class_interleave: in-place not supported");
    const PoCtx c = make_ctx(T);
    F29 sc;
    memset(&sc, 0, sizeof sc);
    if (scale) sc = host_rep(fr_arg(scale), P);
    const dim3 grid((uint32_t)((size + 255) / 256)), block(256);
    ProfScope ps("class_interleave_kernel", stream);
    switch (classes) {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(class_interleave_kernel<1>), grid, block, 0, stream, (const Fr*)d_in, (uint64_t)size, (uint64_t)in_stride, reverse, scale ? 1 : 0, sc, (Fr*)d_out, c); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(class_interleave_kernel<2>), grid, block, 0, stream, (const Fr*)d_in, (uint64_t)size, (uint64_t)in_stride, reverse, scale ? 1 : 0, sc, (Fr*)d_out, c); break;
    case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(class_interleave_kernel<4>), grid, block, 0, stream, (const Fr*)d_in, (uint64_t)size, (uint64_t)in_stride, reverse, scale ? 1 : 0, sc, (Fr*)d_out, c); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(class_interleave_kernel<8>), grid, block, 0, stream, (const Fr*)d_in, (uint64_t)size, (uint64_t)in_stride, reverse, scale ? 1 : 0, sc, (Fr*)d_out, c); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "class_interleave launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- blinding
struct BlindParams {
    Fr b[4];
    int k;
};
// poly[i] -= b_i ; poly[n+i] += b_i   ((sum b_i X^i)(X^n - 1) + poly), dispatcher2.rs:311-312,347-348 / worker.rs:400-401
__global__ void blind_kernel(Fr* poly, uint64_t n, const BlindParams B, const FrParams P) {
    const int i = threadIdx.x;
    if (n >= (uint64_t)B.k) {                  // the two index ranges are disjoint: one lane per blinder
        if (i < B.k) {
            store_fr(poly + i, fp_sub(load_fr(poly + i), B.b[i], P));
            store_fr(poly + n + i, fp_add(load_fr(poly + n + i), B.b[i], P));
        }
    } else if (i == 0) {                       // a domain smaller than the mask (n = 2, three blinders): [0, k) and [n, n + k) overlap, one lane does all
        for (int j = 0; j < B.k; j++) store_fr(poly + j, fp_sub(load_fr(poly + j), B.b[j], P));
        for (int j = 0; j < B.k; j++) store_fr(poly + n + j, fp_add(load_fr(poly + n + j), B.b[j], P));
    }
}
int blind_run(NttTables& T, void* d_poly, size_t n, const uint64_t* blinders, size_t k, hipStream_t stream) {
    if (k == 0) return PLONK_OK;
    if (k > 4 || n == 0) return plonk_fail(PLONK_ERR_ARG, "blind: %zu blinders for n = %zu (1..4 blinders, n >= 1)", k, n);
    BlindParams B;
    memset(&B, 0, sizeof B);
    for (size_t i = 0; i < k; i++) {
        if (!fr_arg_ok(blinders + 4 * i, T.fp)) return plonk_fail(PLONK_ERR_ARG, "blind: blinder %zu not reduced", i);
        B.b[i] = fr_arg(blinders + 4 * i);
    }
    B.k = (int)k;
    hipLaunchKernelGGL(blind_kernel, dim3(1), dim3(64), 0, stream, (Fr*)d_poly, (uint64_t)n, B, T.fp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "blind launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- degree (DensePolynomial trimming)
// index of the highest non-zero coefficient + 1 (0 for the zero polynomial): what DensePolynomial::from_coefficients_vec
// leaves after trimming, used for the WrongQuotientPolyDegree check of dispatcher2.rs:511-518
// One workgroup scans DEG_CH * 256 consecutive coefficients (coalesced: lane-strided), folds its highest non-zero index in LDS and issues ONE atomicMax — and only if
// it found one.  (Until round 6 every non-zero coefficient issued its own atomicMax on the one result word: 84 M serialised atomics for the degree-(5n + 7) quotient of a
// 2^24-gate circuit, 14 - 16 ms of round 3 — tools/commit_round_probe.py found it as the gap between round 3's commitment phase and the same commitments alone.)
#define DEG_CH 8
__global__ void __launch_bounds__(256) poly_degree_kernel(const Fr* __restrict__ poly, uint64_t len, unsigned long long* __restrict__ top) {
    __shared__ unsigned long long best[256];
    const uint64_t base = (uint64_t)blockIdx.x * (256 * DEG_CH);
    unsigned long long cand = 0;
#pragma unroll
    for (int k = 0; k < DEG_CH; k++) {
        const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        if (i < len && !fp_is_zero(load_fr(poly + i))) cand = (unsigned long long)(i + 1);      // i grows with k: the last hit is this lane's highest
    }
    best[threadIdx.x] = cand;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d && best[threadIdx.x + d] > best[threadIdx.x]) best[threadIdx.x] = best[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0 && best[0] != 0) atomicMax(top, best[0]);
}
int poly_degree_run(const void* d_poly, size_t len, int64_t* degree, void* scratch, hipStream_t stream) {
    unsigned long long* top = (unsigned long long*)scratch;
    HIP_TRY(hipMemsetAsync(top, 0, 8, stream));
    if (len) hipLaunchKernelGGL(poly_degree_kernel, dim3((uint32_t)((len + 256 * DEG_CH - 1) / (256 * DEG_CH))), dim3(256), 0, stream, (const Fr*)d_poly, (uint64_t)len, top);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "poly_degree launch: %s", hipGetErrorString(e));
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, top, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *degree = (int64_t)h - 1;
    return PLONK_OK;
}
