// plonk_internal.hpp — internal C++ interfaces behind the C ABI of include/plonk_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "fp.hpp"
#include "fp29.hpp"
#include "../../include/plonk_hip.h"

typedef Fp<8> Fr;
typedef FpParams<8> FrParams;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            snprintf(plonk_last_error_buf(), 512, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, \
                     hipGetErrorString(_e));                                                   \
            return PLONK_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

char* plonk_last_error_buf();
int plonk_fail(int code, const char* fmt, ...);

// Raised dynamic-LDS limits are a per-DEVICE property of a kernel: a guard remembers the devices it has set the attribute on (one process per GPU is
// the deployment, but a second worker on another device of the same process must not launch without it).  Two host threads racing on a fresh device
// both set the same value.
struct DeviceOnce {
    uint64_t done = 0;                      // bit d: device d has the attribute (devices >= 64 set it on every call)
    template <class F> hipError_t run(F&& set) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = dev >= 0 && dev < 64 ? (uint64_t)1 << dev : 0;
        if (bit && (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit)) return hipSuccess;
        e = set();
        if (e == hipSuccess && bit) __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
        return e;
    }
};

// ----------------------------------------------------------------------------------------------- per-kernel timing
// HIP-event timing of individual kernel launches on the stream they are launched on (bench.py's
// roofline leg).  Disabled by default (no events are created).
struct KernelProfiler {       // process-wide; contexts on different host threads record into it concurrently
    std::mutex mu;
    bool enabled = false;
    struct Rec { const char* name; hipEvent_t a, b; };
    std::vector<Rec> pending;
    std::unordered_map<std::string, std::pair<double, uint64_t>> totals;   // name -> (ms, launches)
    void resolve();
    void reset();
};
KernelProfiler& kernel_profiler();
struct ProfScope {       // RAII: brackets the launches issued in its lifetime
    const char* name; hipStream_t stream; hipEvent_t a = nullptr, b = nullptr; bool on;
    ProfScope(const char* n, hipStream_t s) : name(n), stream(s), on(kernel_profiler().enabled) {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, stream); }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(b, stream);
            std::lock_guard<std::mutex> g(kernel_profiler().mu);
            kernel_profiler().pending.push_back({name, a, b});
        }
    }
};

// ----------------------------------------------------------------------------------------------- NTT
struct NttTables {
    int curve = 0;
    FrParams fp;
    F29Params fp29;
    int two_adicity = 0;
    int lt = 0;                       // two-level table split: 2^lt entries per level, 2*lt >= two_adicity
    // all device tables hold constants c*2^261 mod p as 9 x 29-bit limbs (fp29.hpp)
    F29* tw_small[2] = {nullptr, nullptr};   // [dir] w_Rmax^e
    F29S* tw_shoup[2] = {nullptr, nullptr};  // [dir] w_Rmax^e prepared for the precomputed-quotient multiplier (plain residue + quotient constant)
    bool use_shoup = false;                  // precomputed-quotient butterflies (bounds: ntt_kernels.hpp); option "ntt_shoup", PLONK_NTT_NO_SHOUP=1 sets the initial value to off
    F29* tw_lo[2] = {nullptr, nullptr};      // [dir] w_Nmax^e
    F29* tw_hi[2] = {nullptr, nullptr};      // [dir] w_Nmax^(e<<lt)
    F29* g_lo[2] = {nullptr, nullptr};       // [0] g^e      [1] g^-e
    F29* g_hi[2] = {nullptr, nullptr};       // [0] g^(e<<lt) ...
    std::unordered_map<int, F29*> tw_lo_scaled;   // key = log_m (inverse only): w^-e * 2^-log_m
    std::unordered_map<uint64_t, Fr*> planes;     // inter-pass factor planes (see ntt_engine.hip: plane_key)
    std::unordered_map<uint64_t, F29*> rowtabs;   // coset row tables g^(a*r_1)
    // first-pass tables of class-decomposed coset evaluations (ntt_engine.hip: get_shift_set), keyed by (shift, log M, log B, first width)
    struct ShiftSet {
        Fr* planes = nullptr;       // [B][M]: w_M^(b*i) * h_q^b            (transforms of two or more passes)
        F29* rowtabs = nullptr;     // [B][R_1]: h_q^(a * r_1)
        F29* foldc = nullptr;       // [B][NTT_MAX_FOLD]: (h_q^M)^u
        size_t bytes = 0;
    };
    std::unordered_map<std::string, ShiftSet> shift_sets;
    size_t plane_bytes = 0;                       // HBM currently held by planes
    size_t plane_budget = (size_t)48 << 30;       // stop creating planes beyond this (fall back to on-the-fly factors)
    // quotient.hip: g * w_Nmax^e (constant form) and 1/(x_i - 1) per quotient-domain size (key = log m)
    F29* quot_x_lo = nullptr;
    std::unordered_map<int, Fr*> quot_inv_xm1;     // key: log m | class stride << 8 | class offset << 16
    // poly_ops.hip: three-level power tables z^i keyed by (z, levels, scale) — LRU cache (pow_order: least recent first)
    std::unordered_map<std::string, F29*> pow_tabs;
    std::vector<std::string> pow_order;
    // per-context tuning knobs (plonk_set_option): tests and experiments only; a context is driven by one host thread at a time
    int max_log_r = 9;                      // "ntt_max_log_r": widest in-LDS transform of a pass (<= NTT_LOG_RMAX)
    int quotient_fuse = 6;                  // "quotient_fuse": kernel formulations of quotient.hip (6 = the shipped compact kernel; 0 = the round 2-3 kernel; 1-5, 7: experiments)
    std::vector<Fr> h_pow2_inv;             // 2^-k in Montgomery form, k = 0..two_adicity
    Fr h_root[2];                           // w_Nmax, w_Nmax^-1 (Montgomery), Nmax = 2^(2*lt) clipped to two-adicity
};

enum NttLayout { NTT_CONTIGUOUS = 0, NTT_INTERLEAVED = 1 };

struct ScaleSpec {          // exponent = idx*(bq*q + b0) + (aq*q + a0)
    int kind = 0;           // 0 none, 1 powers of g (coset), 2 powers of g^-1, 3 powers of w_N (fwd dir), 4 powers of w_N^-1
    uint64_t aq = 0, a0 = 0, bq = 0, b0 = 0;
    int log_order = 0;      // for kind 3/4: N = 2^log_order (exponent is scaled to the Nmax table)
};

struct NttCall {
    const Fr* in = nullptr;   // destroyed when passes > 1
    Fr* out = nullptr;        // must differ from `in` unless a single contiguous pass is used
    int log_m = 0;            // transform size per array
    uint64_t batch = 1;       // number of arrays
    NttLayout layout = NTT_CONTIGUOUS;   // element (q,pos): contiguous q*M+pos ; interleaved pos*batch+q
    bool inverse = false;     // w^-1 and 1/M scaling
    ScaleSpec pro;            // applied to inputs (idx = position in array)
    ScaleSpec epi;            // applied to outputs (idx = natural-order output index)
    uint64_t q_offset = 0;    // added to q in both scale specs
    int split_log = -1;       // output re-blocking for the all-to-all send buffer (see ntt_kernels.hpp)
    uint64_t split_blk = 0;
    NttLayout out_layout = NTT_CONTIGUOUS;
    // Class-decomposed evaluation of ONE coefficient vector on the cosets h_q * <w_M>, h_q = shift * w_(M*batch)^q, q < batch
    // (coset_eval_run): every array reads the same `in` (in_len coefficients, NOT modified; zero beyond in_len; coefficients beyond M
    // fold back), layout contiguous, forward only, no other scale.  With out_layout = NTT_INTERLEAVED the result is the natural-order
    // evaluation vector on the coset shift * <w_(M*batch)> — the zero-padding-aware coset FFT of dispatcher2.rs:387-424.
    // `work` (M*batch elements) receives the intermediate passes.
    bool shared_in = false;
    uint64_t in_len = 0;
    Fr shift;
    Fr* work = nullptr;
    // Batched form (the distributed row pass on zero-padded rows): `in_rows` independent coefficient vectors of in_len coefficients,
    // in_pitch elements apart; batch = in_rows * classes, array = row * classes + class; the classes of a row interleave into that
    // row's natural order in the output (row r -> out + r * M * classes, or the split / epilogue addressing with q = r).  Allowed
    // on top: the row-twiddle epilogue w_N^(+-(r + q_offset) * k) (epi.kind 3 / 4, bq = 1), the per-row coset constant g^(r + q_offset)
    // (row_coset_const) and split_log / split_blk.
    uint64_t in_rows = 1;
    uint64_t in_pitch = 0;
    bool row_coset_const = false;
};

int ntt_tables_create(NttTables& T, int curve, hipStream_t stream);
void ntt_tables_trim(NttTables& T);      // release every table / plane that is rebuilt on demand (plonk_trim)
void ntt_tables_destroy(NttTables& T);
int ntt_run(NttTables& T, const NttCall& c, hipStream_t stream);
bool ntt_single_pass_inplace_ok(const NttTables& T, const NttCall& c);
std::vector<int> ntt_plan_widths(int log_m, int max_log_r);

int transpose_fr(const Fr* in, Fr* out, uint64_t rows, uint64_t cols, hipStream_t stream);

// ----------------------------------------------------------------------------------------------- MSM
struct MsmWorkspace {
    void* d_buf = nullptr;
    size_t bytes = 0;
    // per-context tuning knobs (plonk_set_option): tests and experiments only
    int slice_log = 26;       // "msm_slice_log": MSMs above 2^slice_log points run slice by slice (workspace sizing)
    int batch_max = 32;       // "msm_batch_max": scalar vectors per launch set of plonk_commit_many_dev (1 = one MSM at a time)
    int fused_order = 1;      // "msm_fused_order": the bucket-size histogram taken inside the level-2 sort and scanned inside the placement (round 5): 1 = for launches of >= 2^23 points, 2 = always, 0 = never
    int fused_y3 = 1;         // "msm_fused_y3": Y3 of the mixed addition under one Montgomery reduction (ec_lazy.hpp); 0 = two products
    int sort_stage_cap = 0;   // "msm_sort_stage_cap": > 0 caps the LDS staging buffer of the level-2 sort (entries; tests force its chunked path), 0 = what the LDS budget leaves
    int reduce_grid = 0;      // "msm_reduce_grid": 1 = the window reduction as row / column tree sums + bit sums (msm_grid_sums_kernel) instead of the running-sum pyramid; experiment, not measured yet
    int acc_persist = 4;      // "msm_acc_persist": workgroups per CU of the persistent bucket accumulation (0 = one lane per bucket over the whole grid; < 0: an absolute grid of that many workgroups, for tests)
};
void msm_ws_release(MsmWorkspace& ws);
// Fixed-base window table (msm_engine.hip: msm_table_kernel): T planes of `stride` points, plane t = 2^(c*G*t) * bases; a scalar vector's W windows
// accumulate into G bucket sets (window g + t*G reads plane t).
struct MsmTable {
    int c = 0;            // 0 = no table (plane 0 only)
    int W = 1;            // windows per scalar at this width
    int G = 1;            // bucket sets per scalar
    int T = 1;            // planes = ceil(W / G)
    uint64_t stride = 0;  // points per plane (= n_bases)
    bool force = false;   // msm_precompute = 2: use the table for every MSM it can serve, whatever the cost model says (tests, experiments)
};
// bases: device, RESIDENT FORM produced by bases_to_limbs() (64 B BN254 / 96 B BLS12-381 per point: msm_engine.hip, BaseRec); with a table,
// the pointer addresses plane 0 (+ the range start) and the other planes follow at multiples of tab.stride.
// scalars: device, canonical 8xu32 — or Montgomery-form Fr when scalars_mont (into_repr is then taken inside the digit kernel).
// out_jac: host, 3*Q*... written as X||Y||Z Montgomery u32 limbs.
int msm_run(int curve, const void* d_bases, const uint32_t* d_scalars, bool scalars_mont, size_t n, uint32_t* h_out_jac,
            MsmWorkspace& ws, int window_bits, const MsmTable& tab, hipStream_t stream);
// K scalar vectors (lens[k] valid entries, zero beyond) against the same bases[0 .. n): h_out_jac receives K Jacobian triples.
int msm_run_many(int curve, const void* d_bases, const uint32_t* const* d_scalars, const size_t* lens, int K, bool scalars_mont, size_t n,
                 uint32_t* h_out_jac, MsmWorkspace& ws, int window_bits, const MsmTable& tab, hipStream_t stream);
int msm_table_plan(int curve, size_t n, int mode, size_t budget_bytes, int force_c, int force_sets, int* W_out, int* G_out, int* T_out);
int msm_table_build(int curve, void* d_table, size_t n, size_t stride, int shift, int T, hipStream_t stream);
int msm_jac_add_host(int curve, const uint32_t* a, const uint32_t* b, uint32_t* out);
int msm_jac_to_affine_host(int curve, const uint32_t* jac, uint32_t* out_xy, int* is_inf);
int bases_convert_ark(int curve, const void* d_raw, size_t n, void* d_compact, unsigned long long* d_bad, hipStream_t stream);
// every base a curve point (or the (0, 0) read as infinity)?  PLONK_ERR_ARG naming the first offender otherwise.  d_bad: two device words
// {first offending index, reasons}; keep = they already hold the ark conversion's findings
int bases_check(int curve, const void* d_xy, size_t n, unsigned long long* d_bad, bool keep, const char* who, hipStream_t stream);
int bases_to_limbs(int curve, const void* d_xy, size_t n, void* d_out, hipStream_t stream);
size_t msm_limb_base_bytes(int curve);

// elementwise helpers (device pointers)
int fr_from_mont_dev(int curve, const Fr* in, Fr* out, size_t n, hipStream_t stream);
int field_op_dev(int curve, int field, int op, const void* a, const void* b, void* out, size_t n, hipStream_t stream);
int synth_fr_dev(int curve, uint64_t seed, Fr* out, size_t n, hipStream_t stream);
int synth_bases_dev(int curve, uint64_t seed, size_t unique, size_t n, void* d_out, hipStream_t stream);
int synth_bases_distinct_dev(int curve, uint64_t seed, size_t n, void* d_out, hipStream_t stream);
int synth_srs_dev(int curve, const uint64_t* tau_mont, size_t n, void* d_out, hipStream_t stream);
int synth_circuit_dev(int curve, uint64_t seed, size_t n, size_t num_inputs, const uint64_t* k_mont, const Fr& omega_n, void* d_wires, void* d_sel,
                      void* d_sigma, void* d_id_perm, void* d_perm_idx, void* d_pub, hipStream_t stream);

const FrParams& fr_params(int curve);

// ----------------------------------------------------------------------------------------------- O(n) prover steps (poly_ops.hip, quotient.hip)
int quotient_evals_run(NttTables& T, const plonk_quotient_inputs* in, size_t n, size_t m, const uint64_t* alpha, const uint64_t* beta,
                       const uint64_t* gamma, const uint64_t* k, uint32_t cls_stride, uint32_t cls_offset, void* d_out, hipStream_t stream);
size_t perm_product_scratch_bytes(size_t n);
int perm_product_run(NttTables& T, const void* const* wires, const void* id_perm, const void* perm_idx, const uint64_t* beta, const uint64_t* gamma,
                     size_t n_all, size_t j0, size_t cnt, void* d_out, void* scratch, hipStream_t stream);
int class_interleave_run(NttTables& T, const void* d_in, size_t classes, size_t size, size_t in_stride, int reverse, const uint64_t* scale, void* d_out,
                         hipStream_t stream);
size_t poly_scratch_bytes(size_t len);
int poly_eval_run(NttTables& T, const void* d_poly, size_t len, const uint64_t* point, uint64_t* out_host, void* scratch, hipStream_t stream);
int poly_lincomb_run(NttTables& T, size_t k, const void* const* polys, const size_t* lens, const uint64_t* coeffs, void* d_out, size_t out_len,
                     hipStream_t stream);
int poly_div_linear_run(NttTables& T, const void* d_poly, size_t len, const uint64_t* point, void* d_out, void* scratch, hipStream_t stream);
int blind_run(NttTables& T, void* d_poly, size_t n, const uint64_t* blinders, size_t k, hipStream_t stream);
int poly_degree_run(const void* d_poly, size_t len, int64_t* degree, void* scratch, hipStream_t stream);
size_t coset_scratch_bytes(size_t size);
int coset_eval_run(NttTables& T, const void* d_poly, size_t len, size_t size, const uint64_t* shift, void* d_out, void* scratch, hipStream_t stream);
int coset_interp_run(NttTables& T, void* d_evals, size_t size, const uint64_t* shift, const uint64_t* scale, size_t i0, size_t count, void* d_out,
                     void* scratch, hipStream_t stream);

// ----------------------------------------------------------------------------------------------- RCCL transport (comm_rccl.hip)
struct PlonkComm;
int comm_unique_id(void* out128);
int comm_create(PlonkComm** out, const void* id128, int rank, int world, int device);
void comm_destroy(PlonkComm* c);
int comm_rank(const PlonkComm* c);
int comm_world(const PlonkComm* c);
int comm_rccl_version();
int comm_alltoall(PlonkComm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t stream);
int comm_allgather(PlonkComm* c, const void* send, void* recv, size_t bytes, hipStream_t stream);
int comm_allgather_host(PlonkComm* c, const void* in, size_t bytes, void* out, hipStream_t stream);
