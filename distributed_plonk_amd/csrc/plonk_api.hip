// plonk_api.hip — the C ABI of include/plonk_hip.h: worker state (reference `State`, worker.rs:42-59)
// as an opaque per-GPU context, and one entry point per PlonkSlave / PlonkPeer method.
#include <cstdarg>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <algorithm>
#include <vector>

#include "constants.h"
#include "ec.hpp"
#include "plonk_internal.hpp"


// ---------------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = {0};
char* plonk_last_error_buf() { return g_err; }
int plonk_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* plonk_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------- per-kernel timing
KernelProfiler& kernel_profiler() { static KernelProfiler p; return p; }
void KernelProfiler::resolve() {
    std::lock_guard<std::mutex> g(mu);
    for (auto& r : pending) {
        float ms = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& t = totals[r.name];
            t.first += ms; t.second += 1;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    pending.clear();
}
void KernelProfiler::reset() { resolve(); std::lock_guard<std::mutex> g(mu); totals.clear(); }

// ---------------------------------------------------------------------------------------------- context
struct FftTask {          // reference FftTask, worker.rs:32-40
    bool is_quot = false, is_inv = false, is_coset = false;
    std::vector<plonk_fft_workload> wl;
    size_t me = 0;
    int log_n = 0;
    uint64_t r = 0, c = 0, nrows = 0, ncols = 0;
    Fr* d_rows = nullptr;      // [nrows][c]      (owned unless rows_external)
    bool rows_external = false;
    uint64_t compact_len = 0;  // > 0: d_rows is [nrows][compact_len], the leading coefficients of zero-padded rows (plonk_fft1_dev_compact)
    Fr* d_work = nullptr;      // row-pass workspace of the compact form
    Fr* d_send = nullptr;      // S blocks of [nrows][ncols]
    Fr* d_recv = nullptr;      // [r][ncols]
    std::vector<uint8_t> row_present;
    uint64_t rows_filled = 0;
    bool prepared = false;
};

struct plonk_ctx {
    int device = 0, curve = 0;
    hipStream_t stream = nullptr;
    NttTables tables;
    // SRS (State.bases), kept in the MSM's resident limb form (flimb.hpp)
    void* d_bases = nullptr;                    // plane 0 of the fixed-base window table (msm_table) when one is built
    size_t n_bases = 0;
    MsmTable msm_table;
    int msm_precompute = 0;                     // 0 off (default: measured a wash on MI355X, DESIGN.md §4.2, profiles/r03_msm_table_experiment.txt), 1 when the cost model likes it, 2 always
    size_t msm_table_budget = (size_t)64 << 30;
    int msm_table_c = 0, msm_table_sets = 0;    // > 0: pin the table's window width / bucket sets per scalar (tests, experiments); 0: the cost model
    // domains (State.domain / quot_domain and their r/c splits are derived on demand)
    size_t domain_size = 0, quot_domain_size = 0;
    std::map<uint64_t, FftTask> tasks;          // State.fft_tasks
    Fr* d_wire = nullptr;                       // State.wire
    size_t wire_len = 0;
    // scratch
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    void* d_scratch2 = nullptr;
    size_t scratch2_bytes = 0;
    MsmWorkspace msm_ws;
    int msm_window = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    // small free-list of exchange buffers so that back-to-back transforms do not hipMalloc/hipFree
    std::vector<std::pair<size_t, void*>> pool;
    PlonkComm* comm = nullptr;                  // RCCL communicator (plonk_comm_init), owned
    int comm_ordinal = 0;                       // this communicator's place among the device's communicators, in creation order (comm_order_check)
    int check_bases = 1;                        // init: every base must be a curve point (option "check_bases")
    unsigned long long* d_bad = nullptr;        // two words for that check's verdict
};

// state of the collective order check (comm_order_check, further down)
namespace {
struct DeviceCollectives {
    plonk_ctx* control = nullptr;        // the context whose communicator carries the tags (the oldest LIVE communicator of this device)
    std::vector<plonk_ctx*> live;        // contexts of this device that hold a communicator, in creation order (the control's successors)
    int created = 0;                     // communicators created so far = the next one's ordinal
    uint64_t seq = 0;                    // CHECKED collectives entered (a tag was exchanged)
};
std::mutex g_coll_mu;
DeviceCollectives g_coll[64];
int g_comm_check = -1;                   // -1: not read yet (PLONK_COMM_CHECK_ORDER)
bool comm_check_on() {
    if (g_comm_check < 0) { const char* e = getenv("PLONK_COMM_CHECK_ORDER"); g_comm_check = (e && *e && *e != '0') ? 1 : 0; }
    return g_comm_check == 1;
}
const char* coll_name(uint64_t k) { return k == 1 ? "all-to-all" : k == 2 ? "all-gather" : k == 3 ? "all-gather (host)" : "?"; }
}  // namespace
static void comm_forget(plonk_ctx* ctx);

static int pool_get(plonk_ctx* ctx, size_t bytes, void** out) {
    for (size_t i = 0; i < ctx->pool.size(); i++)
        if (ctx->pool[i].first == bytes) {
            *out = ctx->pool[i].second;
            ctx->pool.erase(ctx->pool.begin() + i);
            return PLONK_OK;
        }
    HIP_TRY(hipMalloc(out, bytes));
    return PLONK_OK;
}
static void pool_put(plonk_ctx* ctx, size_t bytes, void* p) {
    if (!p) return;
    if (ctx->pool.size() >= 6) {
        (void)hipFree(ctx->pool.front().second);
        ctx->pool.erase(ctx->pool.begin());
    }
    ctx->pool.push_back({bytes, p});
}

static size_t aff_bytes(int curve) { return curve == PLONK_BN254 ? 64 : 96; }
static size_t jac_bytes(int curve) { return curve == PLONK_BN254 ? 96 : 144; }
static size_t ark_aff_bytes(int curve) { return curve == PLONK_BN254 ? 72 : 104; }

static int ilog2_exact(size_t n) {
    if (n == 0 || (n & (n - 1))) return -1;
    int l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

static int ensure_scratch(plonk_ctx* ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return PLONK_OK;
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr; ctx->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&ctx->d_scratch, bytes));
    ctx->scratch_bytes = bytes;
    return PLONK_OK;
}
static int ensure_scratch2(plonk_ctx* ctx, size_t bytes) {
    if (ctx->scratch2_bytes >= bytes) return PLONK_OK;
    if (ctx->d_scratch2) (void)hipFree(ctx->d_scratch2);
    ctx->d_scratch2 = nullptr; ctx->scratch2_bytes = 0;
    HIP_TRY(hipMalloc(&ctx->d_scratch2, bytes));
    ctx->scratch2_bytes = bytes;
    return PLONK_OK;
}

#define CHECK_CTX(ctx)                                                  \
    do {                                                                \
        if (!(ctx)) return plonk_fail(PLONK_ERR_ARG, "null context");   \
        HIP_TRY(hipSetDevice((ctx)->device));                           \
    } while (0)

static void free_task(plonk_ctx* ctx, FftTask& t) {
    const size_t tile_bytes = t.nrows * t.c * 32;
    if (t.d_rows && !t.rows_external) pool_put(ctx, tile_bytes, t.d_rows);
    if (t.d_work) pool_put(ctx, tile_bytes, t.d_work);
    t.d_work = nullptr;
    if (t.d_send) pool_put(ctx, tile_bytes, t.d_send);
    if (t.d_recv && t.d_recv != t.d_send) pool_put(ctx, tile_bytes, t.d_recv);
    t.d_rows = t.d_send = t.d_recv = nullptr;
}

extern "C" int plonk_create(plonk_ctx** out, int device, int curve) {
    if (!out) return plonk_fail(PLONK_ERR_ARG, "plonk_create: null out");
    if (curve != PLONK_BN254 && curve != PLONK_BLS12_381) return plonk_fail(PLONK_ERR_ARG, "plonk_create: unknown curve %d", curve);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return plonk_fail(PLONK_ERR_HIP, "plonk_create: device %d not present (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    struct Partial {                // a context that fails half-way gives its stream / events / tables back
        void operator()(plonk_ctx* c) const {
            ntt_tables_destroy(c->tables);
            if (c->ev0) (void)hipEventDestroy(c->ev0);
            if (c->ev1) (void)hipEventDestroy(c->ev1);
            if (c->stream) (void)hipStreamDestroy(c->stream);
            delete c;
        }
    };
    std::unique_ptr<plonk_ctx, Partial> ctx(new plonk_ctx());
    ctx->device = device;
    ctx->curve = curve;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&ctx->ev0));
    HIP_TRY(hipEventCreate(&ctx->ev1));
    int rc = ntt_tables_create(ctx->tables, curve, ctx->stream);
    if (rc) return rc;
    *out = ctx.release();
    return PLONK_OK;
}

extern "C" void plonk_destroy(plonk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->tasks) free_task(ctx, kv.second);
    for (auto& pb : ctx->pool) (void)hipFree(pb.second);
    comm_forget(ctx);
    comm_destroy(ctx->comm);
    ntt_tables_destroy(ctx->tables);
    if (ctx->d_bases) (void)hipFree(ctx->d_bases);
    if (ctx->d_wire) (void)hipFree(ctx->d_wire);
    if (ctx->d_bad) (void)hipFree(ctx->d_bad);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_scratch2) (void)hipFree(ctx->d_scratch2);
    msm_ws_release(ctx->msm_ws);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" void* plonk_stream(plonk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
extern "C" int plonk_sync(plonk_ctx* ctx) {
    CHECK_CTX(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

extern "C" int plonk_set_option(plonk_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return plonk_fail(PLONK_ERR_ARG, "plonk_set_option: null");
    if (!strcmp(key, "msm_window")) { ctx->msm_window = (int)value; return PLONK_OK; }
    if (!strcmp(key, "comm_check_order")) { g_comm_check = value ? 1 : 0; return PLONK_OK; }          // PROCESS-wide (every rank must agree); env PLONK_COMM_CHECK_ORDER
    if (!strcmp(key, "check_bases")) { ctx->check_bases = value ? 1 : 0; return PLONK_OK; }            // default 1; takes effect at the next init
    if (!strcmp(key, "msm_precompute")) { ctx->msm_precompute = (int)value; return PLONK_OK; }      // takes effect at the next init
    if (!strcmp(key, "msm_table_c")) {                                                                // takes effect at the next init
        // 0 = the plan's choice; a pinned width must be one the table plan considers (msm_engine.hip: MSM_TABLE_MAX_C = 21).  A width that is valid
        // but unusable for the SRS at hand (top window nearly empty, sort geometry, budget) falls back to "no table" at init, also in mode 2.
        if (value != 0 && (value < 4 || value > 21)) return plonk_fail(PLONK_ERR_ARG, "msm_table_c = %lld (0 or 4..21)", (long long)value);
        ctx->msm_table_c = (int)value;
        return PLONK_OK;
    }
    if (!strcmp(key, "msm_table_sets")) { ctx->msm_table_sets = (int)value; return PLONK_OK; }        // takes effect at the next init
    if (!strcmp(key, "msm_table_budget_mib")) { ctx->msm_table_budget = (size_t)value << 20; return PLONK_OK; }
    // every knob lives in the context it was set on (another context, possibly driven from another host thread, is not affected)
    if (!strcmp(key, "ntt_max_log_r")) { ctx->tables.max_log_r = (int)value; return PLONK_OK; }
    if (!strcmp(key, "msm_batch_max")) { ctx->msm_ws.batch_max = (int)value; return PLONK_OK; }      // vectors per launch set of commit_many
    if (!strcmp(key, "msm_fused_order")) { ctx->msm_ws.fused_order = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return PLONK_OK; }   // 1 = large launches (default), 2 = always, 0 = never
    if (!strcmp(key, "msm_fused_y3")) { ctx->msm_ws.fused_y3 = value ? 1 : 0; return PLONK_OK; }      // default 1
    if (!strcmp(key, "msm_sort_stage_cap")) { ctx->msm_ws.sort_stage_cap = (int)std::max<int64_t>(0, value); return PLONK_OK; }   // tests: force the chunked level-2 sort
    if (!strcmp(key, "msm_acc_persist")) { ctx->msm_ws.acc_persist = (int)std::max<int64_t>(-65536, std::min<int64_t>(value, 8)); return PLONK_OK; }   // default 4; < 0: an absolute grid of -value workgroups (tests)
    if (!strcmp(key, "msm_reduce_grid")) { ctx->msm_ws.reduce_grid = value ? 1 : 0; return PLONK_OK; }   // experiment: grid reduction (msm_engine.hip, 5b); default 0
    if (!strcmp(key, "ntt_shoup")) {              // precomputed-quotient butterflies (ntt_kernels.hpp); the environment variable PLONK_NTT_NO_SHOUP only sets the initial value
        ctx->tables.use_shoup = value != 0;
        return PLONK_OK;
    }
    if (!strcmp(key, "quotient_fuse")) { ctx->tables.quotient_fuse = (int)value; return PLONK_OK; }   // experiments, see quotient.hip
    if (!strcmp(key, "msm_slice_log")) { ctx->msm_ws.slice_log = (int)value; return PLONK_OK; }       // MSMs above 2^value points are sliced (8..26)
    return plonk_fail(PLONK_ERR_ARG, "plonk_set_option: unknown key %s", key);
}

extern "C" int plonk_last_kernel_ms(plonk_ctx* ctx, double* out_ms) {
    CHECK_CTX(ctx);
    if (!out_ms || !ctx->ev_valid) return plonk_fail(PLONK_ERR_STATE, "no timed call yet");
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *out_ms = ms;
    return PLONK_OK;
}

static int check_domain(plonk_ctx* ctx, size_t size, const char* what) {
    if (size == 0) return PLONK_OK;
    int l = ilog2_exact(size);
    if (l < 0) return plonk_fail(PLONK_ERR_DOMAIN, "%s %zu is not a power of two", what, size);
    if (l > ctx->tables.two_adicity) return plonk_fail(PLONK_ERR_DOMAIN, "%s 2^%d exceeds the field's two-adicity %d", what, l, ctx->tables.two_adicity);
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- init @0
static int set_domains(plonk_ctx* ctx, size_t domain_size, size_t quot_domain_size) {
    // Radix2EvaluationDomain::new rounds up to a power of two (worker.rs:143,150); fail like
    // DomainCreationError when it does not exist.
    auto round_up = [](size_t n) { size_t s = 1; while (s < n) s <<= 1; return n ? s : 0; };
    domain_size = round_up(domain_size);
    quot_domain_size = round_up(quot_domain_size);
    int rc;
    if ((rc = check_domain(ctx, domain_size, "domain_size"))) return rc;
    if ((rc = check_domain(ctx, quot_domain_size, "quot_domain_size"))) return rc;
    ctx->domain_size = domain_size;
    ctx->quot_domain_size = quot_domain_size;
    return PLONK_OK;
}

// SRS -> resident limb form, plus the fixed-base window table when it pays off (one-time work per `init`)
static int bad_words(plonk_ctx* ctx, bool reset) {
    if (!ctx->d_bad) HIP_TRY(hipMalloc((void**)&ctx->d_bad, 16));
    if (reset) {
        static const unsigned long long init[2] = {~0ull, 0ull};
        HIP_TRY(hipMemcpyAsync(ctx->d_bad, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    }
    return PLONK_OK;
}

// `flags_seen`: the ark conversion has already left its findings (bad infinity bytes) in ctx->d_bad
static int install_bases(plonk_ctx* ctx, const void* d_xy, size_t n_bases, const char* who, bool flags_seen = false) {
    if (ctx->check_bases) {                     // before anything is installed: a refused SRS leaves the context without bases
        int rc = bad_words(ctx, false);
        if (!rc) rc = bases_check(ctx->curve, d_xy, n_bases, ctx->d_bad, flags_seen, who, ctx->stream);
        if (rc) return rc;
    }
    int W = 1, G = 1, T = 1;
    const int c = msm_table_plan(ctx->curve, n_bases, ctx->msm_precompute, ctx->msm_table_budget, ctx->msm_table_c, ctx->msm_table_sets, &W, &G, &T);
    const size_t pb = msm_limb_base_bytes(ctx->curve);
    if (hipMalloc(&ctx->d_bases, n_bases * pb * (size_t)T) != hipSuccess) {
        (void)hipGetLastError();
        T = 1;                                  // no room for the table: plain bases
        HIP_TRY(hipMalloc(&ctx->d_bases, n_bases * pb));
    }
    int rc = bases_to_limbs(ctx->curve, d_xy, n_bases, ctx->d_bases, ctx->stream);
    ctx->msm_table = MsmTable();
    if (!rc && T > 1) {
        rc = msm_table_build(ctx->curve, ctx->d_bases, n_bases, n_bases, c * G, T, ctx->stream);
        ctx->msm_table.c = c; ctx->msm_table.W = W; ctx->msm_table.G = G; ctx->msm_table.T = T; ctx->msm_table.stride = n_bases;
        ctx->msm_table.force = ctx->msm_precompute == 2;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (!rc) ctx->n_bases = n_bases;
    return rc;
}

extern "C" int plonk_init(plonk_ctx* ctx, const void* bases, size_t n_bases, int base_layout, size_t domain_size, size_t quot_domain_size) {
    CHECK_CTX(ctx);
    if (n_bases && !bases) return plonk_fail(PLONK_ERR_ARG, "plonk_init: null bases");
    if (base_layout != PLONK_BASES_XY && base_layout != PLONK_BASES_ARK) return plonk_fail(PLONK_ERR_ARG, "plonk_init: layout %d", base_layout);
    int rc = set_domains(ctx, domain_size, quot_domain_size);
    if (rc) return rc;
    if (ctx->d_bases) (void)hipFree(ctx->d_bases);
    ctx->d_bases = nullptr; ctx->n_bases = 0; ctx->msm_table = MsmTable();
    if (n_bases) {
        const size_t ab = aff_bytes(ctx->curve);
        struct Staging {            // the (x, y) staging copy is released on every exit, after the stream has finished with it
            void* p = nullptr; hipStream_t s;
            ~Staging() { if (p) { (void)hipStreamSynchronize(s); (void)hipFree(p); } }
        } d_xy;
        d_xy.s = ctx->stream;
        HIP_TRY(hipMalloc(&d_xy.p, n_bases * ab));
        if (base_layout == PLONK_BASES_XY) {
            HIP_TRY(hipMemcpyAsync(d_xy.p, bases, n_bases * ab, hipMemcpyHostToDevice, ctx->stream));
        } else {
            const size_t rb = ark_aff_bytes(ctx->curve) * n_bases;
            if ((rc = ensure_scratch(ctx, rb))) return rc;
            HIP_TRY(hipMemcpyAsync(ctx->d_scratch, bases, rb, hipMemcpyHostToDevice, ctx->stream));
            if (ctx->check_bases && (rc = bad_words(ctx, true))) return rc;
            if ((rc = bases_convert_ark(ctx->curve, ctx->d_scratch, n_bases, d_xy.p, ctx->check_bases ? ctx->d_bad : nullptr, ctx->stream))) return rc;
        }
        if ((rc = install_bases(ctx, d_xy.p, n_bases, "plonk_init", base_layout == PLONK_BASES_ARK))) return rc;
    }
    return PLONK_OK;
}

extern "C" int plonk_init_dev(plonk_ctx* ctx, const void* d_bases_xy, size_t n_bases, size_t domain_size, size_t quot_domain_size) {
    CHECK_CTX(ctx);
    if (n_bases && !d_bases_xy) return plonk_fail(PLONK_ERR_ARG, "plonk_init_dev: null bases");
    int rc = set_domains(ctx, domain_size, quot_domain_size);
    if (rc) return rc;
    if (ctx->d_bases) (void)hipFree(ctx->d_bases);
    ctx->d_bases = nullptr; ctx->n_bases = 0; ctx->msm_table = MsmTable();
    if (n_bases) {
        if ((rc = install_bases(ctx, d_bases_xy, n_bases, "plonk_init_dev"))) return rc;
    }
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- varMsm @1
static int msm_device(plonk_ctx* ctx, size_t start, size_t n, const uint32_t* d_scalars, uint64_t* out_jac, bool scalars_mont = false) {
    if (n == 0) {       // empty MSM = zero (1,1,0)
        uint64_t a[18], b[18];
        memset(a, 0, sizeof a); memset(b, 0, sizeof b);
        return msm_jac_add_host(ctx->curve, (uint32_t*)a, (uint32_t*)b, (uint32_t*)out_jac);
    }
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    int rc = msm_run(ctx->curve, (const char*)ctx->d_bases + start * msm_limb_base_bytes(ctx->curve), d_scalars, scalars_mont, n, (uint32_t*)out_jac, ctx->msm_ws,
                     ctx->msm_window, ctx->msm_table, ctx->stream);
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    return rc;
}

extern "C" int plonk_var_msm(plonk_ctx* ctx, const plonk_msm_workload* wl, const uint64_t* scalars, size_t n_scalars, uint64_t* out_jacobian) {
    CHECK_CTX(ctx);
    if (!wl || !out_jacobian || (n_scalars && !scalars)) return plonk_fail(PLONK_ERR_ARG, "plonk_var_msm: null argument");
    if (wl->start > wl->end || wl->end > ctx->n_bases)
        return plonk_fail(PLONK_ERR_ARG, "plonk_var_msm: range [%llu,%llu) outside the %zu resident bases", (unsigned long long)wl->start,
                          (unsigned long long)wl->end, ctx->n_bases);
    const size_t n = std::min<size_t>(wl->end - wl->start, n_scalars);     // multi_scalar_mul takes min(len)
    int rc = ensure_scratch(ctx, std::max<size_t>(n, 1) * 32);
    if (rc) return rc;
    if (n) HIP_TRY(hipMemcpyAsync(ctx->d_scratch, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    return msm_device(ctx, wl->start, n, (const uint32_t*)ctx->d_scratch, out_jacobian);
}

extern "C" int plonk_msm_dev(plonk_ctx* ctx, size_t start, size_t end, const void* d_scalars, uint64_t* out_jacobian) {
    CHECK_CTX(ctx);
    if (!out_jacobian || (!d_scalars && end > start)) return plonk_fail(PLONK_ERR_ARG, "plonk_msm_dev: null argument");
    if (start > end || end > ctx->n_bases) return plonk_fail(PLONK_ERR_ARG, "plonk_msm_dev: range outside the resident bases");
    return msm_device(ctx, start, end - start, (const uint32_t*)d_scalars, out_jacobian);
}

// commit_polynomial: into_repr + zero-pad to the SRS length (zero scalars add nothing) + MSM
extern "C" int plonk_commit_dev(plonk_ctx* ctx, const void* d_coeffs_mont, size_t n_coeffs, uint64_t* out_jacobian) {
    CHECK_CTX(ctx);
    if (!out_jacobian || (n_coeffs && !d_coeffs_mont)) return plonk_fail(PLONK_ERR_ARG, "plonk_commit_dev: null argument");
    const size_t n = std::min(n_coeffs, ctx->n_bases);
    return msm_device(ctx, 0, n, (const uint32_t*)d_coeffs_mont, out_jacobian, /*scalars_mont=*/true);     // into_repr inside the digit kernel
}

extern "C" int plonk_commit_range_dev(plonk_ctx* ctx, const void* d_coeffs_mont, size_t start, size_t count, uint64_t* out_jacobian) {
    CHECK_CTX(ctx);
    if (!out_jacobian || (count && !d_coeffs_mont)) return plonk_fail(PLONK_ERR_ARG, "plonk_commit_range_dev: null argument");
    if (start > ctx->n_bases) return plonk_fail(PLONK_ERR_ARG, "plonk_commit_range_dev: start %zu beyond the %zu resident bases", start, ctx->n_bases);
    const size_t n = std::min(count, ctx->n_bases - start);
    return msm_device(ctx, start, n, (const uint32_t*)d_coeffs_mont, out_jacobian, /*scalars_mont=*/true);
}

// k independent commit_polynomial calls against the same key in one set of launches (the commitments of a prover round:
// dispatcher2.rs:313-321 five wires, :519-531 five quotient parts, :690-697 two openings)
extern "C" int plonk_commit_many_dev(plonk_ctx* ctx, size_t k, const void* const* d_coeffs_mont, const size_t* n_coeffs, size_t start, uint64_t* out_jacobians) {
    CHECK_CTX(ctx);
    if (k == 0) return PLONK_OK;
    if (!d_coeffs_mont || !n_coeffs || !out_jacobians) return plonk_fail(PLONK_ERR_ARG, "plonk_commit_many_dev: null argument");
    if (k > 4096) return plonk_fail(PLONK_ERR_ARG, "plonk_commit_many_dev: %zu polynomials", k);
    if (start > ctx->n_bases) return plonk_fail(PLONK_ERR_ARG, "plonk_commit_many_dev: start %zu beyond the %zu resident bases", start, ctx->n_bases);
    std::vector<const uint32_t*> ptrs(k);
    std::vector<size_t> lens(k);
    size_t n = 0;
    for (size_t i = 0; i < k; i++) {
        if (n_coeffs[i] && !d_coeffs_mont[i]) return plonk_fail(PLONK_ERR_ARG, "plonk_commit_many_dev: polynomial %zu is null", i);
        lens[i] = std::min(n_coeffs[i], ctx->n_bases - start);
        ptrs[i] = (const uint32_t*)d_coeffs_mont[i];
        n = std::max(n, lens[i]);
    }
    const size_t jb = jac_bytes(ctx->curve);
    if (n == 0) {                       // every polynomial empty: k zeros
        for (size_t i = 0; i < k; i++) {
            int rc = msm_device(ctx, 0, 0, nullptr, (uint64_t*)((char*)out_jacobians + i * jb));
            if (rc) return rc;
        }
        return PLONK_OK;
    }
    for (size_t i = 0; i < k; i++)
        if (!ptrs[i]) ptrs[i] = ptrs[0] ? ptrs[0] : (const uint32_t*)ctx->d_bases;      // never dereferenced: its length is 0
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    int rc = msm_run_many(ctx->curve, (const char*)ctx->d_bases + start * msm_limb_base_bytes(ctx->curve), ptrs.data(), lens.data(), (int)k, /*scalars_mont=*/true, n,
                          (uint32_t*)out_jacobians, ctx->msm_ws, ctx->msm_window, ctx->msm_table, ctx->stream);
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    return rc;
}

extern "C" int plonk_commit(plonk_ctx* ctx, const uint64_t* coeffs_mont, size_t n_coeffs, uint64_t* out_jacobian) {
    CHECK_CTX(ctx);
    if (!out_jacobian || (n_coeffs && !coeffs_mont)) return plonk_fail(PLONK_ERR_ARG, "plonk_commit: null argument");
    const size_t n = std::min(n_coeffs, ctx->n_bases);
    int rc = ensure_scratch(ctx, std::max<size_t>(n, 1) * 32);
    if (rc) return rc;
    if (n) HIP_TRY(hipMemcpyAsync(ctx->d_scratch, coeffs_mont, n * 32, hipMemcpyHostToDevice, ctx->stream));
    return plonk_commit_dev(ctx, ctx->d_scratch, n, out_jacobian);
}

extern "C" int plonk_g1_add(int curve, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    if (!a || !b || !out) return plonk_fail(PLONK_ERR_ARG, "plonk_g1_add: null");
    return msm_jac_add_host(curve, (const uint32_t*)a, (const uint32_t*)b, (uint32_t*)out);
}
// Keccak-f[1600] on a 200-byte state (lane (x, y) at byte offset 8 * (x + 5 y), little-endian): the permutation under merlin's STROBE-128, i.e.
// under the reference's Fiat-Shamir transcript (dispatcher2.rs:44-154).  Host-only and tiny, like the point helpers above: a proof's transcript is
// ~25 permutations — 7-15 ms of a host interpreter's time per proof (5 % of an 8-rank proof), microseconds here.
extern "C" int plonk_keccak_f1600(uint8_t* state200) {
    if (!state200) return plonk_fail(PLONK_ERR_ARG, "plonk_keccak_f1600: null");
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
                                    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
                                    0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
                                    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                                    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};      // [x][y]
    auto rol = [](uint64_t v, int r) { r &= 63; return r ? (v << r) | (v >> (64 - r)) : v; };
    uint64_t a[5][5];
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) {
            uint64_t v = 0;
            for (int b = 7; b >= 0; b--) v = (v << 8) | state200[8 * (x + 5 * y) + b];
            a[x][y] = v;
        }
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], d[5], bb[5][5];
        for (int x = 0; x < 5; x++) c[x] = a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) a[x][y] ^= d[x];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) bb[y][(2 * x + 3 * y) % 5] = rol(a[x][y], ROT[x][y]);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) a[x][y] = bb[x][y] ^ (~bb[(x + 1) % 5][y] & bb[(x + 2) % 5][y]);
        a[0][0] ^= RC[rnd];
    }
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++)
            for (int b = 0; b < 8; b++) state200[8 * (x + 5 * y) + b] = (uint8_t)(a[x][y] >> (8 * b));
    return PLONK_OK;
}

extern "C" int plonk_g1_to_affine(int curve, const uint64_t* jac, uint64_t* out_xy, int* is_infinity) {
    if (!jac || !out_xy || !is_infinity) return plonk_fail(PLONK_ERR_ARG, "plonk_g1_to_affine: null");
    return msm_jac_to_affine_host(curve, (const uint32_t*)jac, (uint32_t*)out_xy, is_infinity);
}

// ---------------------------------------------------------------------------------------------- whole-vector NTT
extern "C" int plonk_ntt_dev(plonk_ctx* ctx, void* d_in, void* d_out, size_t n, int is_inv, int is_coset) {
    CHECK_CTX(ctx);
    if (!d_in || !d_out) return plonk_fail(PLONK_ERR_ARG, "plonk_ntt_dev: null");
    const int log_n = ilog2_exact(n);
    if (log_n < 0) return plonk_fail(PLONK_ERR_DOMAIN, "plonk_ntt_dev: size %zu is not a power of two", n);
    if (log_n > ctx->tables.two_adicity) return plonk_fail(PLONK_ERR_DOMAIN, "plonk_ntt_dev: 2^%d exceeds two-adicity", log_n);
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    if (log_n == 0) {
        if (d_in != d_out) HIP_TRY(hipMemcpyAsync(d_out, d_in, 32, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        NttCall c;
        c.in = (const Fr*)d_in;
        c.log_m = log_n;
        c.batch = 1;
        c.inverse = is_inv != 0;
        if (is_coset && !is_inv) { c.pro.kind = 1; c.pro.b0 = 1; }      // x[n] * g^n
        if (is_coset && is_inv) { c.epi.kind = 2; c.epi.b0 = 1; }       // X[k] * g^-k
        Fr* out = (Fr*)d_out;
        bool via_scratch = false;
        if (d_in == d_out && !ntt_single_pass_inplace_ok(ctx->tables, c)) {
            int rc = ensure_scratch(ctx, n * 32);
            if (rc) return rc;
            out = (Fr*)ctx->d_scratch;
            via_scratch = true;
        }
        c.out = out;
        int rc = ntt_run(ctx->tables, c, ctx->stream);
        if (rc) return rc;
        if (via_scratch) HIP_TRY(hipMemcpyAsync(d_out, out, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    return PLONK_OK;
}

extern "C" int plonk_ntt(plonk_ctx* ctx, uint64_t* v, size_t n, int is_inv, int is_coset) {
    CHECK_CTX(ctx);
    if (!v) return plonk_fail(PLONK_ERR_ARG, "plonk_ntt: null");
    if (ilog2_exact(n) < 0) return plonk_fail(PLONK_ERR_DOMAIN, "plonk_ntt: size %zu is not a power of two", n);
    int rc = ensure_scratch2(ctx, 2 * n * 32);
    if (rc) return rc;
    Fr* a = (Fr*)ctx->d_scratch2;
    Fr* b = a + n;
    HIP_TRY(hipMemcpyAsync(a, v, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = plonk_ntt_dev(ctx, a, b, n, is_inv, is_coset))) return rc;
    HIP_TRY(hipMemcpyAsync(v, b, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- transpose
extern "C" int plonk_transpose_dev(plonk_ctx* ctx, const void* d_in, void* d_out, size_t rows, size_t cols) {
    CHECK_CTX(ctx);
    if (!d_in || !d_out) return plonk_fail(PLONK_ERR_ARG, "plonk_transpose_dev: null");
    return transpose_fr((const Fr*)d_in, (Fr*)d_out, rows, cols, ctx->stream);
}
extern "C" int plonk_transpose(plonk_ctx* ctx, uint64_t* v, size_t rows, size_t cols) {
    CHECK_CTX(ctx);
    if (!v) return plonk_fail(PLONK_ERR_ARG, "plonk_transpose: null");
    const size_t n = rows * cols;
    if (n == 0) return PLONK_OK;
    int rc = ensure_scratch2(ctx, 2 * n * 32);
    if (rc) return rc;
    Fr* a = (Fr*)ctx->d_scratch2;
    Fr* b = a + n;
    HIP_TRY(hipMemcpyAsync(a, v, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = transpose_fr(a, b, rows, cols, ctx->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(v, b, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- fftInit @2
extern "C" int plonk_fft_init(plonk_ctx* ctx, uint64_t id, const plonk_fft_workload* wl, size_t n_wl, size_t me, int is_quot, int is_inv,
                              int is_coset) {
    CHECK_CTX(ctx);
    if (!wl || n_wl == 0 || me >= n_wl) return plonk_fail(PLONK_ERR_ARG, "plonk_fft_init: bad workloads (n=%zu, me=%zu)", n_wl, me);
    const size_t N = is_quot ? ctx->quot_domain_size : ctx->domain_size;
    const int log_n = ilog2_exact(N);
    if (N == 0 || log_n < 2) return plonk_fail(PLONK_ERR_DOMAIN, "plonk_fft_init: %s domain of size %zu not initialised / too small", is_quot ? "quot" : "main", N);
    FftTask t;
    t.is_quot = is_quot; t.is_inv = is_inv; t.is_coset = is_coset;
    t.wl.assign(wl, wl + n_wl);
    t.me = me;
    t.log_n = log_n;
    t.r = (uint64_t)1 << (log_n >> 1);           // worker.rs:144
    t.c = N / t.r;
    t.nrows = wl[me].row_end - wl[me].row_start;
    t.ncols = wl[me].col_end - wl[me].col_start;
    // the all-to-all needs the reference's even contiguous partition (dispatcher2.rs:272-291)
    for (size_t s = 0; s < n_wl; s++) {
        if (wl[s].row_start != s * t.nrows || wl[s].row_end != (s + 1) * t.nrows || wl[s].col_start != s * t.ncols ||
            wl[s].col_end != (s + 1) * t.ncols)
            return plonk_fail(PLONK_ERR_ARG, "plonk_fft_init: workload %zu is not the even contiguous partition", s);
    }
    if (t.nrows * n_wl != t.r || t.ncols * n_wl != t.c || ilog2_exact(t.nrows) < 0 || ilog2_exact(t.ncols) < 0)
        return plonk_fail(PLONK_ERR_ARG, "plonk_fft_init: %zu workers do not evenly split r=%llu, c=%llu", n_wl, (unsigned long long)t.r,
                          (unsigned long long)t.c);
    auto old = ctx->tasks.find(id);
    if (old != ctx->tasks.end()) { free_task(ctx, old->second); ctx->tasks.erase(old); }   // HashMap::insert replaces
    t.row_present.assign(t.nrows, 0);
    ctx->tasks[id] = t;
    return PLONK_OK;
}

static int get_task(plonk_ctx* ctx, uint64_t id, FftTask** out) {
    auto it = ctx->tasks.find(id);
    if (it == ctx->tasks.end()) return plonk_fail(PLONK_ERR_STATE, "unknown fft task id %llu", (unsigned long long)id);
    *out = &it->second;
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- fft1 @3
extern "C" int plonk_fft1(plonk_ctx* ctx, uint64_t id, uint64_t i, const uint64_t* v, size_t len) {
    CHECK_CTX(ctx);
    FftTask* t;
    int rc = get_task(ctx, id, &t);
    if (rc) return rc;
    if (!v || len != t->c) return plonk_fail(PLONK_ERR_ARG, "plonk_fft1: row length %zu != c = %llu", len, (unsigned long long)t->c);
    if (i >= t->nrows) return plonk_fail(PLONK_ERR_ARG, "plonk_fft1: local row %llu >= %llu", (unsigned long long)i, (unsigned long long)t->nrows);
    if (t->prepared || t->rows_external) return plonk_fail(PLONK_ERR_STATE, "plonk_fft1: rows already consumed");
    if (!t->d_rows) { int prc = pool_get(ctx, t->nrows * t->c * 32, (void**)&t->d_rows); if (prc) return prc; }
    t->compact_len = 0;          // dense rows
    HIP_TRY(hipMemcpyAsync(t->d_rows + i * t->c, v, t->c * 32, hipMemcpyHostToDevice, ctx->stream));
    if (!t->row_present[i]) { t->row_present[i] = 1; t->rows_filled++; }
    return PLONK_OK;
}

extern "C" int plonk_fft1_dev(plonk_ctx* ctx, uint64_t id, void* d_rows) {
    CHECK_CTX(ctx);
    FftTask* t;
    int rc = get_task(ctx, id, &t);
    if (rc) return rc;
    if (!d_rows) return plonk_fail(PLONK_ERR_ARG, "plonk_fft1_dev: null");
    if (t->prepared) return plonk_fail(PLONK_ERR_STATE, "plonk_fft1_dev: already prepared");
    if (t->d_rows && !t->rows_external) pool_put(ctx, t->nrows * t->c * 32, t->d_rows);
    t->d_rows = (Fr*)d_rows;     // consumed: the row pass uses it as workspace
    t->rows_external = true;
    t->compact_len = 0;          // dense [nrows][c] rows, also after an earlier plonk_fft1_dev_compact on this task
    t->rows_filled = t->nrows;
    return PLONK_OK;
}

// The rows of a ZERO-PADDED vector: only the leading `row_len` coefficients of every decimated row are given ([num_rows][row_len],
// row_len <= c); the rest of the row is zero by construction and is never materialised.  The reference pads n + 2 / n + 3
// coefficients to the 8n-point domain before decimating (dispatcher2.rs:746, 754): row b of length c then has c/8 (+ 1 for b < 3)
// leading non-zero entries.  Forward transforms only.  The buffer is not modified.
extern "C" int plonk_fft1_dev_compact(plonk_ctx* ctx, uint64_t id, const void* d_rows, size_t row_len) {
    CHECK_CTX(ctx);
    FftTask* t;
    int rc = get_task(ctx, id, &t);
    if (rc) return rc;
    if (!d_rows) return plonk_fail(PLONK_ERR_ARG, "plonk_fft1_dev_compact: null");
    if (t->prepared) return plonk_fail(PLONK_ERR_STATE, "plonk_fft1_dev_compact: already prepared");
    if (t->is_inv) return plonk_fail(PLONK_ERR_ARG, "plonk_fft1_dev_compact: zero-padded rows are a forward-transform input");
    if (row_len == 0 || row_len > t->c) return plonk_fail(PLONK_ERR_ARG, "plonk_fft1_dev_compact: row_len %zu outside [1, %llu]", row_len, (unsigned long long)t->c);
    if (t->d_rows && !t->rows_external) pool_put(ctx, t->nrows * t->c * 32, t->d_rows);
    t->d_rows = (Fr*)const_cast<void*>(d_rows);
    t->rows_external = true;
    t->compact_len = row_len;
    t->rows_filled = t->nrows;
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- RCCL transport
extern "C" int plonk_comm_unique_id(void* out_id) {
    if (!out_id) return plonk_fail(PLONK_ERR_ARG, "plonk_comm_unique_id: null");
    return comm_unique_id(out_id);
}
// ---- the ranks must enter their collectives in the SAME ORDER (comm_rccl.hip: one total order per device).  A host that does not — two tasks
// finished in different orders on two ranks, a lane taken from a different queue — deadlocks inside RCCL, every GPU at 100 %, no message.
// PLONK_COMM_CHECK_ORDER=1 (every rank; or option "comm_check_order") makes that an error instead: before each collective the ranks
// all-gather a 24-byte tag — (ordinal of the communicator among the device's communicators in creation order, kind, bytes, the device's
// collective count) — over the device's FIRST communicator, which every rank creates first and which therefore always lines up, compare
// on the host, and every rank returns PLONK_ERR_STATE when the tags differ.  A host round trip per collective: a diagnostic for the first
// runs on a new machine (tests/test_gpu_multirank.py), not a production setting.
static int comm_order_check(plonk_ctx* ctx, uint64_t kind, uint64_t bytes) {
    if (!comm_check_on() || ctx->device < 0 || ctx->device >= 64) return PLONK_OK;
    plonk_ctx* control;
    uint64_t tag[3];
    {
        std::lock_guard<std::mutex> g(g_coll_mu);
        DeviceCollectives& d = g_coll[ctx->device];
        control = d.control;
        if (!control || !control->comm) return plonk_fail(PLONK_ERR_STATE, "collective order check: no live communicator on device %d to carry the tags", ctx->device);
        // communicators of different sizes: nothing to line up against — and nothing counted: ranks outside such a sub-communicator never see
        // this collective, so counting it would leave their sequence numbers behind for every later checked collective
        if (comm_world(ctx->comm) != comm_world(control->comm)) return PLONK_OK;
        tag[0] = ((uint64_t)(uint32_t)ctx->comm_ordinal << 8) | kind;
        tag[1] = bytes;
        tag[2] = d.seq++;
    }
    const int world = comm_world(control->comm);
    std::vector<uint64_t> all((size_t)3 * world);
    int rc = comm_allgather_host(control->comm, tag, sizeof tag, all.data(), control->stream);
    if (rc) return rc;
    for (int r = 0; r < world; r++)
        if (memcmp(&all[(size_t)3 * r], tag, sizeof tag) != 0)
            return plonk_fail(PLONK_ERR_STATE, "collective #%llu of device %d: this rank enters an %s of its communicator %llu (%llu bytes), rank %d an %s of its "
                              "communicator %llu (%llu bytes, its collective #%llu): the ranks issue their collectives in different orders — refused instead of "
                              "deadlocking inside RCCL", (unsigned long long)tag[2], ctx->device, coll_name(tag[0] & 0xff), (unsigned long long)(tag[0] >> 8),
                              (unsigned long long)tag[1], r, coll_name(all[(size_t)3 * r] & 0xff), (unsigned long long)(all[(size_t)3 * r] >> 8),
                              (unsigned long long)all[(size_t)3 * r + 1], (unsigned long long)all[(size_t)3 * r + 2]);
    return PLONK_OK;
}

extern "C" int plonk_comm_init(plonk_ctx* ctx, const void* id, int rank, int world) {
    CHECK_CTX(ctx);
    if (!id) return plonk_fail(PLONK_ERR_ARG, "plonk_comm_init: null id");
    if (ctx->comm) return plonk_fail(PLONK_ERR_STATE, "plonk_comm_init: the context already has a communicator");
    int rc = comm_create(&ctx->comm, id, rank, world, ctx->device);
    if (!rc && ctx->device >= 0 && ctx->device < 64) {
        std::lock_guard<std::mutex> g(g_coll_mu);
        DeviceCollectives& d = g_coll[ctx->device];
        ctx->comm_ordinal = d.created++;
        d.live.push_back(ctx);
        if (!d.control) d.control = ctx;
    }
    return rc;
}
// The control communicator is the OLDEST LIVE one of the device: when its context goes, the next in creation order takes over (every rank
// creates and destroys its communicators in the same order, so the successor lines up as the first one did) instead of every later checked
// collective failing with "the first communicator is gone" while other communicators are alive.
static void comm_forget(plonk_ctx* ctx) {
    if (ctx->device < 0 || ctx->device >= 64) return;
    std::lock_guard<std::mutex> g(g_coll_mu);
    DeviceCollectives& d = g_coll[ctx->device];
    d.live.erase(std::remove(d.live.begin(), d.live.end(), ctx), d.live.end());
    if (d.control == ctx) d.control = d.live.empty() ? nullptr : d.live.front();
}
extern "C" int plonk_comm_destroy(plonk_ctx* ctx) {
    CHECK_CTX(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    comm_forget(ctx);
    comm_destroy(ctx->comm);
    ctx->comm = nullptr;
    return PLONK_OK;
}
extern "C" int plonk_comm_info(plonk_ctx* ctx, int* rank, int* world, int* rccl_version) {
    CHECK_CTX(ctx);
    if (!ctx->comm) return plonk_fail(PLONK_ERR_STATE, "plonk_comm_info: no communicator (plonk_comm_init)");
    if (rank) *rank = comm_rank(ctx->comm);
    if (world) *world = comm_world(ctx->comm);
    if (rccl_version) *rccl_version = comm_rccl_version();
    return PLONK_OK;
}
// A stand-in for the exchange of an n_ranks job when only ONE GPU is there to run one rank's share (bench.py --simulate-ranks): the blocks
// that would leave for / arrive from the n_ranks - 1 peers are moved send -> recv on the same stream, device to device — the same bytes
// through the same buffers at HBM speed, so the two-lane overlap of the distributed transform meets a non-zero exchange.  Not a transport.
extern "C" int plonk_exchange_standin(void* user, const void* send, void* recv, size_t bytes_per_peer, int n_ranks, void* stream) {
    (void)user;
    if (!send || !recv || n_ranks < 1) return plonk_fail(PLONK_ERR_ARG, "plonk_exchange_standin: null / no ranks");
    if (n_ranks > 1)
        HIP_TRY(hipMemcpyAsync((char*)recv + bytes_per_peer, (const char*)send + bytes_per_peer, bytes_per_peer * (size_t)(n_ranks - 1), hipMemcpyDeviceToDevice,
                               (hipStream_t)stream));
    return PLONK_OK;
}
extern "C" int plonk_exchange_rccl(void* user, const void* send, void* recv, size_t bytes_per_peer, int n_ranks, void* stream) {
    plonk_ctx* ctx = (plonk_ctx*)user;
    if (!ctx || !ctx->comm) return plonk_fail(PLONK_ERR_STATE, "plonk_exchange_rccl: no communicator (plonk_comm_init)");
    if (n_ranks != comm_world(ctx->comm)) return plonk_fail(PLONK_ERR_ARG, "plonk_exchange_rccl: %d blocks for %d ranks", n_ranks, comm_world(ctx->comm));
    if (int rc = comm_order_check(ctx, 1, bytes_per_peer)) return rc;
    return comm_alltoall(ctx->comm, send, recv, bytes_per_peer, (hipStream_t)stream);
}
extern "C" int plonk_comm_alltoall_dev(plonk_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_peer) {
    CHECK_CTX(ctx);
    if (!ctx->comm) return plonk_fail(PLONK_ERR_STATE, "plonk_comm_alltoall_dev: no communicator (plonk_comm_init)");
    if (!d_send || !d_recv) return plonk_fail(PLONK_ERR_ARG, "plonk_comm_alltoall_dev: null");
    if (int rc = comm_order_check(ctx, 1, bytes_per_peer)) return rc;
    return comm_alltoall(ctx->comm, d_send, d_recv, bytes_per_peer, ctx->stream);
}
extern "C" int plonk_comm_allgather_dev(plonk_ctx* ctx, const void* d_send, void* d_recv, size_t bytes) {
    CHECK_CTX(ctx);
    if (!ctx->comm) return plonk_fail(PLONK_ERR_STATE, "plonk_comm_allgather_dev: no communicator (plonk_comm_init)");
    if (!d_send || !d_recv) return plonk_fail(PLONK_ERR_ARG, "plonk_comm_allgather_dev: null");
    if (int rc = comm_order_check(ctx, 2, bytes)) return rc;
    return comm_allgather(ctx->comm, d_send, d_recv, bytes, ctx->stream);
}
extern "C" int plonk_comm_allgather_host(plonk_ctx* ctx, const void* in, size_t bytes, void* out) {
    CHECK_CTX(ctx);
    if (!ctx->comm) return plonk_fail(PLONK_ERR_STATE, "plonk_comm_allgather_host: no communicator (plonk_comm_init)");
    if (!in || !out || !bytes) return plonk_fail(PLONK_ERR_ARG, "plonk_comm_allgather_host: null / empty");
    if (int rc = comm_order_check(ctx, 3, bytes)) return rc;
    return comm_allgather_host(ctx->comm, in, bytes, out, ctx->stream);
}

// ---------------------------------------------------------------------------------------------- fft2Prepare @4 (+ fftExchange)
extern "C" int plonk_fft2_prepare(plonk_ctx* ctx, uint64_t id, plonk_exchange_fn exchange, void* user) {
    CHECK_CTX(ctx);
    FftTask* t;
    int rc = get_task(ctx, id, &t);
    if (rc) return rc;
    if (t->prepared) return plonk_fail(PLONK_ERR_STATE, "plonk_fft2_prepare: called twice");
    if (t->rows_filled != t->nrows) return plonk_fail(PLONK_ERR_STATE, "plonk_fft2_prepare: %llu of %llu rows received", (unsigned long long)t->rows_filled,
                                                      (unsigned long long)t->nrows);
    const size_t S = t->wl.size();
    // transport precedence: an explicit callback; else the context's RCCL communicator when the task spans exactly its ranks; a
    // single-workload task on a context that joined a larger communicator stays local (nothing to exchange)
    if (!exchange && ctx->comm && (size_t)comm_world(ctx->comm) == S) {
        exchange = plonk_exchange_rccl;
        user = ctx;
    }
    if (!exchange && ctx->comm && S > 1)
        return plonk_fail(PLONK_ERR_ARG, "plonk_fft2_prepare: %zu workloads but the communicator has %d ranks", S, comm_world(ctx->comm));
    if (S > 1 && !exchange) return plonk_fail(PLONK_ERR_ARG, "plonk_fft2_prepare: %zu ranks need plonk_comm_init or an exchange callback", S);
    const size_t tile_bytes = t->nrows * t->c * 32;     // == r * ncols * 32
    if (!t->d_send && (rc = pool_get(ctx, tile_bytes, (void**)&t->d_send))) return rc;      // kept across a failed attempt, not re-taken
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    // row pass (fft1_helper, worker.rs:66-94) for all local rows, written straight into the
    // per-peer blocks of the send buffer (the pack of worker.rs:327-330)
    NttCall c;
    c.in = t->d_rows;
    c.out = t->d_send;
    c.log_m = t->log_n - (t->log_n >> 1);
    c.batch = t->nrows;
    c.inverse = t->is_inv;
    c.q_offset = t->wl[t->me].row_start;
    if (t->is_coset && !t->is_inv) { c.pro.kind = 1; c.pro.aq = 1; c.pro.b0 = t->r; }   // g^(i + j*r)
    c.epi.kind = t->is_inv ? 4 : 3;                                                      // w_N^(+-i*j)
    c.epi.bq = 1;
    c.epi.log_order = t->log_n;
    c.split_log = ilog2_exact(t->ncols);
    c.split_blk = t->nrows * t->ncols;
    if (t->compact_len) {
        // zero-padded rows: row b's size-c transform is 2^k independent (c / 2^k)-point transforms of its leading coefficients on
        // the sub-cosets (g^r * w_c^q) * <w_(c/2^k)> (the row's own constant g^b and the twiddle w_N^(b*k') ride in the output plane)
        const uint64_t len = t->compact_len;
        int k = 0;
        while (k < 4 && c.log_m - k > 1 && 2 * len < 3 * (t->c >> (k + 1))) k++;       // same rule as coset_eval_run
        if (!t->d_work && (rc = pool_get(ctx, tile_bytes, (void**)&t->d_work))) return rc;
        c.pro = ScaleSpec();
        c.shared_in = true;
        c.in_rows = t->nrows;
        c.in_len = len;
        c.in_pitch = len;
        c.work = t->d_work;
        c.log_m -= k;
        c.batch = t->nrows << k;
        const FrParams& FP = fr_params(ctx->curve);
        Fr sh = fp_one(FP);
        if (t->is_coset) {
            sh = fp_from_limbs<8>(ctx->curve == PLONK_BN254 ? BN254_FR_GENERATOR_MONT : BLS12_381_FR_GENERATOR_MONT);
            for (int i = 0; i < (t->log_n >> 1); i++) sh = fp_sqr(sh, FP);             // g^r
        }
        c.shift = sh;
        c.row_coset_const = t->is_coset;
    }
    if ((rc = ntt_run(ctx->tables, c, ctx->stream))) return rc;
    if (exchange) {          // also honoured for a single rank (all-to-all with oneself), so the transport is testable on one GPU
        if ((!t->d_recv || t->d_recv == t->d_send) && (rc = pool_get(ctx, tile_bytes, (void**)&t->d_recv))) return rc;
        const int xr = exchange(user, t->d_send, t->d_recv, t->nrows * t->ncols * 32, (int)S, (void*)ctx->stream);
        if (xr) return plonk_fail(PLONK_ERR_EXCHANGE, "exchange callback returned %d", xr);
    } else {
        t->d_recv = t->d_send;
    }
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    if (t->d_rows && !t->rows_external) pool_put(ctx, tile_bytes, t->d_rows);     // task.rows = vec![] (worker.rs:341)
    t->d_rows = nullptr;
    t->prepared = true;
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- fft2 @5
static int fft2_common(plonk_ctx* ctx, uint64_t id, Fr* d_out, int layout) {
    FftTask* t;
    int rc = get_task(ctx, id, &t);
    if (rc) return rc;
    if (!t->prepared) return plonk_fail(PLONK_ERR_STATE, "plonk_fft2: fft2_prepare has not run for this task");
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    // column pass (fft2_helper, worker.rs:96-115): the received blocks form an [r][ncols] matrix,
    // so the scatter-transpose of worker.rs:432-435 is just a stride in the loads.
    NttCall c;
    c.in = t->d_recv;
    c.out = d_out;
    c.log_m = t->log_n >> 1;
    c.batch = t->ncols;
    c.layout = NTT_INTERLEAVED;
    c.out_layout = layout == 0 ? NTT_CONTIGUOUS : NTT_INTERLEAVED;
    c.inverse = t->is_inv;
    c.q_offset = t->wl[t->me].col_start;
    if (t->is_coset && t->is_inv) { c.epi.kind = 2; c.epi.aq = 1; c.epi.b0 = t->c; }     // g^-(i + j*c)
    rc = ntt_run(ctx->tables, c, ctx->stream);
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    return rc;
}

extern "C" int plonk_fft2_dev(plonk_ctx* ctx, uint64_t id, void* d_out, int layout) {
    CHECK_CTX(ctx);
    if (!d_out || (layout != 0 && layout != 1)) return plonk_fail(PLONK_ERR_ARG, "plonk_fft2_dev: bad argument");
    int rc = fft2_common(ctx, id, (Fr*)d_out, layout);
    if (rc) return rc;
    // no host synchronisation: the task's buffers go back to the context's pool and are only ever reused by
    // work ordered on the same stream, so several transforms (on several contexts) can be in flight
    auto it = ctx->tasks.find(id);
    free_task(ctx, it->second);
    ctx->tasks.erase(it);                       // worker.rs:378
    return PLONK_OK;
}

extern "C" int plonk_fft2(plonk_ctx* ctx, uint64_t id, uint64_t* out_cols) {
    CHECK_CTX(ctx);
    if (!out_cols) return plonk_fail(PLONK_ERR_ARG, "plonk_fft2: null");
    FftTask* t;
    int rc = get_task(ctx, id, &t);
    if (rc) return rc;
    const size_t bytes = t->ncols * t->r * 32;
    if ((rc = ensure_scratch2(ctx, bytes))) return rc;
    if ((rc = fft2_common(ctx, id, (Fr*)ctx->d_scratch2, 0))) return rc;
    HIP_TRY(hipMemcpyAsync(out_cols, ctx->d_scratch2, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    auto it = ctx->tasks.find(id);
    free_task(ctx, it->second);
    ctx->tasks.erase(it);
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- round1 @6
extern "C" int plonk_round1(plonk_ctx* ctx, const uint64_t* evals, size_t n, const uint64_t* blinders, uint64_t* out_commit) {
    CHECK_CTX(ctx);
    if (!evals || !blinders || !out_commit) return plonk_fail(PLONK_ERR_ARG, "plonk_round1: null");
    if (n != ctx->domain_size || n < 2) return plonk_fail(PLONK_ERR_ARG, "plonk_round1: %zu evaluations, domain is %zu", n, ctx->domain_size);
    int rc = ensure_scratch2(ctx, (n + 2) * 32);
    if (rc) return rc;
    if (ctx->wire_len < n + 2) {
        if (ctx->d_wire) (void)hipFree(ctx->d_wire);
        ctx->d_wire = nullptr; ctx->wire_len = 0;
        HIP_TRY(hipMalloc((void**)&ctx->d_wire, (n + 2) * 32));
        ctx->wire_len = n + 2;
    }
    Fr* tmp = (Fr*)ctx->d_scratch2;
    HIP_TRY(hipMemcpyAsync(tmp, evals, n * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_wire + n, 0, 64, ctx->stream));
    {
        NttCall c;
        c.in = tmp; c.out = ctx->d_wire; c.log_m = ilog2_exact(n); c.batch = 1; c.inverse = true;     // worker.rs:398
        rc = ntt_run(ctx->tables, c, ctx->stream);
    }
    if (!rc) rc = blind_run(ctx->tables, ctx->d_wire, n, blinders, 2, ctx->stream);                    // worker.rs:400-401
    if (!rc) rc = plonk_commit_dev(ctx, ctx->d_wire, n + 2, out_commit);                               // worker.rs:403-405
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

extern "C" int plonk_get_wire(plonk_ctx* ctx, uint64_t* out, size_t n_coeffs) {
    CHECK_CTX(ctx);
    if (!out || n_coeffs > ctx->wire_len) return plonk_fail(PLONK_ERR_ARG, "plonk_get_wire: %zu > resident %zu", n_coeffs, ctx->wire_len);
    HIP_TRY(hipMemcpyAsync(out, ctx->d_wire, n_coeffs * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

// ---------------------------------------------------------------------------------------------- quotient evaluations (§8f rank 1)
static int quotient_common(plonk_ctx* ctx, const plonk_quotient_inputs* in, const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma,
                           const uint64_t* k, uint32_t cls_stride, uint32_t cls_offset, void* d_out, const char* who) {
    CHECK_CTX(ctx);
    if (!in || !alpha || !beta || !gamma || !k || !d_out) return plonk_fail(PLONK_ERR_ARG, "%s: null", who);
    for (int j = 0; j < 13; j++) if (!in->selectors[j]) return plonk_fail(PLONK_ERR_ARG, "%s: null selector %d", who, j);
    for (int j = 0; j < 5; j++) if (!in->sigmas[j] || !in->wires[j]) return plonk_fail(PLONK_ERR_ARG, "%s: null sigma/wire %d", who, j);
    if (!in->perm || !in->pub_input) return plonk_fail(PLONK_ERR_ARG, "%s: null perm/pub_input", who);
    if (ctx->domain_size < 2 || ctx->quot_domain_size < ctx->domain_size)
        return plonk_fail(PLONK_ERR_STATE, "%s: domains not initialised (n = %zu, m = %zu)", who, ctx->domain_size, ctx->quot_domain_size);
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    int rc = quotient_evals_run(ctx->tables, in, ctx->domain_size, ctx->quot_domain_size, alpha, beta, gamma, k, cls_stride, cls_offset, d_out, ctx->stream);
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    return rc;
}
extern "C" int plonk_quotient_evals_dev(plonk_ctx* ctx, const plonk_quotient_inputs* in, const uint64_t* alpha, const uint64_t* beta,
                                        const uint64_t* gamma, const uint64_t* k, void* d_out) {
    return quotient_common(ctx, in, alpha, beta, gamma, k, 1, 0, d_out, "plonk_quotient_evals_dev");
}
extern "C" int plonk_quotient_evals_class_dev(plonk_ctx* ctx, const plonk_quotient_inputs* in, const uint64_t* alpha, const uint64_t* beta,
                                              const uint64_t* gamma, const uint64_t* k, uint32_t class_stride, uint32_t class_offset, void* d_out) {
    return quotient_common(ctx, in, alpha, beta, gamma, k, class_stride, class_offset, d_out, "plonk_quotient_evals_class_dev");
}

// ---------------------------------------------------------------------------------------------- permutation product (§8f rank 2)
extern "C" int plonk_perm_product_dev(plonk_ctx* ctx, const void* const d_wires[5], const void* d_id_perm, const void* d_perm_idx,
                                      const uint64_t* beta, const uint64_t* gamma, size_t n, void* d_out) {
    CHECK_CTX(ctx);
    if (!d_wires || !d_id_perm || !d_perm_idx || !beta || !gamma || !d_out) return plonk_fail(PLONK_ERR_ARG, "plonk_perm_product_dev: null");
    for (int j = 0; j < 5; j++) if (!d_wires[j]) return plonk_fail(PLONK_ERR_ARG, "plonk_perm_product_dev: null wire %d", j);
    if (n < 2 || n >= ((size_t)1 << 32)) return plonk_fail(PLONK_ERR_ARG, "plonk_perm_product_dev: n = %zu", n);
    int rc = ensure_scratch2(ctx, perm_product_scratch_bytes(n));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    rc = perm_product_run(ctx->tables, d_wires, d_id_perm, d_perm_idx, beta, gamma, n, 0, n, d_out, ctx->d_scratch2, ctx->stream);
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    return rc;
}
extern "C" int plonk_perm_product_range_dev(plonk_ctx* ctx, const void* const d_wires[5], const void* d_id_perm, const void* d_perm_idx,
                                            const uint64_t* beta, const uint64_t* gamma, size_t n, size_t first, size_t count, void* d_out) {
    CHECK_CTX(ctx);
    if (!d_wires || !d_id_perm || !d_perm_idx || !beta || !gamma || !d_out) return plonk_fail(PLONK_ERR_ARG, "plonk_perm_product_range_dev: null");
    for (int j = 0; j < 5; j++) if (!d_wires[j]) return plonk_fail(PLONK_ERR_ARG, "plonk_perm_product_range_dev: null wire %d", j);
    if (n < 2 || n >= ((size_t)1 << 32)) return plonk_fail(PLONK_ERR_ARG, "plonk_perm_product_range_dev: n = %zu", n);
    if (count == 0 || first >= n || count > n - first)
        return plonk_fail(PLONK_ERR_ARG, "plonk_perm_product_range_dev: gates [%zu, %zu + %zu) of %zu", first, first, count, n);
    int rc = ensure_scratch2(ctx, perm_product_scratch_bytes(count));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));          // plonk_last_kernel_ms, like the whole-vector entry point
    rc = perm_product_run(ctx->tables, d_wires, d_id_perm, d_perm_idx, beta, gamma, n, first, count, d_out, ctx->d_scratch2, ctx->stream);
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->ev_valid = true;
    return rc;
}
extern "C" int plonk_class_interleave_dev(plonk_ctx* ctx, const void* d_in, size_t classes, size_t size, size_t in_stride, int reverse, const uint64_t* scale,
                                          void* d_out) {
    CHECK_CTX(ctx);
    if (!d_in || !d_out) return plonk_fail(PLONK_ERR_ARG, "plonk_class_interleave_dev: null");
    return class_interleave_run(ctx->tables, d_in, classes, size, in_stride, reverse, scale, d_out, ctx->stream);
}

// ---------------------------------------------------------------------------------------------- round 4/5 polynomial ops (§8f rank 3)
extern "C" int plonk_poly_eval_dev(plonk_ctx* ctx, const void* d_poly, size_t len, const uint64_t* point, uint64_t* out) {
    CHECK_CTX(ctx);
    if ((!d_poly && len) || !point || !out) return plonk_fail(PLONK_ERR_ARG, "plonk_poly_eval_dev: null");
    int rc = ensure_scratch2(ctx, poly_scratch_bytes(len));
    if (rc) return rc;
    return poly_eval_run(ctx->tables, d_poly, len, point, out, ctx->d_scratch2, ctx->stream);
}
extern "C" int plonk_poly_lincomb_dev(plonk_ctx* ctx, size_t k, const void* const* d_polys, const size_t* lens, const uint64_t* coeffs,
                                      void* d_out, size_t out_len) {
    CHECK_CTX(ctx);
    if (!d_polys || !lens || !coeffs || (!d_out && out_len)) return plonk_fail(PLONK_ERR_ARG, "plonk_poly_lincomb_dev: null");
    return poly_lincomb_run(ctx->tables, k, d_polys, lens, coeffs, d_out, out_len, ctx->stream);
}
extern "C" int plonk_poly_div_linear_dev(plonk_ctx* ctx, const void* d_poly, size_t len, const uint64_t* point, void* d_out) {
    CHECK_CTX(ctx);
    if ((!d_poly && len) || !point || (!d_out && len > 1)) return plonk_fail(PLONK_ERR_ARG, "plonk_poly_div_linear_dev: null");
    int rc = ensure_scratch2(ctx, poly_scratch_bytes(len));
    if (rc) return rc;
    return poly_div_linear_run(ctx->tables, d_poly, len, point, d_out, ctx->d_scratch2, ctx->stream);
}
extern "C" int plonk_poly_degree_dev(plonk_ctx* ctx, const void* d_poly, size_t len, int64_t* degree) {
    CHECK_CTX(ctx);
    if ((!d_poly && len) || !degree) return plonk_fail(PLONK_ERR_ARG, "plonk_poly_degree_dev: null");
    int rc = ensure_scratch2(ctx, 256);
    if (rc) return rc;
    return poly_degree_run(d_poly, len, degree, ctx->d_scratch2, ctx->stream);
}
extern "C" int plonk_coset_eval_dev(plonk_ctx* ctx, const void* d_poly, size_t len, size_t size, const uint64_t* shift, void* d_out) {
    CHECK_CTX(ctx);
    if ((!d_poly && len) || !shift || !d_out) return plonk_fail(PLONK_ERR_ARG, "plonk_coset_eval_dev: null");
    if (d_poly == d_out) return plonk_fail(PLONK_ERR_ARG, "plonk_coset_eval_dev: in-place not supported");
    int rc = ensure_scratch2(ctx, coset_scratch_bytes(size));
    if (rc) return rc;
    return coset_eval_run(ctx->tables, d_poly, len, size, shift, d_out, ctx->d_scratch2, ctx->stream);
}
extern "C" int plonk_coset_interp_dev(plonk_ctx* ctx, void* d_evals, size_t size, const uint64_t* shift, const uint64_t* scale, size_t i0, size_t count,
                                      void* d_out) {
    CHECK_CTX(ctx);
    if (!d_evals || !shift || !scale || (!d_out && count)) return plonk_fail(PLONK_ERR_ARG, "plonk_coset_interp_dev: null");
    int rc = ensure_scratch2(ctx, coset_scratch_bytes(size));
    if (rc) return rc;
    return coset_interp_run(ctx->tables, d_evals, size, shift, scale, i0, count, d_out, ctx->d_scratch2, ctx->stream);
}
extern "C" int plonk_blind_dev(plonk_ctx* ctx, void* d_poly, size_t n, const uint64_t* blinders, size_t k) {
    CHECK_CTX(ctx);
    if (!d_poly || (!blinders && k)) return plonk_fail(PLONK_ERR_ARG, "plonk_blind_dev: null");
    return blind_run(ctx->tables, d_poly, n, blinders, k, ctx->stream);
}

// ---------------------------------------------------------------------------------------------- memory / synth / debug
extern "C" int plonk_dev_alloc(plonk_ctx* ctx, size_t bytes, void** out) {
    CHECK_CTX(ctx);
    if (!out) return plonk_fail(PLONK_ERR_ARG, "plonk_dev_alloc: null");
    const hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {              // an out-of-memory report that explains itself: what was asked for, what the device had left
        (void)hipGetLastError();
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        return plonk_fail(PLONK_ERR_HIP, "plonk_dev_alloc: hipMalloc of %.2f GiB -> %s (device %d: %.1f GiB free of %.1f GiB)", (double)bytes / (double)(1ull << 30),
                          hipGetErrorString(e), ctx->device, (double)free_b / (double)(1ull << 30), (double)total_b / (double)(1ull << 30));
    }
    return PLONK_OK;
}
// A worker lives across circuits (worker.rs:42-59: one State per process): what a context caches for the LAST problem size — the exchange buffers of
// finished FFT tasks (up to six, 4 GiB each after a 2^27-point transform), NTT factor planes (up to tens of GiB), the MSM workspace, scratch — stays
// allocated until the context dies.  plonk_trim gives all of it back; everything is rebuilt on demand by the next call that needs it.  The SRS, the
// domains, open FFT tasks and the communicator are untouched.
extern "C" int plonk_trim(plonk_ctx* ctx) {
    CHECK_CTX(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto& pb : ctx->pool) (void)hipFree(pb.second);
    ctx->pool.clear();
    ntt_tables_trim(ctx->tables);
    msm_ws_release(ctx->msm_ws);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_scratch2) (void)hipFree(ctx->d_scratch2);
    ctx->d_scratch = ctx->d_scratch2 = nullptr;
    ctx->scratch_bytes = ctx->scratch2_bytes = 0;
    return PLONK_OK;
}
extern "C" int plonk_dev_free(plonk_ctx* ctx, void* p) {
    CHECK_CTX(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(p));
    return PLONK_OK;
}
extern "C" int plonk_memcpy_h2d(plonk_ctx* ctx, void* d, const void* h, size_t bytes) {
    CHECK_CTX(ctx);
    HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}
extern "C" int plonk_memcpy_d2h(plonk_ctx* ctx, void* h, const void* d, size_t bytes) {
    CHECK_CTX(ctx);
    HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}
extern "C" int plonk_memset_dev(plonk_ctx* ctx, void* d, int byte, size_t bytes) {
    CHECK_CTX(ctx);
    if (!d && bytes) return plonk_fail(PLONK_ERR_ARG, "plonk_memset_dev: null");
    if (bytes) HIP_TRY(hipMemsetAsync(d, byte, bytes, ctx->stream));
    return PLONK_OK;
}
extern "C" int plonk_memcpy_d2d(plonk_ctx* ctx, void* dst, const void* src, size_t bytes) {
    CHECK_CTX(ctx);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}
extern "C" int plonk_memcpy_d2d_async(plonk_ctx* ctx, void* dst, const void* src, size_t bytes) {
    CHECK_CTX(ctx);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PLONK_OK;
}
extern "C" int plonk_profile_enable(plonk_ctx* ctx, int on) {
    CHECK_CTX(ctx);
    kernel_profiler().enabled = on != 0;
    return PLONK_OK;
}
extern "C" int plonk_profile_reset(plonk_ctx* ctx) {
    CHECK_CTX(ctx);
    kernel_profiler().reset();
    return PLONK_OK;
}
extern "C" int plonk_profile_get(plonk_ctx* ctx, const char* name, double* total_ms, uint64_t* launches) {
    CHECK_CTX(ctx);
    if (!name || !total_ms || !launches) return plonk_fail(PLONK_ERR_ARG, "plonk_profile_get: null");
    kernel_profiler().resolve();
    auto it = kernel_profiler().totals.find(name);
    *total_ms = it == kernel_profiler().totals.end() ? 0.0 : it->second.first;
    *launches = it == kernel_profiler().totals.end() ? 0 : it->second.second;
    return PLONK_OK;
}
extern "C" int plonk_synth_fr(plonk_ctx* ctx, uint64_t seed, void* d_out, size_t n) {
    CHECK_CTX(ctx);
    if (!d_out && n) return plonk_fail(PLONK_ERR_ARG, "plonk_synth_fr: null");
    return synth_fr_dev(ctx->curve, seed, (Fr*)d_out, n, ctx->stream);
}
extern "C" int plonk_synth_bases(plonk_ctx* ctx, uint64_t seed, size_t unique, size_t n, void* d_out) {
    CHECK_CTX(ctx);
    if (!d_out && n) return plonk_fail(PLONK_ERR_ARG, "plonk_synth_bases: null");
    if (unique == 0) return synth_bases_distinct_dev(ctx->curve, seed, n, d_out, ctx->stream);
    return synth_bases_dev(ctx->curve, seed, unique, n, d_out, ctx->stream);
}
extern "C" int plonk_synth_srs(plonk_ctx* ctx, const uint64_t* tau, size_t n, void* d_out) {
    CHECK_CTX(ctx);
    if (!tau || (!d_out && n)) return plonk_fail(PLONK_ERR_ARG, "plonk_synth_srs: null");
    return synth_srs_dev(ctx->curve, tau, n, d_out, ctx->stream);
}
extern "C" int plonk_synth_circuit(plonk_ctx* ctx, uint64_t seed, size_t n, size_t num_inputs, const uint64_t* k, void* d_wires, void* d_selector_evals,
                                   void* d_sigma_evals, void* d_id_perm, void* d_perm_idx, void* d_pub_input) {
    CHECK_CTX(ctx);
    if (!k || !d_wires || !d_selector_evals || !d_sigma_evals || !d_id_perm || !d_perm_idx || !d_pub_input)
        return plonk_fail(PLONK_ERR_ARG, "plonk_synth_circuit: null");
    if (n < 2 || (n & (n - 1))) return plonk_fail(PLONK_ERR_DOMAIN, "plonk_synth_circuit: n = %zu is not a power of two >= 2", n);
    int log_n = 0;
    while (((size_t)1 << log_n) < n) log_n++;
    if (log_n > ctx->tables.two_adicity) return plonk_fail(PLONK_ERR_DOMAIN, "plonk_synth_circuit: 2^%d exceeds the two-adicity", log_n);
    if (num_inputs > n) return plonk_fail(PLONK_ERR_ARG, "plonk_synth_circuit: %zu public inputs for %zu gates", num_inputs, n);
    Fr w = ctx->tables.h_root[0];                       // primitive 2^two_adicity-th root, squared down to order n
    for (int i = log_n; i < ctx->tables.two_adicity; i++) w = fp_sqr(w, ctx->tables.fp);
    return synth_circuit_dev(ctx->curve, seed, n, num_inputs, k, w, d_wires, d_selector_evals, d_sigma_evals, d_id_perm, d_perm_idx, d_pub_input,
                             ctx->stream);
}
extern "C" int plonk_debug_field_op(plonk_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    CHECK_CTX(ctx);
    if (!a || !out) return plonk_fail(PLONK_ERR_ARG, "plonk_debug_field_op: null");
    const size_t eb = (field == 1 && ctx->curve == PLONK_BLS12_381) ? 48 : 32;
    int rc = ensure_scratch2(ctx, 3 * n * eb + 48);
    if (rc) return rc;
    char* base = (char*)ctx->d_scratch2;
    HIP_TRY(hipMemcpyAsync(base, a, n * eb, hipMemcpyHostToDevice, ctx->stream));
    if (b) HIP_TRY(hipMemcpyAsync(base + n * eb, b, n * eb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = field_op_dev(ctx->curve, field, op, base, b ? base + n * eb : nullptr, base + 2 * n * eb, n, ctx->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, base + 2 * n * eb, n * eb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}
