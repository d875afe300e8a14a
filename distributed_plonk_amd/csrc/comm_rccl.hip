// comm_rccl.hip — the worker<->worker transport INSIDE the library: RCCL over xGMI.
//
// What it replaces (reference /root/reference/src): the S^2 per-exchange TCP + Cap'n Proto connections of
// `fft2_prepare` / `PlonkPeer::fft_exchange` (worker.rs:280-345, 412-438: connect to every peer, ship
// rows[.][cs_p..ce_p], scatter on receipt) become ONE grouped ncclSend/ncclRecv all-to-all on the context's
// stream; the `result: Data` replies of varMsm that the dispatcher adds up (dispatcher.rs:236-238,
// dispatcher2.rs:887-890) become one all-gather of 96/144-byte points.
//
// librccl is bound at run time (dlopen of its SONAME, then /opt/rocm/lib): a single-GPU user never pages in the 570 MB library,
// and a process that already carries an RCCL (PyTorch's wheel bundles one under the same SONAME) shares that copy instead of
// loading a second one.  No torch, no callback: a Rust worker calls plonk_comm_init once and then plonk_fft2_prepare(ctx, id,
// NULL, NULL) — INTEGRATION.md.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <cstdlib>
#include <mutex>
#include <thread>

#include "plonk_internal.hpp"

namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
RcclApi g_api;
std::mutex g_api_mu;

template <typename F> bool bind(void* h, const char* name, F& f) {
    f = reinterpret_cast<F>(dlsym(h, name));
    return f != nullptr;
}

int rccl_api(const RcclApi** out) {
    std::lock_guard<std::mutex> g(g_api_mu);
    if (!g_api.handle) {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return plonk_fail(PLONK_ERR_EXCHANGE, "RCCL not found: %s", dlerror());
        RcclApi a;
        const bool ok = bind(h, "ncclGetUniqueId", a.GetUniqueId) && bind(h, "ncclCommInitRank", a.CommInitRank) &&
                        bind(h, "ncclCommDestroy", a.CommDestroy) && bind(h, "ncclCommCount", a.CommCount) &&
                        bind(h, "ncclCommUserRank", a.CommUserRank) && bind(h, "ncclGroupStart", a.GroupStart) &&
                        bind(h, "ncclGroupEnd", a.GroupEnd) && bind(h, "ncclSend", a.Send) && bind(h, "ncclRecv", a.Recv) &&
                        bind(h, "ncclAllGather", a.AllGather) && bind(h, "ncclGetErrorString", a.GetErrorString) &&
                        bind(h, "ncclGetVersion", a.GetVersion);
        if (!ok) { dlclose(h); return plonk_fail(PLONK_ERR_EXCHANGE, "RCCL: missing symbol (%s)", dlerror()); }
        a.handle = h;
        g_api = a;
    }
    *out = &g_api;
    return PLONK_OK;
}
}  // namespace

struct PlonkComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    void* d_stage = nullptr;          // small device staging buffer for host-side gathers
    size_t stage_bytes = 0;
};

#define NCCL_TRY(api, expr)                                                                                              \
    do {                                                                                                                 \
        ncclResult_t _r = (expr);                                                                                        \
        if (_r != ncclSuccess) return plonk_fail(PLONK_ERR_EXCHANGE, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, (api)->GetErrorString(_r)); \
    } while (0)

static void comm_device_count(int dev, int delta);      // live communicators per device (the collective order below)

int comm_unique_id(void* out128) {
    const RcclApi* api;
    int rc = rccl_api(&api);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == PLONK_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(api, api->GetUniqueId(&id));
    memcpy(out128, &id, sizeof id);
    return PLONK_OK;
}

int comm_create(PlonkComm** out, const void* id128, int rank, int world, int device) {
    const RcclApi* api;
    int rc = rccl_api(&api);
    if (rc) return rc;
    if (world < 1 || rank < 0 || rank >= world) return plonk_fail(PLONK_ERR_ARG, "plonk_comm_init: rank %d of %d", rank, world);
    if (device < 0 || device >= 64) return plonk_fail(PLONK_ERR_ARG, "plonk_comm_init: device %d (0..63: the per-device collective order is a fixed table)", device);
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    PlonkComm* c = new PlonkComm();
    c->rank = rank; c->world = world; c->device = device;
    ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return plonk_fail(PLONK_ERR_EXCHANGE, "ncclCommInitRank(rank %d of %d): %s", rank, world, api->GetErrorString(r)); }
    // what RCCL itself reports (bench.py prints this, not the launcher's environment)
    int cnt = 0, ur = -1;
    if (api->CommCount(c->comm, &cnt) != ncclSuccess || api->CommUserRank(c->comm, &ur) != ncclSuccess || cnt != world || ur != rank) {
        api->CommDestroy(c->comm);
        delete c;
        return plonk_fail(PLONK_ERR_EXCHANGE, "RCCL reports rank %d of %d, expected %d of %d", ur, cnt, rank, world);
    }
    comm_device_count(device, +1);
    *out = c;
    return PLONK_OK;
}

void comm_destroy(PlonkComm* c) {
    if (!c) return;
    comm_device_count(c->device, -1);
    (void)hipSetDevice(c->device);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->comm && g_api.handle) (void)g_api.CommDestroy(c->comm);
    delete c;
}

int comm_rank(const PlonkComm* c) { return c->rank; }
int comm_world(const PlonkComm* c) { return c->world; }
int comm_rccl_version() {
    const RcclApi* api;
    if (rccl_api(&api)) return 0;
    int v = 0;
    return api->GetVersion(&v) == ncclSuccess ? v : 0;
}

// ---- one total order of collectives per GPU.  A process may hold several communicators on one device (one per context: the two
// lanes of the distributed transform, the two commitment contexts), each issuing on its own stream.  RCCL / NCCL do not guarantee
// progress when collectives of DIFFERENT communicators are in flight on a device at the same time: the ranks' GPUs may schedule the
// two kernels in different orders and each waits for the peer's other kernel.  So every collective of this library first waits (on
// its stream) for the previous collective issued on the same device, whatever communicator or stream that one used, and the issue
// itself happens under a mutex.  Collectives therefore run one after the other in HOST ISSUE ORDER — which callers must keep
// identical on every rank (bench.py and the provers issue them from one thread) — while still overlapping the other streams' compute.
namespace {
struct CollectiveOrder {
    std::mutex mu;
    hipEvent_t last[64] = {};
    std::thread::id owner[64];          // the host thread that issues this device's collectives (set by the first one)
    bool owned[64] = {};
    int comms[64] = {};                 // live communicators per device: the owner is forgotten with the last one
};
CollectiveOrder g_order;

// RAII: wait for the device's previous collective, issue, record.  The ordering only prevents the cross-communicator deadlock if every
// rank issues its collectives in the SAME order, which two host threads racing for the mutex cannot promise: a device's collectives must
// come from one thread (the first to issue one), anything else is refused (PLONK_ERR_STATE) instead of deadlocking with the peers.
// PLONK_COMM_ANY_THREAD=1 lifts the check for callers that serialise their collectives some other way (an async runtime that migrates
// one logical task between OS threads).
struct CollectiveScope {
    std::unique_lock<std::mutex> lock;
    int device;
    hipStream_t stream;
    int rc = PLONK_OK;
    const char* what = "";
    CollectiveScope(int dev, hipStream_t s) : lock(g_order.mu), device(dev), stream(s) {
        if (dev < 0 || dev >= 64) { rc = PLONK_ERR_ARG; what = "device index outside 0..63"; device = -1; return; }
        static const bool any_thread = getenv("PLONK_COMM_ANY_THREAD") != nullptr;
        const std::thread::id me = std::this_thread::get_id();
        if (g_order.owned[dev] && g_order.owner[dev] != me && !any_thread) {
            rc = PLONK_ERR_STATE; what = "collectives of one device must be issued from one host thread (same order on every rank)"; device = -1; return;
        }
        hipEvent_t& e = g_order.last[dev];
        hipError_t he = e ? hipStreamWaitEvent(stream, e, 0) : hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (he != hipSuccess) { rc = PLONK_ERR_HIP; what = "collective ordering event"; device = -1; return; }
        if (!g_order.owned[dev]) { g_order.owner[dev] = me; g_order.owned[dev] = true; }      // only a collective that really gets issued takes the device
    }
    // record the ordering point behind the collective; a failure here would silently drop the ordering, so it is reported
    int finish() {
        if (device < 0) return rc;
        const hipError_t he = hipEventRecord(g_order.last[device], stream);
        device = -1;
        return he == hipSuccess ? PLONK_OK : plonk_fail(PLONK_ERR_HIP, "collective ordering event: %s", hipGetErrorString(he));
    }
    ~CollectiveScope() {
        if (device >= 0) (void)hipEventRecord(g_order.last[device], stream);      // error exits: best effort, the error is already on its way
    }
};
}  // namespace

static void comm_device_count(int dev, int delta) {
    if (dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> g(g_order.mu);
    g_order.comms[dev] += delta;
    if (g_order.comms[dev] <= 0) { g_order.comms[dev] = 0; g_order.owned[dev] = false; }      // the next communicator's first collective picks the thread again
}

// block p of `send` -> rank p ; block p of `recv` <- rank p.  One group = one fused RCCL launch; per pair bytes_per_peer bytes,
// each pair over its own xGMI link (the fabric is point-to-point: 7 links per GPU, no switch).
int comm_alltoall(PlonkComm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t stream) {
    const RcclApi* api;
    int rc = rccl_api(&api);
    if (rc) return rc;
    const char* s = (const char*)send;
    char* r = (char*)recv;
    CollectiveScope order(c->device, stream);
    if (order.rc) return plonk_fail(order.rc, "plonk all-to-all: %s", order.what);
    ProfScope prof("rccl_alltoall", stream);          // HIP events on the stream: the exchange as the GPU saw it (incl. waiting for peers)
    NCCL_TRY(api, api->GroupStart());
    // at most 1 GiB per ncclSend / ncclRecv call: a single-rank world exchanges the WHOLE 4 GiB buffer of a 2^27-point transform with
    // itself, and byte counts at or above 2^32 are not safe to hand to one call
    const size_t CHUNK = (size_t)1 << 30;
    for (int p = 0; p < c->world; p++) {
        for (size_t off = 0; off < bytes_per_peer; off += CHUNK) {
            const size_t len = bytes_per_peer - off < CHUNK ? bytes_per_peer - off : CHUNK;
            ncclResult_t a = api->Send(s + (size_t)p * bytes_per_peer + off, len, ncclInt8, p, c->comm, stream);
            ncclResult_t b = a == ncclSuccess ? api->Recv(r + (size_t)p * bytes_per_peer + off, len, ncclInt8, p, c->comm, stream) : a;
            if (b != ncclSuccess) { (void)api->GroupEnd(); return plonk_fail(PLONK_ERR_EXCHANGE, "ncclSend/Recv (peer %d): %s", p, api->GetErrorString(b)); }
        }
    }
    NCCL_TRY(api, api->GroupEnd());
    return order.finish();
}

int comm_allgather(PlonkComm* c, const void* send, void* recv, size_t bytes, hipStream_t stream) {
    const RcclApi* api;
    int rc = rccl_api(&api);
    if (rc) return rc;
    CollectiveScope order(c->device, stream);
    if (order.rc) return plonk_fail(order.rc, "plonk all-gather: %s", order.what);
    ProfScope prof("rccl_allgather", stream);
    NCCL_TRY(api, api->AllGather(send, recv, bytes, ncclInt8, c->comm, stream));
    return order.finish();
}

// host buffers through a device staging area: in (bytes) -> out (world * bytes); synchronises the stream
int comm_allgather_host(PlonkComm* c, const void* in, size_t bytes, void* out, hipStream_t stream) {
    const size_t need = bytes * ((size_t)c->world + 1);
    if (c->stage_bytes < need) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->d_stage) (void)hipFree(c->d_stage);
        c->d_stage = nullptr; c->stage_bytes = 0;
        HIP_TRY(hipMalloc(&c->d_stage, need));
        c->stage_bytes = need;
    }
    char* d_in = (char*)c->d_stage;
    char* d_out = d_in + bytes;
    HIP_TRY(hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, stream));
    int rc = comm_allgather(c, d_in, d_out, bytes, stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_out, bytes * (size_t)c->world, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return PLONK_OK;
}
