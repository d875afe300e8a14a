// Stand-in for csrc/comm_rccl.hip in the host emulation (TEST INFRASTRUCTURE): the transport interface of plonk_internal.hpp
// (comm_*) over POSIX shared memory instead of RCCL, so that the multi-rank programs — one PROCESS per rank, as on a GPU box —
// can run on CPU through the library's own plonk_comm_* entry points.  "Device memory" of the emulation is private to each
// process, so a collective stages through a shared window: every rank copies its outgoing blocks into its slice of the window,
// a barrier, every rank copies what is addressed to it out of the peers' slices, a barrier; payloads larger than the window go
// chunk by chunk.  Semantics = csrc/comm_rccl.hip's: all-to-all of `bytes_per_peer` per pair (block p of `send` ends up as block
// `rank` of rank p's `recv`), all-gather of `bytes` per rank in rank order, blocking.
#include <fcntl.h>
#include <sched.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>

#include "plonk_internal.hpp"

namespace {
constexpr int MAX_WORLD = 8;
constexpr size_t WINDOW = (size_t)4 << 20;            // bytes of staging per rank
struct Control {
    std::atomic<uint32_t> count, generation, attached;
};
constexpr size_t CTL_BYTES = 4096;
constexpr size_t SHM_BYTES = CTL_BYTES + MAX_WORLD * WINDOW;
std::atomic<uint32_t> g_serial{0};
}  // namespace

struct PlonkComm {
    int rank = 0, world = 1;
    unsigned char* base = nullptr;                    // the mapping (null for a world of one)
    Control* ctl() const { return reinterpret_cast<Control*>(base); }
    unsigned char* window(int r) const { return base + CTL_BYTES + (size_t)r * WINDOW; }
    void barrier() const {
        Control* c = ctl();
        const uint32_t g = c->generation.load(std::memory_order_acquire);
        if (c->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
            c->count.store(0, std::memory_order_relaxed);
            c->generation.fetch_add(1, std::memory_order_release);
        } else {
            while (c->generation.load(std::memory_order_acquire) == g) sched_yield();
        }
    }
};

int comm_unique_id(void* out128) {
    // rank 0 creates the segment; its name travels in the 128-byte id (as ncclUniqueId does for RCCL's bootstrap)
    char name[128];
    memset(name, 0, sizeof(name));
    snprintf(name, sizeof(name), "/plonk_hostemu_%d_%u", (int)getpid(), g_serial.fetch_add(1));
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return plonk_fail(PLONK_ERR_HIP, "host emulation: shm_open(%s) failed", name);
    if (ftruncate(fd, (off_t)SHM_BYTES) != 0) { close(fd); shm_unlink(name); return plonk_fail(PLONK_ERR_HIP, "host emulation: ftruncate failed"); }
    close(fd);                                          // fresh pages are zero: count = generation = attached = 0
    memcpy(out128, name, 128);
    return PLONK_OK;
}

int comm_create(PlonkComm** out, const void* id128, int rank, int world, int device) {
    (void)device;
    if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return plonk_fail(PLONK_ERR_ARG, "host emulation: rank %d of %d", rank, world);
    char name[129];
    memcpy(name, id128, 128);
    name[128] = 0;
    PlonkComm* c = new PlonkComm();
    c->rank = rank; c->world = world;
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) { delete c; return plonk_fail(PLONK_ERR_HIP, "host emulation: shm_open(%s) failed", name); }
    void* m = mmap(nullptr, SHM_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return plonk_fail(PLONK_ERR_HIP, "host emulation: mmap of the shared window failed"); }
    c->base = (unsigned char*)m;
    // the name can go once every rank holds the mapping
    if (c->ctl()->attached.fetch_add(1) + 1 == (uint32_t)world) shm_unlink(name);
    c->barrier();
    *out = c;
    return PLONK_OK;
}

void comm_destroy(PlonkComm* c) {
    if (!c) return;
    if (c->base) munmap(c->base, SHM_BYTES);
    delete c;
}
int comm_rank(const PlonkComm* c) { return c->rank; }
int comm_world(const PlonkComm* c) { return c->world; }
int comm_rccl_version() { return 0; }                 // there is no RCCL underneath

int comm_alltoall(PlonkComm* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t) {
    const unsigned char* s = (const unsigned char*)send;
    unsigned char* r = (unsigned char*)recv;
    const size_t slot = WINDOW / (size_t)c->world & ~(size_t)31;
    for (size_t off = 0; off < bytes_per_peer; off += slot) {
        const size_t len = bytes_per_peer - off < slot ? bytes_per_peer - off : slot;
        for (int p = 0; p < c->world; p++) memcpy(c->window(c->rank) + (size_t)p * slot, s + (size_t)p * bytes_per_peer + off, len);
        c->barrier();
        for (int q = 0; q < c->world; q++) memcpy(r + (size_t)q * bytes_per_peer + off, c->window(q) + (size_t)c->rank * slot, len);
        c->barrier();
    }
    return PLONK_OK;
}

static int allgather(PlonkComm* c, const void* in, void* out, size_t bytes) {
    const unsigned char* s = (const unsigned char*)in;
    unsigned char* r = (unsigned char*)out;
    for (size_t off = 0; off < bytes; off += WINDOW) {
        const size_t len = bytes - off < WINDOW ? bytes - off : WINDOW;
        memcpy(c->window(c->rank), s + off, len);
        c->barrier();
        for (int q = 0; q < c->world; q++) memcpy(r + (size_t)q * bytes + off, c->window(q), len);
        c->barrier();
    }
    return PLONK_OK;
}
int comm_allgather(PlonkComm* c, const void* send, void* recv, size_t bytes, hipStream_t) { return allgather(c, send, recv, bytes); }
int comm_allgather_host(PlonkComm* c, const void* in, size_t bytes, void* out, hipStream_t) { return allgather(c, in, out, bytes); }
