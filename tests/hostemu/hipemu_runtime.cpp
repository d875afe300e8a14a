// Runtime of the host emulation (see hip/hip_runtime.h in this directory): TEST INFRASTRUCTURE.
//
// A launch runs its workgroups on a small pool of OS threads (a shared counter hands out block indices); inside a workgroup every
// HIP thread is a fiber with its own stack, switched by a dozen instructions of x86-64 assembly.  __syncthreads() and the
// wave-level __shfl() park the fiber; after each sweep over the runnable fibers the scheduler releases a barrier once every LIVE
// thread of the block (or of the 64-lane wave) has arrived — threads that returned no longer count, as on the hardware.  A sweep
// without progress is a divergent barrier: the emulator says so and aborts instead of hanging.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#define HIPEMU_ASAN 1
#endif

thread_local hipemu::Idx3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

#if !defined(__x86_64__)
#error "the fiber switch below is x86-64 System V"
#endif
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

namespace {
enum { READY = 0, AT_BLOCK_BARRIER = 1, AT_WAVE_BARRIER = 2, DONE = 3 };
#if defined(HIPEMU_ASAN)
constexpr size_t STACK_BYTES = 1u << 20;
#else
constexpr size_t STACK_BYTES = 256u << 10;
#endif
constexpr size_t SMEM_BYTES = 256u << 10;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = DONE;
    uint32_t shfl_seq = 0;
};

struct Worker {
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    int cur = -1;
    const std::function<void()>* body = nullptr;
    unsigned char* smem = nullptr;
    std::vector<int> shfl_buf[2];
    uint32_t nthreads = 0;
#if defined(HIPEMU_ASAN)
    void* sched_fake = nullptr;
    const void* sched_bottom = nullptr;
    size_t sched_size = 0;
#endif
};
thread_local Worker* tl_worker = nullptr;

void to_fiber(Worker* w, Fiber& f) {
#if defined(HIPEMU_ASAN)
    __sanitizer_start_switch_fiber(&w->sched_fake, f.stack, STACK_BYTES);
#endif
    hipemu_switch(&w->sched_sp, f.sp);
#if defined(HIPEMU_ASAN)
    __sanitizer_finish_switch_fiber(w->sched_fake, nullptr, nullptr);
#endif
}
void to_scheduler(Worker* w, Fiber& f) {
#if defined(HIPEMU_ASAN)
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, w->sched_bottom, w->sched_size);
#endif
    hipemu_switch(&f.sp, w->sched_sp);
#if defined(HIPEMU_ASAN)
    __sanitizer_finish_switch_fiber(fake, &w->sched_bottom, &w->sched_size);
#endif
}

void fiber_entry() {
    Worker* w = tl_worker;
#if defined(HIPEMU_ASAN)
    __sanitizer_finish_switch_fiber(nullptr, &w->sched_bottom, &w->sched_size);
#endif
    for (;;) {
        (*w->body)();
        Fiber& f = w->fibers[w->cur];
        f.state = DONE;
        to_scheduler(w, f);
    }
}

void make_fiber(Fiber& f) {
    void* m = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) { perror("hipemu: mmap of a fiber stack"); abort(); }
    f.stack = (char*)m;
    uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                     // the return address fiber_entry would return to (it never does)
    *--sp = (void*)&fiber_entry;         // popped by hipemu_switch's ret: rsp = top - 8 at entry, as after a call
    for (int i = 0; i < 6; i++) *--sp = nullptr;
    f.sp = sp;
    f.state = DONE;
}

int sweep_order();

void run_block(Worker* w, dim3 block) {
    const uint32_t nt = block.x * block.y * block.z;
    while (w->fibers.size() < nt) {
        w->fibers.emplace_back();
        make_fiber(w->fibers.back());
    }
    w->nthreads = nt;
    for (int b = 0; b < 2; b++)
        if (w->shfl_buf[b].size() < nt) w->shfl_buf[b].resize(nt);
    for (uint32_t t = 0; t < nt; t++) { w->fibers[t].state = READY; w->fibers[t].shfl_seq = 0; }
    uint32_t live = nt;
    const uint32_t nwaves = (nt + 63) / 64;
    const int order = sweep_order();
    uint64_t lcg = 0x9E3779B97F4A7C15ull * (1 + blockIdx.x + 131ull * blockIdx.y);
    while (live) {
        bool progressed = false;
        // HIPEMU_ORDER: the order in which a sweep resumes the runnable threads of a workgroup — 0 ascending (default), 1 descending,
        // 2 a fresh pseudo-random rotation + direction per sweep.  Between two barriers the threads of a block run one after another, so a
        // result that depends on this order is a missing barrier (or code that leans on wave-lockstep execution).
        uint32_t start = 0;
        bool down = order == 1;
        if (order == 2) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            start = (uint32_t)(lcg >> 33) % nt;
            down = (lcg >> 32) & 1;
        }
        for (uint32_t i = 0; i < nt; i++) {
            const uint32_t t = down ? (start + nt - i - (order == 2 ? 0 : 1)) % nt : (start + i) % nt;
            Fiber& f = w->fibers[t];
            if (f.state != READY) continue;
            threadIdx.x = t % block.x;
            threadIdx.y = (t / block.x) % block.y;
            threadIdx.z = t / (block.x * block.y);
            w->cur = (int)t;
            to_fiber(w, f);
            progressed = true;
            if (f.state == DONE) live--;
        }
        uint32_t at_block = 0;
        for (uint32_t t = 0; t < nt; t++) at_block += w->fibers[t].state == AT_BLOCK_BARRIER;
        if (at_block && at_block == live) {
            for (uint32_t t = 0; t < nt; t++)
                if (w->fibers[t].state == AT_BLOCK_BARRIER) w->fibers[t].state = READY;
            progressed = true;
        }
        for (uint32_t wv = 0; wv < nwaves; wv++) {
            uint32_t lv = 0, at = 0;
            const uint32_t hi = std::min(nt, (wv + 1) * 64);
            for (uint32_t t = wv * 64; t < hi; t++) { lv += w->fibers[t].state != DONE; at += w->fibers[t].state == AT_WAVE_BARRIER; }
            if (at && at == lv) {
                for (uint32_t t = wv * 64; t < hi; t++)
                    if (w->fibers[t].state == AT_WAVE_BARRIER) w->fibers[t].state = READY;
                progressed = true;
            }
        }
        if (!progressed) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %u live threads, %u at __syncthreads — a divergent barrier\n", blockIdx.x, blockIdx.y,
                    blockIdx.z, live, at_block);
            abort();
        }
    }
}

// ---------------------------------------------------------------------------------------------- the pool
struct Job {
    dim3 grid, block;
    size_t shmem = 0;
    const std::function<void()>* body = nullptr;
    std::atomic<uint64_t> next{0};
    uint64_t nblocks = 0;
};
// The pool's threads are detached and wait on these for the life of the process: the objects are deliberately never destroyed
// (a condition variable's destructor would wait for its waiters at exit).
std::mutex& g_launch_mutex = *new std::mutex;            // launches are synchronous and one at a time (several host threads may drive contexts)
std::mutex& g_mutex = *new std::mutex;
std::condition_variable& g_cv_work = *new std::condition_variable;
std::condition_variable& g_cv_done = *new std::condition_variable;
Job* g_job = nullptr;
uint64_t g_generation = 0;
int g_busy = 0;
std::vector<std::thread>& g_pool = *new std::vector<std::thread>;

// HIPEMU_FILL=<0..255>: fresh "device" allocations and every workgroup's dynamic LDS start out filled with this byte instead of
// whatever malloc returns — a run whose result changes with the fill pattern reads memory it never wrote.
int sweep_order() {
    static int v = [] {
        const char* e = getenv("HIPEMU_ORDER");
        return e ? atoi(e) : 0;
    }();
    return v;
}
int fill_byte() {
    static int v = [] {
        const char* e = getenv("HIPEMU_FILL");
        return e ? (atoi(e) & 255) : -1;
    }();
    return v;
}

void work_on(Worker* w, Job* job) {
#if defined(HIPEMU_ASAN)
    // only the bytes the launch asked for are LDS; the rest of the window is poisoned so an overrun is reported
    __asan_unpoison_memory_region(w->smem, SMEM_BYTES);
    __asan_poison_memory_region(w->smem + job->shmem, SMEM_BYTES - job->shmem);
#endif
    gridDim = job->grid;
    blockDim = job->block;
    w->body = job->body;
    for (;;) {
        const uint64_t b = job->next.fetch_add(1, std::memory_order_relaxed);
        if (b >= job->nblocks) break;
        blockIdx.x = (uint32_t)(b % job->grid.x);
        blockIdx.y = (uint32_t)((b / job->grid.x) % job->grid.y);
        blockIdx.z = (uint32_t)(b / ((uint64_t)job->grid.x * job->grid.y));
        if (fill_byte() >= 0) memset(w->smem, fill_byte(), job->shmem);      // LDS holds garbage when a workgroup starts
        run_block(w, job->block);
    }
}

Worker* this_worker() {
    if (!tl_worker) {
        tl_worker = new Worker();
        if (posix_memalign((void**)&tl_worker->smem, 256, SMEM_BYTES)) abort();
        memset(tl_worker->smem, 0, SMEM_BYTES);
    }
    return tl_worker;
}

void pool_main() {
    Worker* w = this_worker();
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(g_mutex);
    for (;;) {
        g_cv_work.wait(lk, [&] { return g_generation != seen; });
        seen = g_generation;
        Job* job = g_job;
        lk.unlock();
        work_on(w, job);
        lk.lock();
        if (--g_busy == 0) g_cv_done.notify_all();
    }
}

int pool_size() {
    static int n = [] {
        const char* e = getenv("HIPEMU_THREADS");
        int v = e ? atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1, v);
    }();
    return n;
}
}  // namespace

namespace hipemu {
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& thread_body) {
    if (shmem > SMEM_BYTES) { fprintf(stderr, "hipemu: %zu bytes of dynamic LDS requested\n", shmem); abort(); }
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    if (!nblocks || !(block.x * block.y * block.z)) return;
    std::lock_guard<std::mutex> launch_lock(g_launch_mutex);
    Job job;
    job.grid = grid; job.block = block; job.body = &thread_body; job.nblocks = nblocks; job.shmem = shmem;
    std::unique_lock<std::mutex> lk(g_mutex);
    if (g_pool.empty())
        for (int i = 0; i < pool_size(); i++) { g_pool.emplace_back(pool_main); g_pool.back().detach(); }
    g_job = &job;
    g_busy = (int)g_pool.size();
    g_generation++;
    g_cv_work.notify_all();
    g_cv_done.wait(lk, [&] { return g_busy == 0; });
    g_job = nullptr;
}

void barrier_block() {
    Worker* w = tl_worker;
    Fiber& f = w->fibers[w->cur];
    f.state = AT_BLOCK_BARRIER;
    to_scheduler(w, f);
}

int shfl(int v, int src_lane) {
    Worker* w = tl_worker;
    const int me = w->cur;
    Fiber& f = w->fibers[me];
    const uint32_t slot = f.shfl_seq++ & 1;           // two generations: a lane may reach the next shuffle before a slower one has read this one
    w->shfl_buf[slot][me] = v;
    f.state = AT_WAVE_BARRIER;
    to_scheduler(w, f);
    const int src = (me & ~63) + (src_lane & 63);
    return (uint32_t)src < w->nthreads ? w->shfl_buf[slot][src] : v;
}

void* dyn_smem() { return tl_worker->smem; }
}  // namespace hipemu

// ---------------------------------------------------------------------------------------------- host API
const char* hipGetErrorString(hipError_t e) {
    switch (e) {
    case hipSuccess: return "hipSuccess";
    case hipErrorInvalidValue: return "hipErrorInvalidValue";
    case hipErrorOutOfMemory: return "hipErrorOutOfMemory";
    default: return "hipError(emulated)";
    }
}
hipError_t hipGetLastError() { return hipSuccess; }
// HIPEMU_DEVICES "GPUs" (default 1) that all are this host: a multi-rank test gives every rank process its own device index
static int device_count() {
    const char* e = getenv("HIPEMU_DEVICES");
    return e ? std::max(1, atoi(e)) : 1;
}
static thread_local int tl_device = 0;
hipError_t hipSetDevice(int dev) {
    if (dev < 0 || dev >= device_count()) return hipErrorInvalidValue;
    tl_device = dev;
    return hipSuccess;
}
hipError_t hipGetDevice(int* dev) { *dev = tl_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = device_count(); return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "host emulation (tests/hostemu)");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "hostemu");
    p->totalGlobalMem = (size_t)8 << 30;
    p->sharedMemPerBlock = 160u << 10;
    p->maxSharedMemoryPerMultiProcessor = 160u << 10;
    const char* e = getenv("HIPEMU_CUS");
    p->multiProcessorCount = e ? std::max(1, atoi(e)) : 4;
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) {
    void* m = nullptr;
    if (posix_memalign(&m, 256, bytes ? bytes : 256)) { *p = nullptr; return hipErrorOutOfMemory; }
    if (fill_byte() >= 0) memset(m, fill_byte(), bytes);
    *p = m;
    return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes) { if (free_bytes) *free_bytes = 0; if (total_bytes) *total_bytes = 0; return hipSuccess; }   // not tracked
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, bytes, k); }
hipError_t hipMemset(void* dst, int value, size_t bytes) { if (bytes) memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) { return hipMemset(dst, value, bytes); }
hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
struct ihipEvent_t { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t* e) { *e = new ihipEvent_t(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
namespace {
std::mutex g_attr_mutex;
std::map<std::pair<int, const void*>, size_t> g_max_dyn_lds;          // (device, kernel) -> raised limit
}
hipError_t hipFuncSetAttribute(const void* fn, hipFuncAttribute attr, int value) {
    if (attr == hipFuncAttributeMaxDynamicSharedMemorySize) {
        if (value < 0 || (size_t)value > SMEM_BYTES) return hipErrorInvalidValue;
        std::lock_guard<std::mutex> g(g_attr_mutex);
        g_max_dyn_lds[{tl_device, fn}] = (size_t)value;
    }
    return hipSuccess;
}
namespace hipemu {
void check_dynamic_lds(const void* kernel, size_t shmem, const char* name) {
    if (shmem <= (64u << 10)) return;
    std::lock_guard<std::mutex> g(g_attr_mutex);
    auto it = g_max_dyn_lds.find({tl_device, kernel});
    if (it == g_max_dyn_lds.end() || it->second < shmem) {
        fprintf(stderr, "hipemu: %s launched with %zu bytes of dynamic LDS on device %d without hipFuncSetAttribute(MaxDynamicSharedMemorySize) >= that "
                        "on this device (limit in force: %zu) - the launch fails on a GPU\n", name, shmem, tl_device, it == g_max_dyn_lds.end() ? (size_t)(64u << 10) : it->second);
        abort();
    }
}
}  // namespace hipemu

// marks the library as the emulation: distributed_plonk_amd/_ffi.py refuses to load it unless the test harness opted in
extern "C" int plonk_hostemu_marker() { return 1; }
