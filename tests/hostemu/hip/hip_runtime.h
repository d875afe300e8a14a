// hip/hip_runtime.h of the HOST EMULATION used by tests/test_hostemu_*.py — TEST INFRASTRUCTURE, not a product path.
//
// This container has no GPU.  So that the kernels of distributed_plonk_amd/csrc/*.hip can still be EXECUTED by the CPU test
// suite (and under AddressSanitizer), tests/hostemu/build.py compiles the same sources with g++ against this header instead of
// ROCm's: a workgroup becomes a set of fibers on one OS thread (block-level barriers and the one wave-level broadcast the sources
// use are scheduler yields), LDS is thread-local storage of that OS thread, "device memory" is host memory, streams are
// synchronous.  It checks kernel LOGIC (indexing, the arithmetic, the host-side planning) bit-for-bit against the oracle at small
// sizes; it says nothing about performance, LDS capacity, register pressure or memory-model races — those stay with the `-m gpu`
// tests.  Nothing under distributed_plonk_amd/ knows about it: the emulation library is only ever loaded through the explicit
// PLONK_HIP_LIB override inside those tests, and _ffi.lib() refuses it unless PLONK_ALLOW_HOSTEMU=1 is set as well.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <tuple>
#include <utility>

#define HIPEMU 1
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ thread_local                     // block scope: implicitly static; a workgroup never leaves its OS thread
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)

struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipemu {
struct Idx3 { uint32_t x, y, z; };
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& thread_body);
// the dynamic-LDS rule of the real runtime: a launch asking for more than 64 KiB must have raised the kernel's limit with
// hipFuncSetAttribute(hipFuncAttributeMaxDynamicSharedMemorySize) ON THE DEVICE IT RUNS ON first (a launch that skips it fails on the GPU)
void check_dynamic_lds(const void* kernel, size_t shmem, const char* name);
void barrier_block();
int shfl(int v, int src_lane);
void* dyn_smem();
}  // namespace hipemu

extern thread_local hipemu::Idx3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;
static constexpr int warpSize = 64;

static inline void __syncthreads() { hipemu::barrier_block(); }
static inline int __shfl(int v, int src_lane) { return hipemu::shfl(v, src_lane); }
static inline uint32_t __brev(uint32_t x) {
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    return ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
}
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __clz(int x) { return x ? __builtin_clz((uint32_t)x) : 32; }

// workgroups of one launch run on several OS threads: global-memory atomics are real atomics
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicMax(uint32_t* p, uint32_t v) {
    uint32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline uint64_t min(uint64_t a, uint64_t b) { return a < b ? a : b; }
static inline uint64_t max(uint64_t a, uint64_t b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------- host API (synchronous)
typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 } hipError_t;
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;
static constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    size_t sharedMemPerBlock;
    size_t maxSharedMemoryPerMultiProcessor;
    int multiProcessorCount;
    int warpSize;
};

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipSetDevice(int dev);
hipError_t hipGetDevice(int* dev);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int dev);
hipError_t hipMalloc(void** p, size_t bytes);
template <typename T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
hipError_t hipFree(void* p);
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s = nullptr);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void* fn, hipFuncAttribute attr, int value);
template <typename F> static inline hipError_t hipFuncSetAttribute(F fn, hipFuncAttribute attr, int value) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fn), attr, value);
}

// The arguments are evaluated ONCE (as a real launch does) and every emulated thread calls the kernel with them.
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                                              \
    do {                                                                                                                         \
        auto _hipemu_args = std::make_tuple(__VA_ARGS__);                                                                        \
        (void)(stream);                                                                                                          \
        hipemu::check_dynamic_lds(reinterpret_cast<const void*>(&kernel), (size_t)(shmem), #kernel);                             \
        hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem),                                                                 \
                       [&]() { std::apply([](auto&... _a) { kernel(_a...); }, _hipemu_args); });                                 \
    } while (0)
