// EXPERIMENT: static instruction counts of the two constant multipliers on gfx950 (no GPU needed):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S tools/experiments/isa_count.hip -o /tmp/isa_count.s
//   python tools/experiments/isa_count.py /tmp/isa_count.s
// Each kernel multiplies one lazy element by one constant fetched from memory, the way a butterfly does; the counts include the
// 9 (Montgomery) / 18 (Shoup) constant-limb loads' address arithmetic but no loop.
#include <hip/hip_runtime.h>
#include "f29_shoup.hpp"

__global__ void k_mont(const uint32_t* __restrict__ x, const uint32_t* __restrict__ tw, uint32_t* __restrict__ out, F29Params P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    F29 a, w;
#pragma unroll
    for (int l = 0; l < 9; l++) { a.l[l] = x[l * 65536 + i]; w.l[l] = tw[l * 128 + (i & 127)]; }
    const F29 r = f29_mul(a, w, P);
#pragma unroll
    for (int l = 0; l < 9; l++) out[l * 65536 + i] = r.l[l];
}

__global__ void k_shoup(const uint32_t* __restrict__ x, const uint32_t* __restrict__ tw, uint32_t* __restrict__ out, F29Shoup S) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    F29 a, w, wq;
#pragma unroll
    for (int l = 0; l < 9; l++) { a.l[l] = x[l * 65536 + i]; w.l[l] = tw[l * 128 + (i & 127)]; wq.l[l] = tw[(9 + l) * 128 + (i & 127)]; }
    const F29 r = f29_mul_shoup(a, w, wq, S);
#pragma unroll
    for (int l = 0; l < 9; l++) out[l * 65536 + i] = r.l[l];
}
