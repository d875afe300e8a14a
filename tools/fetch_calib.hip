// fetch_calib.hip — what does rocprofv3's FETCH_SIZE report on gfx950 for (a) a wide coalesced stream and (b) the MSM's access
// pattern: random 72-byte records gathered from a table that does not fit the Infinity Cache?  Known byte counts vs the counter.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o p -- /tmp/fetch_calib
// Expected useful bytes are printed; tools/fetch_calib_report.py divides.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void __launch_bounds__(256) calib_stream_kernel(const uint4* __restrict__ in, uint64_t n16, uint32_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += stride) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

struct Rec72 { uint32_t w[18]; };
__global__ void __launch_bounds__(256) calib_gather72_kernel(const Rec72* __restrict__ tab, uint64_t nrec, uint64_t count, uint32_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    const uint64_t j = (z ^ (z >> 31)) % nrec;
    const uint32_t* p = tab[j].w;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 18; k++) acc ^= p[k];
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint64_t nrec = 1ull << 24, count = 1ull << 26;
    const uint64_t stream_bytes = 4ull << 30;
    void *tab, *buf; uint32_t* out;
    hipMalloc(&tab, nrec * 72); hipMalloc(&buf, stream_bytes); hipMalloc(&out, 64);
    hipMemset(tab, 1, nrec * 72); hipMemset(buf, 1, stream_bytes);
    for (int it = 0; it < 3; it++) {
        hipLaunchKernelGGL(calib_stream_kernel, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)buf, stream_bytes / 16, out);
        hipLaunchKernelGGL(calib_gather72_kernel, dim3((uint32_t)(count / 256)), dim3(256), 0, 0, (const Rec72*)tab, nrec, count, out);
    }
    hipDeviceSynchronize();
    printf("calib_stream_kernel useful_bytes %llu\n", (unsigned long long)stream_bytes);
    printf("calib_gather72_kernel useful_bytes %llu (count %llu x 72 B from a %llu-byte table)\n", (unsigned long long)(count * 72),
           (unsigned long long)count, (unsigned long long)(nrec * 72));
    return 0;
}
