// microbench_valu.hip — measured issue rates of the VALU instructions a 256-bit Montgomery multiplier
// can be built from, on gfx950.  (MI355X_MICROARCH.md has no integer-multiply rows.)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o /tmp/mb && /tmp/mb
// Each kernel issues ITER x 16 independent instructions of one kind per lane; 256 CUs x 4 SIMDs x
// WAVES waves per SIMD.  Output: wave-instructions per clock per SIMD (assuming the measured s_memtime clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define ITER 2048

#define KERNEL16(NAME, ASMSTR, DECL, CONSTRAINTS)                                            \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed, uint64_t* cyc) { \
        DECL;                                                                                 \
        uint64_t t0 = __builtin_readcyclecounter();                                          \
        for (int it = 0; it < ITER; it++) {                                                   \
            asm volatile(ASMSTR CONSTRAINTS);                                                 \
        }                                                                                     \
        uint64_t t1 = __builtin_readcyclecounter();                                          \
        uint32_t acc = 0;                                                                     \
        FOLD;                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                     \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                              \
    }

// 8 independent 32-bit accumulators
#define DECL32 uint32_t r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7, a = seed * 3 + threadIdx.x, b = seed * 7 + 1
#define FOLD acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7
#define C32 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)
#define REP8(OP) OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)
#define TWICE(X) X X

#define OP_ADD(r) "v_add_u32 " #r ", " #r ", %8\n\t"
#define OP_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n\t"
#define OP_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", %8\n\t"
#define OP_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9\n\t"
#define OP_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", %8\n\t"
#define OP_MULHI24(r) "v_mul_hi_u32_u24 " #r ", " #r ", %8\n\t"
#define OP_MOV(r) "v_mov_b32 " #r ", %8\n\t"
#define OP_ADDCO(r) "v_add_co_u32 " #r ", vcc, " #r ", %8\n\t"
#define OP_ADDC(r) "v_addc_co_u32 " #r ", vcc, " #r ", %8, vcc\n\t"
#define OP_MADU16(r) "v_mad_u32_u16 " #r ", " #r ", %8, %9\n\t"
#define OP_XAD(r) "v_xad_u32 " #r ", " #r ", %8, %9\n\t"
#define OP_ADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9\n\t"
#define OP_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 1, %9\n\t"

KERNEL16(k_add, TWICE(REP8(OP_ADD)), DECL32, C32)
KERNEL16(k_mullo, TWICE(REP8(OP_MULLO)), DECL32, C32)
KERNEL16(k_mulhi, TWICE(REP8(OP_MULHI)), DECL32, C32)
KERNEL16(k_mad24, TWICE(REP8(OP_MAD24)), DECL32, C32)
KERNEL16(k_mul24, TWICE(REP8(OP_MUL24)), DECL32, C32)
KERNEL16(k_mulhi24, TWICE(REP8(OP_MULHI24)), DECL32, C32)
KERNEL16(k_mov, TWICE(REP8(OP_MOV)), DECL32, C32)
#undef C32
#define C32 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc"
KERNEL16(k_addco, TWICE(REP8(OP_ADDCO)), DECL32, C32)
KERNEL16(k_addc, TWICE(REP8(OP_ADDC)), DECL32, C32)
KERNEL16(k_add3, TWICE(REP8(OP_ADD3)), DECL32, C32)
KERNEL16(k_mad_u32_u16, TWICE(REP8(OP_MADU16)), DECL32, C32)

// 64-bit accumulators
#undef DECL32
#undef FOLD
#undef C32
#define DECL32 uint64_t r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7; uint32_t a = seed * 3 + threadIdx.x, b = seed * 7 + 1; uint64_t c64 = seed * 11
#define FOLD acc = (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) ^ (uint32_t)((r0 ^ r1 ^ r2 ^ r3) >> 32)
#define C32 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "v"(c64) : "vcc"
#define OP_MAD64(r) "v_mad_u64_u32 " #r ", vcc, %8, %9, " #r "\n\t"
#define OP_LSHLADD64(r) "v_lshl_add_u64 " #r ", " #r ", 0, %10\n\t"
KERNEL16(k_mad64, TWICE(REP8(OP_MAD64)), DECL32, C32)
KERNEL16(k_lshladd64, TWICE(REP8(OP_LSHLADD64)), DECL32, C32)
// dependent chains on the 64-bit accumulator (the multiplier's product scanning adds nine limb products into ONE accumulator per column):
// 16 mads per iteration on 1, 2 or 4 accumulators -> issue-to-dependent-issue latency of v_mad_u64_u32
#define OP16_DEP1 OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0) OP_MAD64(%0)
#define OP16_DEP2 OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%0) OP_MAD64(%1)
#define OP16_DEP4 OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%2) OP_MAD64(%3) OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%2) OP_MAD64(%3) OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%2) OP_MAD64(%3) OP_MAD64(%0) OP_MAD64(%1) OP_MAD64(%2) OP_MAD64(%3)
KERNEL16(k_mad64_dep1, OP16_DEP1, DECL32, C32)
KERNEL16(k_mad64_dep2, OP16_DEP2, DECL32, C32)
KERNEL16(k_mad64_dep4, OP16_DEP4, DECL32, C32)
// a dependent mad chain with the column hand-over of the multiplier: 8 mads, then acc >>= 29 (v_lshrrev_b64), repeated
#define OP_SHR64(r) "v_lshrrev_b64 " #r ", 29, " #r "\n\t"
#define OP8_COL(r) OP_MAD64(r) OP_MAD64(r) OP_MAD64(r) OP_MAD64(r) OP_MAD64(r) OP_MAD64(r) OP_MAD64(r) OP_SHR64(r)
KERNEL16(k_mad64_col1, OP8_COL(%0) OP8_COL(%0), DECL32, C32)
#define OP8_COL2(r, q) OP_MAD64(r) OP_MAD64(q) OP_MAD64(r) OP_MAD64(q) OP_MAD64(r) OP_MAD64(q) OP_MAD64(r) OP_MAD64(q) OP_MAD64(r) OP_MAD64(q) OP_MAD64(r) OP_MAD64(q) OP_MAD64(r) OP_MAD64(q) OP_SHR64(r) OP_SHR64(q)
KERNEL16(k_mad64_col2, OP8_COL2(%0, %1), DECL32, C32)
// the same dependent chain with the `s_nop 0` hipcc's hazard recognizer puts between an inline-asm VGPR definition and the next VALU reading it —
// what the "+v" accumulator pins of fp29.hpp / flimb.hpp cost until round 4 (9072 s_nop in the 2^8 NTT pass kernel, one per mad)
#define OP_MAD64_NOP(r) "v_mad_u64_u32 " #r ", vcc, %8, %9, " #r "\n\ts_nop 0\n\t"
#define OP16_DEP1_NOP OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0) OP_MAD64_NOP(%0)
KERNEL16(k_mad64_dep1_nop, OP16_DEP1_NOP, DECL32, C32)
// mad64 with an SGPR multiplier operand (modulus limb)
#undef C32
#define C32 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "s"(seed), "v"(c64) : "vcc"
KERNEL16(k_mad64_sgpr, TWICE(REP8(OP_MAD64)), DECL32, C32)

// fp64 fma
#undef DECL32
#undef FOLD
#undef C32
#define DECL32 double r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7, a = 1.0 + 1e-9 * threadIdx.x, b = 1e-7 * seed
#define FOLD acc = (uint32_t)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7)
#define C32 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)
#define OP_FMA64(r) "v_fma_f64 " #r ", " #r ", %8, %9\n\t"
#define OP_ADD64F(r) "v_add_f64 " #r ", " #r ", %9\n\t"
KERNEL16(k_fma64, TWICE(REP8(OP_FMA64)), DECL32, C32)
KERNEL16(k_addf64, TWICE(REP8(OP_ADD64F)), DECL32, C32)

typedef void (*kern_t)(uint32_t*, uint32_t, uint64_t*);
struct Entry { const char* name; kern_t k; };

int main() {
    Entry tests[] = {{"v_add_u32", k_add}, {"v_mov_b32", k_mov}, {"v_add_co_u32", k_addco}, {"v_addc_co_u32", k_addc},
                     {"v_add3_u32", k_add3}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi},
                     {"v_mad_u64_u32", k_mad64}, {"v_mad_u64_u32(sgpr)", k_mad64_sgpr}, {"mad64 chain x1", k_mad64_dep1}, {"mad64 chain x1 + s_nop 0", k_mad64_dep1_nop}, {"mad64 chain x2", k_mad64_dep2}, {"mad64 chain x4", k_mad64_dep4}, {"mad64 col x1 (7+shr)", k_mad64_col1}, {"mad64 col x2 (7+shr)", k_mad64_col2}, {"v_lshl_add_u64", k_lshladd64},
                     {"v_mad_u32_u24", k_mad24}, {"v_mul_u32_u24", k_mul24}, {"v_mul_hi_u32_u24", k_mulhi24},
                     {"v_mad_u32_u16", k_mad_u32_u16}, {"v_fma_f64", k_fma64}, {"v_add_f64", k_addf64}};
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, 256 * 4 * 8 * 256 * 4);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-22s %8s %8s | %10s %12s\n", "instruction", "waves/SIMD", "ms", "clk/instr/wave", "Ginstr/s(lane)");
    for (auto& t : tests) {
        for (int waves_per_simd : {1, 2, 3, 4}) {
            const int blocks = 256 * waves_per_simd;     // 256 threads = 4 waves = 1 per SIMD
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 3u, cyc);   // warm
            hipEventRecord(e0);
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 3u, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double instr_per_wave = (double)ITER * 16;
            // s_memtime ticks at 100 MHz constant on some parts; derive clocks from wall time @ nominal 2.4 GHz as well
            const double clk_wall = ms * 1e-3 * 2.4e9;
            const double per = clk_wall / (instr_per_wave * waves_per_simd);
            const double lanes_per_s = instr_per_wave * 64.0 * blocks * 4 / (ms * 1e-3) / 1e9;
            printf("%-22s %8d %8.3f | %10.2f %12.1f   (memtime delta %llu)\n", t.name, waves_per_simd, ms, per, lanes_per_s, (unsigned long long)c);
        }
    }
    return 0;
}
