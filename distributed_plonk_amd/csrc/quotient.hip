// quotient.hip — coset evaluations of the TurboPlonk quotient polynomial, pointwise over the m = 8n points
// x_i = g * w_m^i  (the kernel that sits between the 25 coset-NTTs and the quotient's
// coset-iNTT + 5 commitments; keeping it on the device removes a 25 x m x 32 B host round trip).
//
// What it replaces: the serial loop of /root/reference/src/dispatcher2.rs:435-504 (gate equation :459-477,
// permutation argument :479-495, L1 term :497-503, 1/Z_H factor :372-379).  Streaming kernel: 26 loads + 2 table
// loads and one store of 32 B per point; 52 limb products + 8 squarings but only 50 Montgomery reductions: the selector sums and
// the final combination are three-term dot products with one reduction each (fp29.hpp: f29_dot<3>).
//
// Representation bookkeeping (mont(a, b) = a*b/2^261; inputs arrive in the reference's R = 2^256 Montgomery form, the kernel's
// radix is R' = 2^261).  Default kernel (quotient_evals_kernel): NOTHING is lifted.  A product of two R-form values comes out as
// (ab)*R*2^-5, so a monomial of depth k carries 2^-5k: the gate's terms fall into three classes — q_lc*w and q_o*e (2^-5),
// q_mul*w*w (2^-10), q_hash*w^5 and q_ecc*abcde (2^-25) — each summed lazily and brought back to R form by ONE product with
// 2^(261+5k); the permutation products carry 2^-25 on both sides of the difference and the factor is folded into the alpha/Z_H
// constant; every other constant is stored in whichever of the two forms makes its product land in R form.  56 products (8 of
// them squarings) per point instead of 60 with lifted wires.
// The fused variants (quotient_evals_kernel_f2/f3, option "quotient_fuse") work on lifted (R') wires:
//   wires, z            R  --(* 2^266)-->  R' = 2^261 form        (products of R' values stay in R')
//   selector * (R' value) -> R form ;  sigma * (beta*2^266) -> R' ;  (R' value) * (alpha*2^256) -> R form
#include <cstdlib>
#include <cstring>

#include "constants.h"
#include "ntt_kernels.hpp"
#include "plonk_internal.hpp"

struct QuotParams {
    const Fr* sel[13];
    const Fr* sig[5];
    const Fr* wire[5];
    const Fr* z;
    const Fr* pi;
    Fr* out;
    uint64_t m;            // points of the whole quotient domain
    uint64_t m_local;      // points this launch evaluates: index k <-> global point j = cls_offset + cls_stride * k
    uint32_t cls_stride, cls_offset;
    uint32_t ratio, lt, x_shift;
    const F29* x_lo;       // g * w_Nmax^e       (constant form = R' form)
    const F29* x_hi;       // w_Nmax^(e << lt)
    const Fr* inv_xm1;     // 1 / (x_j - 1) for the points of this class, R' form, canonical, packed (index k)
    F29Params fp;
    F29 r2fix;             // 2^266            : R  -> R'
    F29 gamma_rp;          // gamma * 2^261
    F29 kbeta_rp[5];       // k_j * beta * 2^261
    F29 beta_fix;          // beta * 2^266     : sigma (R) -> sigma*beta (R')
    F29 a2n_r;             // alpha^2/n * 2^256
    F29 zh_inv_rp[8];      // 1/Z_H(x_i) * 2^261, i < m/n
    F29 zh_alpha_r[8];     // alpha/Z_H(x_i) * 2^256
    // unlifted kernel
    F29 fix5, fix10, fix25;   // 2^(261+5), 2^(261+10), 2^(261+25): (value * R * 2^-5k) -> value * R
    F29 gamma_r;              // gamma * 2^256
    F29 kbeta_r[5];           // k_j * beta * 2^256    : x (2^261 form) -> k_j beta x in R form
    F29 beta_c;               // beta * 2^261          : sigma (R) -> sigma beta (R)
    F29 zh_alpha_25[8];       // alpha/Z_H(x_i) * 2^(261+25)
    F29 a2n_c;                // alpha^2/n * 2^261
    F29 one_r;                // 2^256
};

struct LazySum {           // sum of normalised values; limbs re-normalised every third addition
    F29 v;
    int pending;
    __device__ __forceinline__ void init(const F29& a) { v = a; pending = 0; }
    __device__ __forceinline__ void add(const F29& a) {
        v = f29_add(v, a);
        if (++pending == 3) { f29_norm(v); pending = 0; }
    }
    __device__ __forceinline__ F29 get() { if (pending) { f29_norm(v); pending = 0; } return v; }
};

__device__ __forceinline__ F29 ldq(const Fr* p, uint64_t i) { return f29_from_sat(load_fr(p + i)); }

// one evaluation point (local index i)
__device__ __forceinline__ void quotient_point_unlifted(const QuotParams& P, const uint64_t i);

__device__ __forceinline__ void quotient_evals_unlifted(const QuotParams& P) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // local index k
    if (i >= P.m_local) return;
    quotient_point_unlifted(P, i);
}
// The point's arithmetic is ~12 000 straight-line instructions (68 KiB of code: more than the 64 KiB instruction cache two CUs
// share), executed exactly once per wave in the kernels above, so every wave streams the whole kernel through the instruction
// cache.  Here a wave walks PTS points in a rolled loop: the code is fetched once per PTS points (profiles/r03_quotient_slow_mode.txt).
template <int PTS>
__device__ __forceinline__ void quotient_evals_unlifted_loop(const QuotParams& P) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll 1
    for (int it = 0; it < PTS; it++, i += stride) {
        if (i < P.m_local) quotient_point_unlifted(P, i);
    }
}

__device__ __forceinline__ void quotient_point_unlifted(const QuotParams& P, const uint64_t i) {
    const uint64_t j = (uint64_t)P.cls_offset + (uint64_t)P.cls_stride * i;  // global point index
    const F29Params& fp = P.fp;
    F29 w5[5];
#pragma unroll
    for (int q = 0; q < 5; q++) w5[q] = ldq(P.wire[q], i);                   // R form, canonical
    const F29 &a = w5[0], &b = w5[1], &c = w5[2], &d = w5[3], &e = w5[4];

    // ---- gate equation (dispatcher2.rs:459-477), three depth classes; each class is folded into the sum as soon as it is complete
    // so that few partial values stay alive (the kernel runs four waves per SIMD: 128 VGPRs)
    LazySum g;
    g.init(ldq(P.sel[11], i));                                               // q_c
    g.add(ldq(P.pi, i));                                                     // + pub_input
    {
        LazySum t1;                                                          // depth 1: * R * 2^-5
        t1.init(f29_mul(ldq(P.sel[0], i), a, fp));
        t1.add(f29_mul(ldq(P.sel[1], i), b, fp));
        t1.add(f29_mul(ldq(P.sel[2], i), c, fp));
        t1.add(f29_mul(ldq(P.sel[3], i), d, fp));
        F29 s1 = f29_sub2p(t1.get(), f29_mul(ldq(P.sel[10], i), e, fp), fp); // - q_o * e
        f29_norm(s1);
        g.add(f29_mul(s1, P.fix5, fp));
    }
    __builtin_amdgcn_sched_barrier(0);
    F29 abcde;
    {
        const F29 ab = f29_mul(a, b, fp), cd = f29_mul(c, d, fp);            // * R * 2^-5
        F29 s2 = f29_add(f29_mul(ldq(P.sel[4], i), ab, fp), f29_mul(ldq(P.sel[5], i), cd, fp));     // depth 2: * R * 2^-10
        f29_norm(s2);
        g.add(f29_mul(s2, P.fix10, fp));
        abcde = f29_mul(f29_mul(ab, cd, fp), e, fp);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        LazySum t3;                                                          // depth 5: * R * 2^-25
        t3.init(f29_mul(ldq(P.sel[12], i), abcde, fp));                      // q_ecc * ab * cd * e
#pragma unroll
        for (int q = 0; q < 4; q++) t3.add(f29_mul(ldq(P.sel[6 + q], i), f29_mul(f29_sqr(f29_sqr(w5[q], fp), fp), w5[q], fp), fp));   // q_hash * w^5
        g.add(f29_mul(t3.get(), P.fix25, fp));
    }
    __builtin_amdgcn_sched_barrier(0);
    const F29 gate = g.get();                                                // R form, < 6.2 p

    // ---- permutation argument (:479-495): both products carry 2^-25
    const F29 zc = ldq(P.z, i);
    const uint64_t E = j << P.x_shift, mask = ((uint64_t)1 << P.lt) - 1;
    const F29 x = f29_mul(load_f29(P.x_lo + (E & mask)), load_f29(P.x_hi + ((E >> P.lt) & mask)), fp);       // x * 2^261
    F29 acc1 = zc, acc2 = ldq(P.z, (i + P.ratio / P.cls_stride) & (P.m_local - 1));     // z(x), z(w x): point j + ratio, same class
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const F29 t = f29_add(w5[q], P.gamma_r);                                         // limbs < 2^30
        const F29 u = f29_add(t, f29_mul(x, P.kbeta_r[q], fp));                          // w + gamma + k_j x beta   (< 3.4 p)
        const F29 v = f29_add(t, f29_mul(ldq(P.sig[q], i), P.beta_c, fp));               // w + gamma + sigma_j beta
        acc1 = f29_mul(u, acc1, fp);
        acc2 = f29_mul(v, acc2, fp);
    }
    F29 diff = f29_sub2p(acc1, acc2, fp);
    f29_norm(diff);                                                                      // (acc1 - acc2) * R * 2^-25, < 3.4 p

    // ---- 1/Z_H(x) * (gate + alpha * (acc1 - acc2)) + alpha^2/n * (z(x) - 1)/(x - 1)   (:497-503, :372-379)
    F29 zm1 = f29_sub2p(zc, P.one_r, fp);
    f29_norm(zm1);
    const uint32_t ci = (uint32_t)(j & (P.ratio - 1));
    LazySum r;
    r.init(f29_mul(gate, P.zh_inv_rp[ci], fp));
    r.add(f29_mul(diff, P.zh_alpha_25[ci], fp));
    r.add(f29_mul(f29_mul(zm1, ldq(P.inv_xm1, i), fp), P.a2n_c, fp));
    F29 out = r.get();                                                                   // < 4.1 p
    out = f29_canon_lazy(out, fp);
    store_fr(P.out + i, f29_to_sat(out));
}

// The COMPACT formulation (round 4; option quotient_fuse = 6, the default): the same arithmetic as quotient_point_unlifted —
// same products, same lazy classes, same constants, same result bits — with the two repetitive parts written as loops of one body
// (#pragma unroll 1): the four q_hash * w^5 terms and the five wire factors of the permutation argument re-read their wire value (an L1 / L2
// hit: the point loaded it for the linear terms a few hundred instructions earlier) and pick their selector / sigma pointer and their k_j*beta
// constant from the kernel-argument block with the scalar loop counter, so the five wire values no longer stay in registers across the whole
// point and nothing spills (the round 2-3 kernel: 208 B of scratch per lane at its 128-VGPR cap; this one: 141 VGPRs, none).
// What it does NOT do is shrink the code: hipcc had left the same loops of quotient_point_unlifted rolled already (its unroll budget), and
// llvm-readelf gives 67 656 bytes for this kernel against 67 888 — both still at the edge of the 64 KiB instruction cache two CUs share
// (profiles/r03_quotient_slow_mode.txt).  Measured: -3 % on two boxes (56.0 -> 54.4 ms, 56.5 -> 54.7 ms at 2^27 points).
__device__ __forceinline__ void quotient_point_compact(const QuotParams& P, const uint64_t i) {
    // Operand loads are issued ONE PRODUCT AHEAD of their use (raw 256-bit values, 8 VGPRs each, converted at the use) with a scheduling
    // barrier behind every batch of loads, so that no product waits for the load in front of it.  The kernel is bound by VALU issue either way
    // (round 4 counters: 12.5 k VALU instructions per point = 98-100 % of the launch's busy cycles): the prefetch is worth what the lost
    // scratch traffic is, no more.
    const uint64_t j = (uint64_t)P.cls_offset + (uint64_t)P.cls_stride * i;  // global point index
    const F29Params& fp = P.fp;
#define QLOAD(ptr) load_fr((ptr) + i)
#define QFENCE() __builtin_amdgcn_sched_barrier(0)
    Fr ra = QLOAD(P.wire[0]), rb = QLOAD(P.wire[1]), rs0 = QLOAD(P.sel[0]), rs1 = QLOAD(P.sel[1]);
    QFENCE();
    F29 abcde;
    LazySum g;
    {
        const F29 a = f29_from_sat(ra), b = f29_from_sat(rb);
        Fr rc = QLOAD(P.wire[2]), rs2 = QLOAD(P.sel[2]);
        QFENCE();
        LazySum t1;                                                          // depth 1: * R * 2^-5
        t1.init(f29_mul(f29_from_sat(rs0), a, fp));
        Fr rd = QLOAD(P.wire[3]), rs3 = QLOAD(P.sel[3]);
        QFENCE();
        t1.add(f29_mul(f29_from_sat(rs1), b, fp));
        const F29 c = f29_from_sat(rc);
        Fr re = QLOAD(P.wire[4]), rs10 = QLOAD(P.sel[10]);
        QFENCE();
        t1.add(f29_mul(f29_from_sat(rs2), c, fp));
        const F29 d = f29_from_sat(rd);
        Fr rs4 = QLOAD(P.sel[4]);
        QFENCE();
        t1.add(f29_mul(f29_from_sat(rs3), d, fp));
        const F29 e = f29_from_sat(re);
        Fr rs5 = QLOAD(P.sel[5]);
        QFENCE();
        F29 s1 = f29_sub2p(t1.get(), f29_mul(f29_from_sat(rs10), e, fp), fp);  // - q_o * e
        f29_norm(s1);
        Fr rs11 = QLOAD(P.sel[11]), rpi = QLOAD(P.pi);
        QFENCE();
        const F29 g1 = f29_mul(s1, P.fix5, fp);
        g.init(f29_from_sat(rs11));                                          // q_c
        g.add(f29_from_sat(rpi));                                            // + pub_input
        g.add(g1);
        QFENCE();
        const F29 ab = f29_mul(a, b, fp), cd = f29_mul(c, d, fp);            // * R * 2^-5
        Fr rs12 = QLOAD(P.sel[12]);
        QFENCE();
        F29 s2 = f29_add(f29_mul(f29_from_sat(rs4), ab, fp), f29_mul(f29_from_sat(rs5), cd, fp));     // depth 2: * R * 2^-10
        f29_norm(s2);
        g.add(f29_mul(s2, P.fix10, fp));
        QFENCE();
        abcde = f29_mul(f29_mul(f29_mul(ab, cd, fp), e, fp), f29_from_sat(rs12), fp);       // q_ecc * ab * cd * e : depth 5, * R * 2^-25
    }
    QFENCE();
    {
        F29 t3 = abcde;
        Fr rw = QLOAD(P.wire[0]), rs = QLOAD(P.sel[6]);
#pragma unroll 1
        for (int q = 0; q < 4; q++) {                                        // + q_hash[q] * w_q^5
            const F29 w = f29_from_sat(rw), sq = f29_from_sat(rs);
            const int qn = (q + 1) & 3;                                      // the last iteration re-reads wire 0 / q_hash[0]: harmless
            rw = QLOAD(P.wire[qn]);
            rs = QLOAD(P.sel[6 + qn]);
            QFENCE();
            t3 = f29_add(t3, f29_mul(sq, f29_mul(f29_sqr(f29_sqr(w, fp), fp), w, fp), fp));
            f29_norm(t3);                                                    // at most five normalised values: < 6.8 p
        }
        g.add(f29_mul(t3, P.fix25, fp));
    }
    QFENCE();
    const F29 gate = g.get();                                                // R form, < 6.2 p

    // ---- permutation argument (:479-495): both products carry 2^-25
    Fr rz = QLOAD(P.z), rzn = load_fr(P.z + ((i + P.ratio / P.cls_stride) & (P.m_local - 1)));          // z(x), z(w x): point j + ratio, same class
    Fr rw = QLOAD(P.wire[0]), rsg = QLOAD(P.sig[0]);
    QFENCE();
    const uint64_t E = j << P.x_shift, mask = ((uint64_t)1 << P.lt) - 1;
    const F29 x = f29_mul(load_f29(P.x_lo + (E & mask)), load_f29(P.x_hi + ((E >> P.lt) & mask)), fp);       // x * 2^261
    const F29 zc = f29_from_sat(rz);
    F29 acc1 = zc, acc2 = f29_from_sat(rzn);
#pragma unroll 1
    for (int q = 0; q < 5; q++) {
        const F29 t = f29_add(f29_from_sat(rw), P.gamma_r);                              // limbs < 2^30
        const F29 sg = f29_from_sat(rsg);
        const int qn = q < 4 ? q + 1 : 0;
        rw = QLOAD(P.wire[qn]);
        rsg = QLOAD(P.sig[qn]);
        QFENCE();
        const F29 u = f29_add(t, f29_mul(x, P.kbeta_r[q], fp));                          // w + gamma + k_j x beta   (< 3.4 p)
        const F29 v = f29_add(t, f29_mul(sg, P.beta_c, fp));                             // w + gamma + sigma_j beta
        acc1 = f29_mul(u, acc1, fp);
        acc2 = f29_mul(v, acc2, fp);
    }
    Fr rinv = QLOAD(P.inv_xm1);
    QFENCE();
    F29 diff = f29_sub2p(acc1, acc2, fp);
    f29_norm(diff);                                                                      // (acc1 - acc2) * R * 2^-25, < 3.4 p

    // ---- 1/Z_H(x) * (gate + alpha * (acc1 - acc2)) + alpha^2/n * (z(x) - 1)/(x - 1)   (:497-503, :372-379)
    F29 zm1 = f29_sub2p(zc, P.one_r, fp);
    f29_norm(zm1);
    const uint32_t ci = (uint32_t)(j & (P.ratio - 1));
    LazySum r;
    r.init(f29_mul(gate, P.zh_inv_rp[ci], fp));
    r.add(f29_mul(diff, P.zh_alpha_25[ci], fp));
    r.add(f29_mul(f29_mul(zm1, f29_from_sat(rinv), fp), P.a2n_c, fp));
    F29 out = r.get();                                                                   // < 4.1 p
    out = f29_canon_lazy(out, fp);
    store_fr(P.out + i, f29_to_sat(out));
#undef QLOAD
#undef QFENCE
}

// The SPLIT formulation (round 5; option quotient_fuse = 8): the compact point cut in two kernels of about half the code
// each (llvm-readelf: see profiles/r05_quotient_split.txt), so that either fits the 64 KiB instruction cache two CUs share with room to
// spare: quotient_gate_kernel (13 selectors, 5 wires, pub_input -> gate / Z_H, 32 products, stored to `out`) and quotient_perm_kernel (5 wires,
// 5 sigmas, z, z(wx), 1/(x-1), the partial -> out in place, 24 products).  Same products, same lazy classes, same result bits; the price is the
// partial's round trip and a second read of the wires: 34 instead of 28 32-byte accesses per point.
__device__ __forceinline__ void quotient_point_gate(const QuotParams& P, const uint64_t i) {
    // Operand loads are issued ONE PRODUCT AHEAD of their use (raw 256-bit values, 8 VGPRs each, converted at the use) with a scheduling
    // barrier behind every batch of loads, so that no product waits for the load in front of it.  The kernel is bound by VALU issue either way
    // (round 4 counters: 12.5 k VALU instructions per point = 98-100 % of the launch's busy cycles): the prefetch is worth what the lost
    // scratch traffic is, no more.
    const uint64_t j = (uint64_t)P.cls_offset + (uint64_t)P.cls_stride * i;  // global point index
    const F29Params& fp = P.fp;
#define QLOAD(ptr) load_fr((ptr) + i)
#define QFENCE() __builtin_amdgcn_sched_barrier(0)
    Fr ra = QLOAD(P.wire[0]), rb = QLOAD(P.wire[1]), rs0 = QLOAD(P.sel[0]), rs1 = QLOAD(P.sel[1]);
    QFENCE();
    F29 abcde;
    LazySum g;
    {
        const F29 a = f29_from_sat(ra), b = f29_from_sat(rb);
        Fr rc = QLOAD(P.wire[2]), rs2 = QLOAD(P.sel[2]);
        QFENCE();
        LazySum t1;                                                          // depth 1: * R * 2^-5
        t1.init(f29_mul(f29_from_sat(rs0), a, fp));
        Fr rd = QLOAD(P.wire[3]), rs3 = QLOAD(P.sel[3]);
        QFENCE();
        t1.add(f29_mul(f29_from_sat(rs1), b, fp));
        const F29 c = f29_from_sat(rc);
        Fr re = QLOAD(P.wire[4]), rs10 = QLOAD(P.sel[10]);
        QFENCE();
        t1.add(f29_mul(f29_from_sat(rs2), c, fp));
        const F29 d = f29_from_sat(rd);
        Fr rs4 = QLOAD(P.sel[4]);
        QFENCE();
        t1.add(f29_mul(f29_from_sat(rs3), d, fp));
        const F29 e = f29_from_sat(re);
        Fr rs5 = QLOAD(P.sel[5]);
        QFENCE();
        F29 s1 = f29_sub2p(t1.get(), f29_mul(f29_from_sat(rs10), e, fp), fp);  // - q_o * e
        f29_norm(s1);
        Fr rs11 = QLOAD(P.sel[11]), rpi = QLOAD(P.pi);
        QFENCE();
        const F29 g1 = f29_mul(s1, P.fix5, fp);
        g.init(f29_from_sat(rs11));                                          // q_c
        g.add(f29_from_sat(rpi));                                            // + pub_input
        g.add(g1);
        QFENCE();
        const F29 ab = f29_mul(a, b, fp), cd = f29_mul(c, d, fp);            // * R * 2^-5
        Fr rs12 = QLOAD(P.sel[12]);
        QFENCE();
        F29 s2 = f29_add(f29_mul(f29_from_sat(rs4), ab, fp), f29_mul(f29_from_sat(rs5), cd, fp));     // depth 2: * R * 2^-10
        f29_norm(s2);
        g.add(f29_mul(s2, P.fix10, fp));
        QFENCE();
        abcde = f29_mul(f29_mul(f29_mul(ab, cd, fp), e, fp), f29_from_sat(rs12), fp);       // q_ecc * ab * cd * e : depth 5, * R * 2^-25
    }
    QFENCE();
    {
        F29 t3 = abcde;
        Fr rw = QLOAD(P.wire[0]), rs = QLOAD(P.sel[6]);
#pragma unroll 1
        for (int q = 0; q < 4; q++) {                                        // + q_hash[q] * w_q^5
            const F29 w = f29_from_sat(rw), sq = f29_from_sat(rs);
            const int qn = (q + 1) & 3;                                      // the last iteration re-reads wire 0 / q_hash[0]: harmless
            rw = QLOAD(P.wire[qn]);
            rs = QLOAD(P.sel[6 + qn]);
            QFENCE();
            t3 = f29_add(t3, f29_mul(sq, f29_mul(f29_sqr(f29_sqr(w, fp), fp), w, fp), fp));
            f29_norm(t3);                                                    // at most five normalised values: < 6.8 p
        }
        g.add(f29_mul(t3, P.fix25, fp));
    }
    QFENCE();
    const F29 gate = g.get();                                                // R form, < 6.2 p
    const uint32_t ci = (uint32_t)(j & (P.ratio - 1));
    store_fr(P.out + i, f29_to_sat(f29_canon(f29_mul(gate, P.zh_inv_rp[ci], fp), fp)));    // gate / Z_H(x): the partial the second kernel completes in place
#undef QLOAD
#undef QFENCE
}
__device__ __forceinline__ void quotient_point_perm(const QuotParams& P, const uint64_t i) {
    // Operand loads are issued ONE PRODUCT AHEAD of their use (raw 256-bit values, 8 VGPRs each, converted at the use) with a scheduling
    // barrier behind every batch of loads, so that no product waits for the load in front of it.  The kernel is bound by VALU issue either way
    // (round 4 counters: 12.5 k VALU instructions per point = 98-100 % of the launch's busy cycles): the prefetch is worth what the lost
    // scratch traffic is, no more.
    const uint64_t j = (uint64_t)P.cls_offset + (uint64_t)P.cls_stride * i;  // global point index
    const F29Params& fp = P.fp;
#define QLOAD(ptr) load_fr((ptr) + i)
#define QFENCE() __builtin_amdgcn_sched_barrier(0)
    // ---- permutation argument (:479-495): both products carry 2^-25
    Fr rz = QLOAD(P.z), rzn = load_fr(P.z + ((i + P.ratio / P.cls_stride) & (P.m_local - 1)));          // z(x), z(w x): point j + ratio, same class
    Fr rw = QLOAD(P.wire[0]), rsg = QLOAD(P.sig[0]);
    QFENCE();
    const uint64_t E = j << P.x_shift, mask = ((uint64_t)1 << P.lt) - 1;
    const F29 x = f29_mul(load_f29(P.x_lo + (E & mask)), load_f29(P.x_hi + ((E >> P.lt) & mask)), fp);       // x * 2^261
    const F29 zc = f29_from_sat(rz);
    F29 acc1 = zc, acc2 = f29_from_sat(rzn);
#pragma unroll 1
    for (int q = 0; q < 5; q++) {
        const F29 t = f29_add(f29_from_sat(rw), P.gamma_r);                              // limbs < 2^30
        const F29 sg = f29_from_sat(rsg);
        const int qn = q < 4 ? q + 1 : 0;
        rw = QLOAD(P.wire[qn]);
        rsg = QLOAD(P.sig[qn]);
        QFENCE();
        const F29 u = f29_add(t, f29_mul(x, P.kbeta_r[q], fp));                          // w + gamma + k_j x beta   (< 3.4 p)
        const F29 v = f29_add(t, f29_mul(sg, P.beta_c, fp));                             // w + gamma + sigma_j beta
        acc1 = f29_mul(u, acc1, fp);
        acc2 = f29_mul(v, acc2, fp);
    }
    Fr rinv = QLOAD(P.inv_xm1);
    QFENCE();
    F29 diff = f29_sub2p(acc1, acc2, fp);
    f29_norm(diff);                                                                      // (acc1 - acc2) * R * 2^-25, < 3.4 p

    // ---- 1/Z_H(x) * (gate + alpha * (acc1 - acc2)) + alpha^2/n * (z(x) - 1)/(x - 1)   (:497-503, :372-379)
    F29 zm1 = f29_sub2p(zc, P.one_r, fp);
    f29_norm(zm1);
    const uint32_t ci = (uint32_t)(j & (P.ratio - 1));
    Fr rpart = QLOAD(P.out);                                                             // gate / Z_H(x), written by quotient_gate_kernel
    QFENCE();
    LazySum r;
    r.init(f29_from_sat(rpart));
    r.add(f29_mul(diff, P.zh_alpha_25[ci], fp));
    r.add(f29_mul(f29_mul(zm1, f29_from_sat(rinv), fp), P.a2n_c, fp));
    F29 out = r.get();                                                                   // < 3.8 p
    out = f29_canon_lazy(out, fp);
    store_fr(P.out + i, f29_to_sat(out));
#undef QLOAD
#undef QFENCE
}

// FUSE = 1: one Montgomery reduction per product (60 of them).  FUSE = 2 / 3: the twelve selector * monomial terms of the gate
// equation and the final combination are taken two / three at a time with ONE reduction per group (f29_dot2 / f29_dot3): 54 / 50
// reductions for the same 60 limb products, at the price of 4 / 6 operands alive per group.
template <int FUSE>
__device__ __forceinline__ void quotient_evals_body(const QuotParams& P) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // local index k
    if (i >= P.m_local) return;
    const uint64_t j = (uint64_t)P.cls_offset + (uint64_t)P.cls_stride * i;  // global point index
    const F29Params& fp = P.fp;
    // wires in R' form
    F29 w5[5];
#pragma unroll
    for (int q = 0; q < 5; q++) w5[q] = f29_mul(ldq(P.wire[q], i), P.r2fix, fp);
    const F29 &a = w5[0], &b = w5[1], &c = w5[2], &d = w5[3], &e = w5[4];

    // ---- gate equation (dispatcher2.rs:459-477); ordered so that few monomials are alive at a time (128-VGPR budget)
    LazySum g;
    g.init(ldq(P.sel[11], i));                                   // q_c
    g.add(ldq(P.pi, i));                                         // + pub_input
    F29 neg_qo = ldq(P.sel[10], i);
#pragma unroll
    for (int l = 0; l < 9; l++) neg_qo.l[l] = fp.c2p[l] - neg_qo.l[l];              // 2p - q_o : the - q_o * e term joins the sums
    f29_norm(neg_qo);
    if (FUSE == 3) {
        g.add(f29_dot3(ldq(P.sel[0], i), a, ldq(P.sel[1], i), b, ldq(P.sel[2], i), c, fp));                      // q_lc[0..2]
        __builtin_amdgcn_sched_barrier(0);
        F29 abcde;
        {
            const F29 ab = f29_mul(a, b, fp), cd = f29_mul(c, d, fp);
            g.add(f29_dot3(ldq(P.sel[3], i), d, ldq(P.sel[4], i), ab, ldq(P.sel[5], i), cd, fp));                // q_lc[3], q_mul[0..1]
            abcde = f29_mul(f29_mul(ab, cd, fp), e, fp);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const F29 h0 = f29_mul(f29_sqr(f29_sqr(a, fp), fp), a, fp), h1 = f29_mul(f29_sqr(f29_sqr(b, fp), fp), b, fp);   // w^5
            g.add(f29_dot3(ldq(P.sel[12], i), abcde, ldq(P.sel[6], i), h0, ldq(P.sel[7], i), h1, fp));           // q_ecc, q_hash[0..1]
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const F29 h2 = f29_mul(f29_sqr(f29_sqr(c, fp), fp), c, fp), h3 = f29_mul(f29_sqr(f29_sqr(d, fp), fp), d, fp);
            g.add(f29_dot3(ldq(P.sel[8], i), h2, ldq(P.sel[9], i), h3, neg_qo, e, fp));                          // q_hash[2..3], -q_o
        }
    } else if (FUSE == 2) {
        g.add(f29_dot2(ldq(P.sel[0], i), a, ldq(P.sel[1], i), b, fp));
        g.add(f29_dot2(ldq(P.sel[2], i), c, ldq(P.sel[3], i), d, fp));
        __builtin_amdgcn_sched_barrier(0);
        {
            const F29 ab = f29_mul(a, b, fp), cd = f29_mul(c, d, fp);
            g.add(f29_dot2(ldq(P.sel[4], i), ab, ldq(P.sel[5], i), cd, fp));
            const F29 abcde = f29_mul(f29_mul(ab, cd, fp), e, fp);
            g.add(f29_dot2(ldq(P.sel[12], i), abcde, neg_qo, e, fp));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const F29 h0 = f29_mul(f29_sqr(f29_sqr(a, fp), fp), a, fp), h1 = f29_mul(f29_sqr(f29_sqr(b, fp), fp), b, fp);
            g.add(f29_dot2(ldq(P.sel[6], i), h0, ldq(P.sel[7], i), h1, fp));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const F29 h2 = f29_mul(f29_sqr(f29_sqr(c, fp), fp), c, fp), h3 = f29_mul(f29_sqr(f29_sqr(d, fp), fp), d, fp);
            g.add(f29_dot2(ldq(P.sel[8], i), h2, ldq(P.sel[9], i), h3, fp));
        }
    } else {
        const F29 ab = f29_mul(a, b, fp), cd = f29_mul(c, d, fp);
        g.add(f29_mul(ldq(P.sel[0], i), a, fp));
        g.add(f29_mul(ldq(P.sel[1], i), b, fp));
        g.add(f29_mul(ldq(P.sel[2], i), c, fp));
        g.add(f29_mul(ldq(P.sel[3], i), d, fp));
        g.add(f29_mul(ldq(P.sel[4], i), ab, fp));
        g.add(f29_mul(ldq(P.sel[5], i), cd, fp));
        g.add(f29_mul(ldq(P.sel[12], i), f29_mul(f29_mul(ab, cd, fp), e, fp), fp));
#pragma unroll
        for (int q = 0; q < 4; q++) g.add(f29_mul(ldq(P.sel[6 + q], i), f29_mul(f29_sqr(f29_sqr(w5[q], fp), fp), w5[q], fp), fp));
        g.add(f29_mul(neg_qo, e, fp));
    }
    __builtin_amdgcn_sched_barrier(0);
    const F29 gate = g.get();                                    // normalised, < 8 p

    // ---- evaluation point x_i = g * w_m^i and the permutation argument (:479-495)
    const F29 zc = f29_mul(ldq(P.z, i), P.r2fix, fp);
    const F29 zn = f29_mul(ldq(P.z, (i + P.ratio / P.cls_stride) & (P.m_local - 1)), P.r2fix, fp);     // z(w x): point j + ratio, same class
    const uint64_t E = j << P.x_shift, mask = ((uint64_t)1 << P.lt) - 1;
    const F29 x = f29_mul(load_f29(P.x_lo + (E & mask)), load_f29(P.x_hi + ((E >> P.lt) & mask)), fp);
    F29 acc1 = zc, acc2 = zn;
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const F29 t = f29_add(w5[q], P.gamma_rp);                                    // limbs < 2^30
        const F29 u = f29_add(t, f29_mul(x, P.kbeta_rp[q], fp));                     // w + gamma + k_j x beta
        const F29 v = f29_add(t, f29_mul(ldq(P.sig[q], i), P.beta_fix, fp));         // w + gamma + sigma_j beta
        acc1 = f29_mul(u, acc1, fp);
        acc2 = f29_mul(v, acc2, fp);
    }
    F29 diff = f29_sub2p(acc1, acc2, fp);
    f29_norm(diff);                                                                  // acc1 - acc2, R' form, < 3.4 p

    // ---- 1/Z_H(x) * (gate + alpha * (acc1 - acc2)) + alpha^2/n * (z(x) - 1)/(x - 1)   (:497-503, :372-379).  The constants 1/Z_H
    // (2^261 form: R -> R), alpha/Z_H and alpha^2/n (2^256 form: R' -> R) are per-launch, indexed by the point's coset of H_n.
    F29 one_rp;
#pragma unroll
    for (int l = 0; l < 9; l++) one_rp.l[l] = fp.one[l];
    F29 zm1 = f29_sub2p(zc, one_rp, fp);
    f29_norm(zm1);
    const uint32_t ci = (uint32_t)(j & (P.ratio - 1));
    const F29 l1 = f29_mul(zm1, ldq(P.inv_xm1, i), fp);
    F29 r;
    if (FUSE == 3) {
        r = f29_dot3(gate, P.zh_inv_rp[ci], diff, P.zh_alpha_r[ci], l1, P.a2n_r, fp);          // < 1.2 p
    } else {
        r = f29_add(f29_dot2(gate, P.zh_inv_rp[ci], diff, P.zh_alpha_r[ci], fp), f29_mul(l1, P.a2n_r, fp));
        f29_norm(r);                                                                           // < 2.5 p
        r = f29_canon(r, fp);
    }
    r = f29_canon(r, fp);
    store_fr(P.out + i, f29_to_sat(r));
}

// Variants behind plonk_set_option("quotient_fuse"); the default (6: quotient_evals_kernel_c) is the measured best on every box.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) quotient_evals_kernel(const QuotParams P) { quotient_evals_unlifted(P); }
__global__ void __launch_bounds__(256) quotient_evals_kernel_u3(const QuotParams P) { quotient_evals_unlifted(P); }      // uncapped registers: 3 waves
#define QUOT_LOOP_PTS 8
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) quotient_evals_kernel_loop(const QuotParams P) { quotient_evals_unlifted_loop<QUOT_LOOP_PTS>(P); }
__global__ void __launch_bounds__(256) quotient_evals_kernel_c(const QuotParams P) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.m_local) quotient_point_compact(P, i);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) quotient_evals_kernel_c4(const QuotParams P) {    // the same at four waves per SIMD (128 VGPRs)
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.m_local) quotient_point_compact(P, i);
}
__global__ void __launch_bounds__(256) quotient_gate_kernel(const QuotParams P) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.m_local) quotient_point_gate(P, i);
}
__global__ void __launch_bounds__(256) quotient_perm_kernel(const QuotParams P) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.m_local) quotient_point_perm(P, i);
}
__global__ void __launch_bounds__(256) quotient_evals_kernel_f1(const QuotParams P) { quotient_evals_body<1>(P); }
__global__ void __launch_bounds__(256) quotient_evals_kernel_f2(const QuotParams P) { quotient_evals_body<2>(P); }
__global__ void __launch_bounds__(256) quotient_evals_kernel_f3(const QuotParams P) { quotient_evals_body<3>(P); }

// 1/(x_i - 1) for all m points: Montgomery batch inversion over 16 consecutive points per lane, one Fermat
// inversion per lane.  One-time per domain.
#define QINV_CH 16
__global__ void __launch_bounds__(64) quotient_gen_inv_kernel(Fr* __restrict__ out, uint64_t m, uint32_t cls_stride, uint32_t cls_offset,
                                                              const F29* __restrict__ x_lo, const F29* __restrict__ x_hi, uint32_t lt, uint32_t x_shift,
                                                              const F29Params fp, const F29 pm2_bits /* p - 2 as 29-bit limbs */) {
    const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * QINV_CH;
    if (base >= m) return;
    const uint64_t mask = ((uint64_t)1 << lt) - 1;
    F29 one_rp;
#pragma unroll
    for (int l = 0; l < 9; l++) one_rp.l[l] = fp.one[l];
    F29 d[QINV_CH], pre[QINV_CH];
    F29 run = one_rp;
    for (int k = 0; k < QINV_CH; k++) {
        const uint64_t E = ((uint64_t)cls_offset + (uint64_t)cls_stride * (base + k)) << x_shift;
        const F29 x = f29_mul(load_f29(x_lo + (E & mask)), load_f29(x_hi + ((E >> lt) & mask)), fp);
        F29 t = f29_sub2p(x, one_rp, fp);
        f29_norm(t);
        d[k] = t;
        pre[k] = run;                       // product of d[0..k-1]
        run = f29_mul(t, run, fp);
    }
    // run^(p-2)
    F29 inv = one_rp;
    for (int bit = 9 * 29 - 1; bit >= 0; bit--) {
        inv = f29_mul(inv, inv, fp);
        if ((pm2_bits.l[bit / 29] >> (bit % 29)) & 1) inv = f29_mul(inv, run, fp);
    }
    for (int k = QINV_CH - 1; k >= 0; k--) {
        const F29 r = f29_mul(inv, pre[k], fp);            // 1 / d[k]
        inv = f29_mul(d[k], inv, fp);
        if (base + k < m) store_fr(out + base + k, f29_to_sat(f29_canon(r, fp)));     // a table shorter than one lane's chunk (a coset class of a tiny domain) ends inside it
    }
}

// ---------------------------------------------------------------------------------------------- host
static F29 host_const(const Fr& v_mont, const FrParams& P) { return f29_const_from_mont256(v_mont, P); }

int quotient_evals_run(NttTables& T, const plonk_quotient_inputs* in, size_t n, size_t m, const uint64_t* alpha, const uint64_t* beta,
                       const uint64_t* gamma, const uint64_t* k, uint32_t cls_stride, uint32_t cls_offset, void* d_out, hipStream_t stream) {
    const FrParams& P = T.fp;
    int log_n = 0, log_m = 0;
    while (((size_t)1 << log_n) < n) log_n++;
    while (((size_t)1 << log_m) < m) log_m++;
    if (((size_t)1 << log_n) != n || ((size_t)1 << log_m) != m || m < n || m / n > 8 || m / n < 1)
        return plonk_fail(PLONK_ERR_DOMAIN, "quotient_evals: n = %zu, m = %zu (m/n must be a power of two <= 8)", n, m);
    if (log_m > T.two_adicity) return plonk_fail(PLONK_ERR_DOMAIN, "quotient_evals: 2^%d exceeds the two-adicity", log_m);
    if (cls_stride == 0 || (cls_stride & (cls_stride - 1)) || cls_stride > m / n || cls_offset >= cls_stride)
        return plonk_fail(PLONK_ERR_ARG, "quotient_evals: class %u of %u (the stride must be a power of two dividing m/n = %zu)", cls_offset, cls_stride, m / n);
    const uint64_t m_local = m / cls_stride;
    const uint32_t* g_l = T.curve == PLONK_BN254 ? BN254_FR_GENERATOR_MONT : BLS12_381_FR_GENERATOR_MONT;
    const Fr g_mont = fp_from_limbs<8>(g_l);
    if (!T.quot_x_lo) {          // g * w_Nmax^e, e < 2^lt, constant form
        const size_t cnt = (size_t)1 << T.lt;
        std::vector<F29> h(cnt);
        Fr acc = g_mont;
        for (size_t i = 0; i < cnt; i++) { h[i] = host_const(acc, P); acc = fp_mul(acc, T.h_root[0], P); }
        HIP_TRY(hipMalloc((void**)&T.quot_x_lo, cnt * sizeof(F29)));
        HIP_TRY(hipMemcpyAsync(T.quot_x_lo, h.data(), cnt * sizeof(F29), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    QuotParams q;
    memset(&q, 0, sizeof q);
    q.fp = T.fp29;
    q.m = m;
    q.m_local = m_local;
    q.cls_stride = cls_stride; q.cls_offset = cls_offset;
    q.ratio = (uint32_t)(m / n);
    q.lt = T.lt;
    q.x_shift = T.two_adicity - log_m;
    q.x_lo = T.quot_x_lo;
    q.x_hi = T.tw_hi[0];
    // 1/(x_i - 1) table
    const int inv_key = log_m | (int)(cls_stride << 8) | (int)(cls_offset << 16);
    auto it = T.quot_inv_xm1.find(inv_key);
    if (it == T.quot_inv_xm1.end()) {
        Fr* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, m_local * sizeof(Fr)));
        Fr pm2;                                         // p - 2
        uint64_t br = 2;
        for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)P.p[i] - br; pm2.l[i] = (uint32_t)t; br = (t >> 32) & 1; }
        const uint64_t lanes = (m_local + QINV_CH - 1) / QINV_CH;
        hipLaunchKernelGGL(quotient_gen_inv_kernel, dim3((uint32_t)((lanes + 63) / 64)), dim3(64), 0, stream, d, (uint64_t)m_local, cls_stride, cls_offset, q.x_lo,
                           q.x_hi, q.lt, q.x_shift, q.fp, f29_from_sat(pm2));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { (void)hipFree(d); return plonk_fail(PLONK_ERR_HIP, "quotient_gen_inv launch: %s", hipGetErrorString(e)); }
        T.quot_inv_xm1[inv_key] = d;
        q.inv_xm1 = d;
    } else {
        q.inv_xm1 = it->second;
    }
    // challenge-dependent constants (host; a few dozen field operations)
    const Fr one = fp_one(P);
    const Fr al = fp_from_limbs<8>((const uint32_t*)alpha), be = fp_from_limbs<8>((const uint32_t*)beta), ga = fp_from_limbs<8>((const uint32_t*)gamma);
    Fr r266 = one;                                      // 2^256 mod p (plain residue) doubled ten times = 2^266
    for (int i = 0; i < 10; i++) r266 = fp_add(r266, r266, P);
    q.r2fix = f29_from_sat(r266);
    q.gamma_rp = host_const(ga, P);
    for (int j = 0; j < 5; j++) q.kbeta_rp[j] = host_const(fp_mul(fp_from_limbs<8>((const uint32_t*)k + 8 * j), be, P), P);
    Fr b32 = be;                                        // (32 beta) in Montgomery form -> constant form = beta * 2^266
    for (int i = 0; i < 5; i++) b32 = fp_add(b32, b32, P);
    q.beta_fix = host_const(b32, P);
    Fr nm = fp_zero<8>();
    nm.l[0] = (uint32_t)n; nm.l[1] = (uint32_t)((uint64_t)n >> 32);
    nm = fp_to_mont(nm, P);
    q.a2n_r = f29_from_sat(fp_mul(fp_sqr(al, P), fp_inv(nm, P), P));
    Fr wm = T.h_root[0];                                // w_m = w_Nmax^(2^(s - log m))
    for (int i = 0; i < T.two_adicity - log_m; i++) wm = fp_sqr(wm, P);
    Fr x = g_mont;
    for (uint32_t i = 0; i < q.ratio; i++) {
        const Fr zh = fp_sub(fp_pow_u64(x, (uint64_t)n, P), one, P);
        const Fr zhi = fp_inv(zh, P);
        q.zh_inv_rp[i] = host_const(zhi, P);
        q.zh_alpha_r[i] = f29_from_sat(fp_mul(zhi, al, P));        // Montgomery form of alpha/Z_H = its 2^256 multiple as a plain residue
        x = fp_mul(x, wm, P);
    }
    // constants of the unlifted kernel
    auto pow2_const = [&](int k) {                      // 2^(261+k) mod p as normalised canonical limbs
        Fr t = one;                                     // Montgomery one = 2^256 as a plain residue
        for (int i = 0; i < 5 + k; i++) t = fp_add(t, t, P);
        return f29_from_sat(t);
    };
    q.fix5 = pow2_const(5); q.fix10 = pow2_const(10); q.fix25 = pow2_const(25);
    q.gamma_r = f29_from_sat(ga);
    for (int j = 0; j < 5; j++) q.kbeta_r[j] = f29_from_sat(fp_mul(fp_from_limbs<8>((const uint32_t*)k + 8 * j), be, P));
    q.beta_c = host_const(be, P);
    q.a2n_c = host_const(fp_mul(fp_sqr(al, P), fp_inv(nm, P), P), P);
    q.one_r = f29_from_sat(one);
    {
        Fr x2 = g_mont;
        for (uint32_t i = 0; i < q.ratio; i++) {
            const Fr zh = fp_sub(fp_pow_u64(x2, (uint64_t)n, P), one, P);
            Fr za = fp_mul(fp_inv(zh, P), al, P);       // Montgomery form of alpha/Z_H
            for (int t = 0; t < 25; t++) za = fp_add(za, za, P);
            q.zh_alpha_25[i] = host_const(za, P);       // (alpha/Z_H * 2^25) * 2^261
            x2 = fp_mul(x2, wm, P);
        }
    }
    for (int j = 0; j < 13; j++) q.sel[j] = (const Fr*)in->selectors[j];
    for (int j = 0; j < 5; j++) { q.sig[j] = (const Fr*)in->sigmas[j]; q.wire[j] = (const Fr*)in->wires[j]; }
    q.z = (const Fr*)in->perm;
    q.pi = (const Fr*)in->pub_input;
    q.out = (Fr*)d_out;
    {
        // No aliasing (plonk_hip.h): `out` must not overlap any input vector.  A lane reads z at its own index AND at the shifted index of
        // z(wX) (another lane's output index), and the split form (quotient_fuse = 8) writes `out` in its first kernel before the second reads
        // wires, sigmas and z — an aliased call would return PLONK_OK with a corrupted quotient.
        const char* o0 = (const char*)d_out;
        const char* o1 = o0 + (size_t)m_local * sizeof(Fr);
        auto overlaps = [&](const void* p_) { const char* a = (const char*)p_; return a != nullptr && a < o1 && o0 < a + (size_t)m_local * sizeof(Fr); };
        bool bad = overlaps(q.z) || overlaps(q.pi);
        for (int j = 0; j < 13; j++) bad = bad || overlaps(q.sel[j]);
        for (int j = 0; j < 5; j++) bad = bad || overlaps(q.sig[j]) || overlaps(q.wire[j]);
        if (bad) return plonk_fail(PLONK_ERR_ARG, "quotient_evals: d_out overlaps an input vector (the output must be a buffer of its own)");
    }
    {
        ProfScope ps("quotient_evals_kernel", stream);
        const int fuse = (T.quotient_fuse >= 0 && T.quotient_fuse <= 8) ? T.quotient_fuse : 6;
        const dim3 grid((uint32_t)((m_local + 255) / 256));
        if (fuse == 1) hipLaunchKernelGGL(quotient_evals_kernel_f1, grid, dim3(256), 0, stream, q);
        else if (fuse == 2) hipLaunchKernelGGL(quotient_evals_kernel_f2, grid, dim3(256), 0, stream, q);
        else if (fuse == 3) hipLaunchKernelGGL(quotient_evals_kernel_f3, grid, dim3(256), 0, stream, q);
        else if (fuse == 4) hipLaunchKernelGGL(quotient_evals_kernel_u3, grid, dim3(256), 0, stream, q);
        else if (fuse == 5) hipLaunchKernelGGL(quotient_evals_kernel_loop, dim3((uint32_t)((m_local + 256 * QUOT_LOOP_PTS - 1) / (256 * QUOT_LOOP_PTS))), dim3(256), 0, stream, q);
        else if (fuse == 6) hipLaunchKernelGGL(quotient_evals_kernel_c, grid, dim3(256), 0, stream, q);
        else if (fuse == 7) hipLaunchKernelGGL(quotient_evals_kernel_c4, grid, dim3(256), 0, stream, q);
        else if (fuse == 8) {
            hipLaunchKernelGGL(quotient_gate_kernel, grid, dim3(256), 0, stream, q);
            hipLaunchKernelGGL(quotient_perm_kernel, grid, dim3(256), 0, stream, q);
        }
        else hipLaunchKernelGGL(quotient_evals_kernel, grid, dim3(256), 0, stream, q);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return plonk_fail(PLONK_ERR_HIP, "quotient_evals launch: %s", hipGetErrorString(e));
    return PLONK_OK;
}
