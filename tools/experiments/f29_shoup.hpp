// f29_shoup.hpp — EXPERIMENT (not part of libplonk_hip.so, not hashed into the profiles): a precomputed-quotient ("Shoup") constant
// multiplier on the 9 x 29-bit limbs of fp29.hpp, as a candidate for the NTT's data x twiddle products.
//
// Today (fp29.hpp: f29_mul): Montgomery product scanning, 81 limb products + 81 reduction products + 9 v_mul_lo = 171 multiplier
// instructions, 213 VALU instructions per product.  Every NTT product is data x PRECOMPUTED constant, so the quotient can be
// precomputed too:
//     c  < p                      the constant as a plain residue (no Montgomery factor: x keeps whatever factor it carries)
//     cq = floor(c * 2^261 / p)   its quotient constant, < 2^261
//     q  = floor(x * cq / 2^261)  -> floor(x*c/p) - 1 <= q <= floor(x*c/p)        (x < 2^261)
//     r  = x*c - q*p  in [0, 2p)  and only its low 261 bits are needed: r = (x*c + q*pbar) mod 2^261, pbar = 2^261 - p
// Cost: the HIGH columns 7..16 of x*cq (53 limb products; columns 0..6 only matter through a carry below 2^-20 of a unit: q may come
// out one smaller, r < 3p) + the LOW columns 0..8 of x*c and of q*pbar (45 + 45) = 143 limb products, no v_mul_lo, against 171.
// Needs a second 9-limb constant per twiddle (72 B instead of 36 B): the in-LDS twiddle table of the pass kernel would no longer fit
// beside two 72 KiB tiles, so a kernel using this reads its twiddles through L1.
#pragma once
#include "../../distributed_plonk_amd/csrc/fp29.hpp"

struct F29Shoup {
    uint32_t pbar[9];    // 2^261 - p, normalised 29-bit limbs
};

// x: limbs < 2^31, value < 2^259.4 (the same contract as f29_mul's data operand); c, cq: normalised limbs.
// Result: normalised limbs, value = x*c - q*p with q in {Q - 2, Q - 1, Q}, Q = floor(x*c/p):  0 <= value < 3p.
FP_HD F29 f29_mul_shoup(const F29& x, const F29& c, const F29& cq, const F29Shoup& S) {
    uint64_t acc = 0;
    uint32_t q[9];
    // ---- q = floor(x * cq / 2^261): columns 7 .. 16 (the carry out of columns 7 and 8 is all the lower half contributes)
#pragma unroll
    for (int k = 7; k < 17; k++) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) { acc += (uint64_t)x.l[i] * cq.l[k - i]; F29_CHAIN(acc); }
        if (k >= 9) q[k - 9] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
        F29_CHAIN(acc);
    }
    q[8] = (uint32_t)acc;
    // ---- r = (x*c + q*pbar) mod 2^261: columns 0 .. 8
    F29 r;
    acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)x.l[i] * c.l[k - i]; F29_CHAIN(acc); }
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)q[i] * S.pbar[k - i]; F29_CHAIN(acc); }
        r.l[k] = (uint32_t)acc & F29_MASK;
        acc >>= 29;
        F29_CHAIN(acc);
    }
    return r;
}

// ---- host-side construction (plain big-integer arithmetic on 32-bit words; table building only)
#if !defined(__HIP_DEVICE_COMPILE__)
#include <cstddef>
#include <vector>
using std::size_t;
namespace f29_shoup_host {
typedef std::vector<uint32_t> Big;          // little-endian 32-bit words
inline void trim(Big& a) { while (a.size() > 1 && a.back() == 0) a.pop_back(); }
inline int cmp(const Big& a, const Big& b) {
    if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
    for (size_t i = a.size(); i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
inline Big sub(const Big& a, const Big& b) {            // a >= b
    Big r(a.size());
    uint64_t br = 0;
    for (size_t i = 0; i < a.size(); i++) {
        const uint64_t t = (uint64_t)a[i] - (i < b.size() ? b[i] : 0) - br;
        r[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    trim(r);
    return r;
}
inline Big shl1(const Big& a) {
    Big r(a.size() + 1);
    uint32_t c = 0;
    for (size_t i = 0; i < a.size(); i++) { r[i] = (a[i] << 1) | c; c = a[i] >> 31; }
    r[a.size()] = c;
    trim(r);
    return r;
}
// floor(c * 2^261 / p) by binary long division (c < p)
inline F29 quotient_const(const Fp<8>& c, const FpParams<8>& P) {
    Big rem(c.l, c.l + 8), p(P.p, P.p + 8);
    trim(rem); trim(p);
    Big quo(9, 0);
    for (int b = 260; b >= 0; b--) {
        rem = shl1(rem);
        if (cmp(rem, p) >= 0) { rem = sub(rem, p); quo[b >> 5] |= 1u << (b & 31); }
    }
    F29 r;
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, w = bit >> 5, off = bit & 31;
        uint64_t v = quo[w] >> off;
        if (off + 29 > 32 && w + 1 < 9) v |= (uint64_t)quo[w + 1] << (32 - off);
        r.l[k] = (uint32_t)v & F29_MASK;
    }
    return r;
}
inline F29Shoup make_params(const FpParams<8>& P) {
    Big two261(9, 0), p(P.p, P.p + 8);
    two261[8] = 1u << 5;                                // 2^261 = 2^(8*32 + 5)
    trim(p);
    Big d = sub(two261, p);
    d.resize(9, 0);
    F29Shoup s;
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, w = bit >> 5, off = bit & 31;
        uint64_t v = d[w] >> off;
        if (off + 29 > 32 && w + 1 < 9) v |= (uint64_t)d[w + 1] << (32 - off);
        s.pbar[k] = (uint32_t)v & F29_MASK;
    }
    return s;
}
}  // namespace f29_shoup_host
#endif
