// Host wrapper for tools/experiments/check_f29_shoup.py (g++, no GPU): exposes the experimental multiplier and its constant construction.
#include "f29_shoup.hpp"
#include "../../distributed_plonk_amd/csrc/constants.h"

static const FpParams<8>& params(int curve) { return curve == 0 ? BN254_FR_PARAMS : BLS12_381_FR_PARAMS; }

extern "C" {
// c8: canonical residue (8 x u32) -> c (9 limbs) and cq (9 limbs)
void shoup_const(int curve, const uint32_t* c8, uint32_t* c29, uint32_t* cq29) {
    Fp<8> c;
    for (int i = 0; i < 8; i++) c.l[i] = c8[i];
    const F29 a = f29_from_sat(c), b = f29_shoup_host::quotient_const(c, params(curve));
    for (int i = 0; i < 9; i++) { c29[i] = a.l[i]; cq29[i] = b.l[i]; }
}
// n products: x (n x 9 limbs, lazy), c / cq (n x 9) -> r (n x 9)
void shoup_mul(int curve, const uint32_t* x, const uint32_t* c, const uint32_t* cq, uint32_t* r, long n) {
    const F29Shoup S = f29_shoup_host::make_params(params(curve));
    for (long k = 0; k < n; k++) {
        F29 a, b, q;
        for (int i = 0; i < 9; i++) { a.l[i] = x[9 * k + i]; b.l[i] = c[9 * k + i]; q.l[i] = cq[9 * k + i]; }
        const F29 o = f29_mul_shoup(a, b, q, S);
        for (int i = 0; i < 9; i++) r[9 * k + i] = o.l[i];
    }
}
// the product path's Montgomery multiplier on the same operands, for reference: x * (c * 2^261 mod p) / 2^261
void mont_mul(int curve, const uint32_t* x, const uint32_t* cm, uint32_t* r, long n) {
    const F29Params P = f29_make_params(params(curve));
    for (long k = 0; k < n; k++) {
        F29 a, b;
        for (int i = 0; i < 9; i++) { a.l[i] = x[9 * k + i]; b.l[i] = cm[9 * k + i]; }
        const F29 o = f29_mul(a, b, P);
        for (int i = 0; i < 9; i++) r[9 * k + i] = o.l[i];
    }
}
}
