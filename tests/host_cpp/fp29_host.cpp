// Host build (g++, no GPU) of the device arithmetic headers for tests/test_fp29_host.py: the FP_HD functions of fp29.hpp compile for
// the host too, so the limb arithmetic the NTT kernels run is checked against Python integers without a GPU.
#include "../../distributed_plonk_amd/csrc/fp29.hpp"
#include "../../distributed_plonk_amd/csrc/constants.h"

static const FpParams<8>& params(int curve) { return curve == 0 ? BN254_FR_PARAMS : BLS12_381_FR_PARAMS; }

extern "C" {
// c_mont: the constant in the reference's Montgomery form (8 x u32) -> prepared constant: c (9 limbs), cq (9 limbs)
void shoup_const(int curve, const uint32_t* c_mont, uint32_t* c29, uint32_t* cq29) {
    Fp<8> c;
    for (int i = 0; i < 8; i++) c.l[i] = c_mont[i];
    const F29S s = f29_shoup_from_mont256(c, params(curve));
    for (int i = 0; i < 9; i++) { c29[i] = s.c[i]; cq29[i] = s.cq[i]; }
}
void get_pbar(int curve, uint32_t* out) {
    const F29Params P = f29_make_params(params(curve));
    for (int i = 0; i < 9; i++) out[i] = P.pbar[i];
}
// n products: x (n x 9 limbs, lazy), c / cq (n x 9) -> r (n x 9)
void shoup_mul(int curve, const uint32_t* x, const uint32_t* c, const uint32_t* cq, uint32_t* r, long n) {
    const F29Params P = f29_make_params(params(curve));
    for (long k = 0; k < n; k++) {
        F29 a;
        for (int i = 0; i < 9; i++) a.l[i] = x[9 * k + i];
        const F29 o = f29_mul_shoup(a, c + 9 * k, cq + 9 * k, P);
        for (int i = 0; i < 9; i++) r[9 * k + i] = o.l[i];
    }
}
// the Montgomery multiplier on the same operands: x * (c * 2^261 mod p) / 2^261
void mont_mul(int curve, const uint32_t* x, const uint32_t* cm, uint32_t* r, long n) {
    const F29Params P = f29_make_params(params(curve));
    for (long k = 0; k < n; k++) {
        F29 a, b;
        for (int i = 0; i < 9; i++) { a.l[i] = x[9 * k + i]; b.l[i] = cm[9 * k + i]; }
        const F29 o = f29_mul(a, b, P);
        for (int i = 0; i < 9; i++) r[9 * k + i] = o.l[i];
    }
}
// normalised values below 48p -> canonical
void canon_lazy(int curve, const uint32_t* x, uint32_t* r, long n) {
    const F29Params P = f29_make_params(params(curve));
    for (long k = 0; k < n; k++) {
        F29 a;
        for (int i = 0; i < 9; i++) a.l[i] = x[9 * k + i];
        const F29 o = f29_canon_lazy(a, P);
        for (int i = 0; i < 9; i++) r[9 * k + i] = o.l[i];
    }
}
}
