// Host build (g++, no GPU) of the MSM's device arithmetic for tests/test_ec_lazy_host.py: the unsaturated-limb Montgomery field
// (csrc/flimb.hpp) and the lazy XYZZ curve formulas of the bucket accumulation and the reduction pyramid (csrc/ec_lazy.hpp) are plain
// C++ apart from their __device__ markers, so the very code the kernels run is checked against Python integers without a GPU.
#define __device__
#define __forceinline__ inline
#include "../../distributed_plonk_amd/csrc/ec_lazy.hpp"
#include "../../distributed_plonk_amd/csrc/constants.h"

namespace {
template <int NL, int B> FL<NL, B> ld(const uint32_t* s) {
    FL<NL, B> r;
    for (int i = 0; i < NL; i++) r.l[i] = s[i];
    return r;
}
template <int NL, int B> void st(uint32_t* d, const FL<NL, B>& a) {
    for (int i = 0; i < NL; i++) d[i] = a.l[i];
}
template <int NL, int B> XyzzL<NL, B> ld4(const uint32_t* s) {
    XyzzL<NL, B> r;
    r.x = ld<NL, B>(s); r.y = ld<NL, B>(s + NL); r.zz = ld<NL, B>(s + 2 * NL); r.zzz = ld<NL, B>(s + 3 * NL);
    return r;
}
template <int NL, int B> void st4(uint32_t* d, const XyzzL<NL, B>& a) {
    st(d, a.x); st(d + NL, a.y); st(d + 2 * NL, a.zz); st(d + 3 * NL, a.zzz);
}
template <int NL, int B> AffL<NL, B> ld2(const uint32_t* s) {
    AffL<NL, B> r;
    r.x = ld<NL, B>(s); r.y = ld<NL, B>(s + NL);
    return r;
}

// the same construction msm_engine.hip uses (fl_params<NQ>)
const FLParams<9, 29>& bn() {
    static const FLParams<9, 29> P = fl_make_params<9, 29, 8>(BN254_FQ_PARAMS);
    return P;
}
const FLParams<14, 28>& bls() {
    static const FLParams<14, 28> P = fl_make_params<14, 28, 12>(BLS12_381_FQ_PARAMS);
    return P;
}

template <int NL, int B> void dump(const FLParams<NL, B>& P, uint32_t* out) {
    const uint32_t* rows[8] = {P.p, P.p2, P.c2, P.c4, P.c8, P.one, P.r_std, P.r2fix};
    for (int r = 0; r < 8; r++)
        for (int i = 0; i < NL; i++) out[r * NL + i] = rows[r][i];
    out[8 * NL] = P.inv;
}

// op: 0 mul(a, b), 1 sqr(a), 2 dot2(a, b, c, d)
template <int NL, int B> void field_op(const FLParams<NL, B>& P, int op, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d,
                                       uint32_t* r, long n) {
    for (long k = 0; k < n; k++) {
        const FL<NL, B> x = ld<NL, B>(a + k * NL);
        FL<NL, B> o;
        if (op == 0) o = fl_mul(x, ld<NL, B>(b + k * NL), P);
        else if (op == 1) o = fl_sqr(x, P);
        else o = fl_dot2(x, ld<NL, B>(b + k * NL), ld<NL, B>(c + k * NL), ld<NL, B>(d + k * NL), P);
        st(r + k * NL, o);
    }
}

// op: 0 madd_fast (plain Y3), 1 madd_fast (fused Y3), 2 madd (complete), 3 add (complete), 4 add_fast, 5 dbl, 6 dbl_affine, 7 neg (affine)
// a: accumulator (4 NL limbs), b: affine (2 NL) or accumulator (4 NL) operand; out: 4 NL limbs (neg: 2 NL).  Returns the fast paths' flag (1 otherwise).
template <int NL, int B> int curve_op(const FLParams<NL, B>& P, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    XyzzL<NL, B> acc = ld4<NL, B>(a);
    bool ok = true;
    switch (op) {
    case 0: ok = xyzzl_madd_fast<NL, B, false>(acc, ld2<NL, B>(b), P); break;
    case 1: ok = xyzzl_madd_fast<NL, B, true>(acc, ld2<NL, B>(b), P); break;
    case 2: acc = xyzzl_madd(acc, ld2<NL, B>(b), P); break;
    case 3: acc = xyzzl_add(acc, ld4<NL, B>(b), P); break;
    case 4: ok = xyzzl_add_fast(acc, ld4<NL, B>(b), P); break;
    case 5: acc = xyzzl_dbl(acc, P); break;
    case 6: acc = xyzzl_dbl_affine(ld2<NL, B>(b), P); break;
    case 7: {
        const AffL<NL, B> q = affl_neg(ld2<NL, B>(b), P);
        st(out, q.x); st(out + NL, q.y);
        return 1;
    }
    default: return -1;
    }
    st4(out, acc);
    return ok ? 1 : 0;
}

// saturated Montgomery (R = 2^(32N), the reference's form) <-> limb form (R' = 2^(B NL)): what bases_to_limbs_kernel / store_std do
template <int NL, int B, int N> void from_std(const FLParams<NL, B>& P, const uint32_t* s, uint32_t* out) {
    Fp<N> a;
    for (int i = 0; i < N; i++) a.l[i] = s[i];
    st(out, fl_canon_lt2p(fl_mul(fl_from_sat<NL, B, N>(a), ld<NL, B>(P.r2fix), P), P));
}
template <int NL, int B, int N> void to_std(const FLParams<NL, B>& P, const uint32_t* s, uint32_t* out) {
    const FL<NL, B> v = fl_canon_lt2p(fl_mul(ld<NL, B>(s), ld<NL, B>(P.r_std), P), P);
    const Fp<N> o = fl_to_sat<NL, B, N>(v);
    for (int i = 0; i < N; i++) out[i] = o.l[i];
}
}  // namespace

extern "C" {
void ecl_params(int curve, uint32_t* out) { curve == 0 ? dump(bn(), out) : dump(bls(), out); }
void ecl_field_op(int curve, int op, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* r, long n) {
    curve == 0 ? field_op(bn(), op, a, b, c, d, r, n) : field_op(bls(), op, a, b, c, d, r, n);
}
int ecl_curve_op(int curve, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    return curve == 0 ? curve_op(bn(), op, a, b, out) : curve_op(bls(), op, a, b, out);
}
void ecl_from_std(int curve, const uint32_t* s, uint32_t* out) {
    curve == 0 ? from_std<9, 29, 8>(bn(), s, out) : from_std<14, 28, 12>(bls(), s, out);
}
void ecl_to_std(int curve, const uint32_t* s, uint32_t* out) {
    curve == 0 ? to_std<9, 29, 8>(bn(), s, out) : to_std<14, 28, 12>(bls(), s, out);
}
}
