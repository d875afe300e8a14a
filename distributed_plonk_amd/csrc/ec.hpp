// ec.hpp — G1 (y^2 = x^3 + b, a = 0) point arithmetic in extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), host + device.
//
// Contract to match: only the GROUP ELEMENT of ark-ec 0.3.0's VariableBaseMSM::multi_scalar_mul
// result is observable (reference worker.rs:179-182 returns a Jacobian triple, the dispatcher adds
// triples and normalises: dispatcher.rs:236-240, dispatcher2.rs:887-892), so bucket order and the
// projective representation are free.  XYZZ mixed addition is 8M+2S against 7M+4S for Jacobian and
// needs no field inversion.  The exceptional cases the reference's tests provoke are handled
// exactly: infinity bases (dispatcher2.rs:1101), duplicated bases => P+P (dispatcher.rs:194-196),
// P + (-P).
#pragma once
#include "fp.hpp"

template <int N> struct AffPt { Fp<N> x, y; };              // infinity: x == y == 0 (not on the curve, b != 0)
template <int N> struct XyzzPt { Fp<N> x, y, zz, zzz; };    // infinity: zz == 0

template <int N> FP_HD bool aff_is_inf(const AffPt<N>& p) { return fp_is_zero(p.x) && fp_is_zero(p.y); }
template <int N> FP_HD bool xyzz_is_inf(const XyzzPt<N>& p) { return fp_is_zero(p.zz); }
template <int N> FP_HD XyzzPt<N> xyzz_inf() {
    XyzzPt<N> r;
    r.x = fp_zero<N>(); r.y = fp_zero<N>(); r.zz = fp_zero<N>(); r.zzz = fp_zero<N>();
    return r;
}
template <int N> FP_HD XyzzPt<N> xyzz_from_affine(const AffPt<N>& p, const FpParams<N>& P) {
    XyzzPt<N> r;
    if (aff_is_inf(p)) return xyzz_inf<N>();
    r.x = p.x; r.y = p.y; r.zz = fp_one(P); r.zzz = fp_one(P);
    return r;
}

#if defined(__HIPCC__)
#define EC_COLD __host__ __device__ __attribute__((noinline))
#else
#define EC_COLD __attribute__((noinline))
#endif
// Out-of-line copies for the rare branches (P+P) and for kernels off the hot loop: keeps the hot
// kernels' code small and the build fast.
template <int N> EC_COLD XyzzPt<N> xyzz_dbl_affine_cold(const AffPt<N>& p, const FpParams<N>& P);
template <int N> EC_COLD XyzzPt<N> xyzz_dbl_cold(const XyzzPt<N>& p, const FpParams<N>& P);
template <int N> EC_COLD XyzzPt<N> xyzz_add_cold(const XyzzPt<N>& a, const XyzzPt<N>& b, const FpParams<N>& P);
template <int N> EC_COLD XyzzPt<N> xyzz_madd_cold(const XyzzPt<N>& a, const AffPt<N>& q, const FpParams<N>& P);
template <int N> EC_COLD Fp<N> fp_inv_cold(const Fp<N>& a, const FpParams<N>& P);

// 2*P for affine P (mdbl-2008-s-1)
template <int N> FP_HD XyzzPt<N> xyzz_dbl_affine(const AffPt<N>& p, const FpParams<N>& P) {
    XyzzPt<N> r;
    Fp<N> u = fp_dbl(p.y, P);
    Fp<N> v = fp_sqr(u, P);
    Fp<N> w = fp_mul(u, v, P);
    Fp<N> s = fp_mul(p.x, v, P);
    Fp<N> xx = fp_sqr(p.x, P);
    Fp<N> m = fp_add(fp_dbl(xx, P), xx, P);
    r.x = fp_sub(fp_sub(fp_sqr(m, P), s, P), s, P);
    r.y = fp_sub(fp_mul(m, fp_sub(s, r.x, P), P), fp_mul(w, p.y, P), P);
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P (dbl-2008-s-1)
template <int N> FP_HD XyzzPt<N> xyzz_dbl(const XyzzPt<N>& p, const FpParams<N>& P) {
    if (xyzz_is_inf(p)) return p;
    XyzzPt<N> r;
    Fp<N> u = fp_dbl(p.y, P);
    Fp<N> v = fp_sqr(u, P);
    Fp<N> w = fp_mul(u, v, P);
    Fp<N> s = fp_mul(p.x, v, P);
    Fp<N> xx = fp_sqr(p.x, P);
    Fp<N> m = fp_add(fp_dbl(xx, P), xx, P);
    r.x = fp_sub(fp_sub(fp_sqr(m, P), s, P), s, P);
    r.y = fp_sub(fp_mul(m, fp_sub(s, r.x, P), P), fp_mul(w, p.y, P), P);
    r.zz = fp_mul(v, p.zz, P);
    r.zzz = fp_mul(w, p.zzz, P);
    return r;
}

// acc + q, q affine (madd-2008-s), complete
template <int N> FP_HD XyzzPt<N> xyzz_madd(const XyzzPt<N>& a, const AffPt<N>& q, const FpParams<N>& P) {
    if (aff_is_inf(q)) return a;
    if (xyzz_is_inf(a)) return xyzz_from_affine(q, P);
    Fp<N> u2 = fp_mul(q.x, a.zz, P);
    Fp<N> s2 = fp_mul(q.y, a.zzz, P);
    Fp<N> p = fp_sub(u2, a.x, P);
    Fp<N> r = fp_sub(s2, a.y, P);
    if (fp_is_zero(p)) {
        if (fp_is_zero(r)) return xyzz_dbl_affine_cold(q, P);
        return xyzz_inf<N>();
    }
    XyzzPt<N> o;
    Fp<N> pp = fp_sqr(p, P);
    Fp<N> ppp = fp_mul(p, pp, P);
    Fp<N> qq = fp_mul(a.x, pp, P);
    o.x = fp_sub(fp_sub(fp_sub(fp_sqr(r, P), ppp, P), qq, P), qq, P);
    o.y = fp_sub(fp_mul(r, fp_sub(qq, o.x, P), P), fp_mul(a.y, ppp, P), P);
    o.zz = fp_mul(a.zz, pp, P);
    o.zzz = fp_mul(a.zzz, ppp, P);
    return o;
}

// a + b (add-2008-s), complete
template <int N> FP_HD XyzzPt<N> xyzz_add(const XyzzPt<N>& a, const XyzzPt<N>& b, const FpParams<N>& P) {
    if (xyzz_is_inf(a)) return b;
    if (xyzz_is_inf(b)) return a;
    Fp<N> u1 = fp_mul(a.x, b.zz, P);
    Fp<N> u2 = fp_mul(b.x, a.zz, P);
    Fp<N> s1 = fp_mul(a.y, b.zzz, P);
    Fp<N> s2 = fp_mul(b.y, a.zzz, P);
    Fp<N> p = fp_sub(u2, u1, P);
    Fp<N> r = fp_sub(s2, s1, P);
    if (fp_is_zero(p)) {
        if (fp_is_zero(r)) return xyzz_dbl_cold(a, P);
        return xyzz_inf<N>();
    }
    XyzzPt<N> o;
    Fp<N> pp = fp_sqr(p, P);
    Fp<N> ppp = fp_mul(p, pp, P);
    Fp<N> qq = fp_mul(u1, pp, P);
    o.x = fp_sub(fp_sub(fp_sub(fp_sqr(r, P), ppp, P), qq, P), qq, P);
    o.y = fp_sub(fp_mul(r, fp_sub(qq, o.x, P), P), fp_mul(s1, ppp, P), P);
    o.zz = fp_mul(fp_mul(a.zz, b.zz, P), pp, P);
    o.zzz = fp_mul(fp_mul(a.zzz, b.zzz, P), ppp, P);
    return o;
}

// k * p for a small non-negative integer k (double-and-add, MSB first)
template <int N> FP_HD XyzzPt<N> xyzz_mul_small(const XyzzPt<N>& p, uint64_t k, const FpParams<N>& P) {
    XyzzPt<N> acc = xyzz_inf<N>();
    for (int i = 63; i >= 0; i--) {
        acc = xyzz_dbl_cold(acc, P);
        if ((k >> i) & 1) acc = xyzz_add_cold(acc, p, P);
    }
    return acc;
}

// affine normalisation (one inversion)
template <int N> FP_HD AffPt<N> xyzz_to_affine(const XyzzPt<N>& p, const FpParams<N>& P) {
    AffPt<N> r;
    if (xyzz_is_inf(p)) { r.x = fp_zero<N>(); r.y = fp_zero<N>(); return r; }
    Fp<N> zi = fp_inv_cold(p.zzz, P);           // 1/ZZZ
    Fp<N> zz_inv = fp_sqr(fp_mul(zi, p.zz, P), P);   // (ZZ/ZZZ)^2 = Z^-2 = 1/ZZ
    r.x = fp_mul(p.x, zz_inv, P);
    r.y = fp_mul(p.y, zi, P);
    return r;
}

// Jacobian (X,Y,Z), as the reference returns from varMsm (worker.rs:179-182): conversions
template <int N> struct JacPt { Fp<N> x, y, z; };
template <int N> FP_HD XyzzPt<N> xyzz_from_jac(const JacPt<N>& j, const FpParams<N>& P) {
    XyzzPt<N> r;
    if (fp_is_zero(j.z)) return xyzz_inf<N>();
    r.x = j.x; r.y = j.y;
    r.zz = fp_sqr(j.z, P);
    r.zzz = fp_mul(r.zz, j.z, P);
    return r;
}
// normalised Jacobian: (x, y, 1) or arkworks' zero (1, 1, 0)
template <int N> FP_HD JacPt<N> jac_from_xyzz_normalised(const XyzzPt<N>& p, const FpParams<N>& P) {
    JacPt<N> r;
    if (xyzz_is_inf(p)) { r.x = fp_one(P); r.y = fp_one(P); r.z = fp_zero<N>(); return r; }
    AffPt<N> a = xyzz_to_affine(p, P);
    r.x = a.x; r.y = a.y; r.z = fp_one(P);
    return r;
}

template <int N> EC_COLD XyzzPt<N> xyzz_dbl_affine_cold(const AffPt<N>& p, const FpParams<N>& P) { return xyzz_dbl_affine(p, P); }
template <int N> EC_COLD XyzzPt<N> xyzz_dbl_cold(const XyzzPt<N>& p, const FpParams<N>& P) { return xyzz_dbl(p, P); }
template <int N> EC_COLD XyzzPt<N> xyzz_add_cold(const XyzzPt<N>& a, const XyzzPt<N>& b, const FpParams<N>& P) { return xyzz_add(a, b, P); }
template <int N> EC_COLD XyzzPt<N> xyzz_madd_cold(const XyzzPt<N>& a, const AffPt<N>& q, const FpParams<N>& P) { return xyzz_madd(a, q, P); }
template <int N> EC_COLD Fp<N> fp_inv_cold(const Fp<N>& a, const FpParams<N>& P) { return fp_inv(a, P); }
