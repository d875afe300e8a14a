// ec_lazy.hpp — XYZZ mixed addition on unsaturated limbs with lazy reduction (hot loop of the MSM
// bucket accumulation).  Same group law and the same exceptional-case handling as ec.hpp; only the
// field representation differs (flimb.hpp).
//
// Invariants of an accumulator (X, Y, ZZ, ZZZ): limbs normalised, values bounded by
//     X < 5.2 p,  Y < 3.3 p,  ZZ < 2 p,  ZZZ < 2 p ;   infinity <=> every limb of ZZ is exactly 0.
// Affine inputs are canonical (< p); infinity is x = y = 0.
// Bound bookkeeping (p/R' <= 2^-7.4):  every product of operands (a, b) is < a*b/R' + p, so
//     P  = U2 + 8p - X1  < 9.2p      R  = S2 + 4p - Y1 < 5.2p
//     PP < 1.5p  PPP < 1.1p  Q < 1.1p  RR < 1.2p
//     X3 = RR + 4p - (PPP + 2Q) < 5.2p          (PPP + 2Q < 3.3p, limbs < 3*2^29 < 2^31)
//     T  = Q + 8p - X3 < 9.1p
//     Y3 = R*T + 2p - Y1*PPP < 3.3p             (each product < 1.3p)
#pragma once
#include "flimb.hpp"

template <int NL, int B> struct AffL { FL<NL, B> x, y; };
template <int NL, int B> struct XyzzL { FL<NL, B> x, y, zz, zzz; };

template <int NL, int B> FP_HD bool affl_is_inf(const AffL<NL, B>& p) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) t |= p.x.l[i] | p.y.l[i];
    return t == 0;
}
template <int NL, int B> FP_HD XyzzL<NL, B> xyzzl_inf() {
    XyzzL<NL, B> r;
    r.x = fl_zero<NL, B>(); r.y = fl_zero<NL, B>(); r.zz = fl_zero<NL, B>(); r.zzz = fl_zero<NL, B>();
    return r;
}

// 2*q for affine q (mdbl-2008-s-1), lazy.  Cold: only reached when a bucket receives the same point twice.
template <int NL, int B>
__device__ __attribute__((noinline)) XyzzL<NL, B> xyzzl_dbl_affine(const AffL<NL, B>& q, const FLParams<NL, B>& P) {
    XyzzL<NL, B> r;
    FL<NL, B> u = fl_add(q.y, q.y);
    fl_norm(u);                                             // 2y < 2p
    const FL<NL, B> v = fl_mul(u, u, P);
    const FL<NL, B> w = fl_mul(u, v, P);
    const FL<NL, B> s = fl_mul(q.x, v, P);
    const FL<NL, B> xx = fl_mul(q.x, q.x, P);
    FL<NL, B> m = fl_add(fl_add(xx, xx), xx);
    fl_norm(m);                                             // 3x^2 < 3.6p
    FL<NL, B> s2 = fl_add(s, s);                            // limbs < 2^30
    FL<NL, B> x3 = fl_sub(fl_mul(m, m, P), s2, P.c4);
    fl_norm(x3);                                            // < 5.2p
    FL<NL, B> t = fl_sub(s, x3, P.c8);
    fl_norm(t);
    FL<NL, B> y3 = fl_sub(fl_mul(m, t, P), fl_mul(w, q.y, P), P.c2);
    fl_norm(y3);
    r.x = x3; r.y = y3; r.zz = v; r.zzz = w;
    return r;
}

// -q for affine q: y -> 2p - y  (value in (p, 2p], which every formula below tolerates)
template <int NL, int B> FP_HD AffL<NL, B> affl_neg(const AffL<NL, B>& q, const FLParams<NL, B>& P) {
    AffL<NL, B> r;
    r.x = q.x;
    r.y = fl_sub(fl_zero<NL, B>(), q.y, P.c2);
    fl_norm(r.y);
    return r;
}

// acc += q without the exceptional cases: returns false (acc untouched) when q has the x of acc
// (P + P or P + (-P)), which the caller hands to the complete out-of-line path.  q must not be infinity.
// FUSED_Y3: Y3 = R*T - Y1*PPP under one Montgomery reduction (fl_dot2) instead of two products and a lazy subtraction.
template <int NL, int B, bool FUSED_Y3 = false>
__device__ __forceinline__ bool xyzzl_madd_fast(XyzzL<NL, B>& a, const AffL<NL, B>& q, const FLParams<NL, B>& P) {
    if (fl_all_zero(a.zz)) {
        a.x = q.x; a.y = q.y;
        a.zz = fl_load_const<NL, B>(P.one); a.zzz = a.zz;
        return true;
    }
    const FL<NL, B> u2 = fl_mul(q.x, a.zz, P);
    const FL<NL, B> s2 = fl_mul(q.y, a.zzz, P);
    FL<NL, B> p = fl_sub(u2, a.x, P.c8);
    fl_norm(p);
    FL<NL, B> r = fl_sub(s2, a.y, P.c4);
    fl_norm(r);
    const FL<NL, B> pp = fl_sqr(p, P);
    if (fl_is_zero_mod_p_lt3p(pp, P)) return false;
    const FL<NL, B> ppp = fl_mul(p, pp, P);
    const FL<NL, B> qq = fl_mul(a.x, pp, P);
    const FL<NL, B> sub = fl_add(ppp, fl_add(qq, qq));       // PPP + 2Q, limbs < 3*2^B
    FL<NL, B> x3 = fl_sub(fl_sqr(r, P), sub, P.c4);
    fl_norm(x3);
    FL<NL, B> t = fl_sub(qq, x3, P.c8);
    fl_norm(t);
    FL<NL, B> y3;
    if (FUSED_Y3) {
        // R*T + (4p - Y1)*PPP  (Y1 < 3.3p; result < (5.2*9.1 + 4*1.1) p * p/R' + p < 1.4p: inside the Y < 3.3p invariant)
        FL<NL, B> ny = fl_sub(fl_zero<NL, B>(), a.y, P.c4);
        fl_norm(ny);
        y3 = fl_dot2(r, t, ny, ppp, P);
    } else {
        y3 = fl_sub(fl_mul(r, t, P), fl_mul(a.y, ppp, P), P.c2);
        fl_norm(y3);
    }
    a.zz = fl_mul(a.zz, pp, P);
    a.zzz = fl_mul(a.zzz, ppp, P);
    a.x = x3;
    a.y = y3;
    return true;
}

// acc + q (madd-2008-s), complete.
template <int NL, int B>
__device__ __forceinline__ XyzzL<NL, B> xyzzl_madd(const XyzzL<NL, B>& a, const AffL<NL, B>& q, const FLParams<NL, B>& P) {
    if (affl_is_inf(q)) return a;
    if (fl_all_zero(a.zz)) {
        XyzzL<NL, B> r;
        r.x = q.x; r.y = q.y;
        r.zz = fl_load_const<NL, B>(P.one); r.zzz = r.zz;
        return r;
    }
    const FL<NL, B> u2 = fl_mul(q.x, a.zz, P);
    const FL<NL, B> s2 = fl_mul(q.y, a.zzz, P);
    FL<NL, B> p = fl_sub(u2, a.x, P.c8);
    fl_norm(p);
    FL<NL, B> r = fl_sub(s2, a.y, P.c4);
    fl_norm(r);
    const FL<NL, B> pp = fl_sqr(p, P);
    if (fl_is_zero_mod_p_lt3p(pp, P)) {                      // same x: P + P or P + (-P)
        const FL<NL, B> rc = fl_canon_small(r, P, 6);
        if (fl_all_zero(rc)) return xyzzl_dbl_affine(q, P);
        return xyzzl_inf<NL, B>();
    }
    XyzzL<NL, B> o;
    const FL<NL, B> ppp = fl_mul(p, pp, P);
    const FL<NL, B> qq = fl_mul(a.x, pp, P);
    FL<NL, B> sub = fl_add(ppp, fl_add(qq, qq));             // PPP + 2Q, limbs < 3*2^B
    o.x = fl_sub(fl_sqr(r, P), sub, P.c4);
    fl_norm(o.x);
    FL<NL, B> t = fl_sub(qq, o.x, P.c8);
    fl_norm(t);
    o.y = fl_sub(fl_mul(r, t, P), fl_mul(a.y, ppp, P), P.c2);
    fl_norm(o.y);
    o.zz = fl_mul(a.zz, pp, P);
    o.zzz = fl_mul(a.zzz, ppp, P);
    return o;
}


// 2*a for an XYZZ accumulator (dbl-2008-s-1), lazy.  Cold (only the P + P branch of xyzzl_add).
template <int NL, int B>
__device__ __attribute__((noinline)) XyzzL<NL, B> xyzzl_dbl(const XyzzL<NL, B>& a, const FLParams<NL, B>& P) {
    if (fl_all_zero(a.zz)) return a;
    XyzzL<NL, B> r;
    FL<NL, B> u = fl_add(a.y, a.y);
    fl_norm(u);                                             // < 6.6p
    const FL<NL, B> v = fl_sqr(u, P);
    const FL<NL, B> w = fl_mul(u, v, P);
    const FL<NL, B> s = fl_mul(a.x, v, P);
    const FL<NL, B> xx = fl_sqr(a.x, P);
    FL<NL, B> m = fl_add(fl_add(xx, xx), xx);
    fl_norm(m);
    const FL<NL, B> s2 = fl_add(s, s);
    FL<NL, B> x3 = fl_sub(fl_sqr(m, P), s2, P.c4);
    fl_norm(x3);
    FL<NL, B> t = fl_sub(s, x3, P.c8);
    fl_norm(t);
    FL<NL, B> y3 = fl_sub(fl_mul(m, t, P), fl_mul(w, a.y, P), P.c2);
    fl_norm(y3);
    r.x = x3; r.y = y3;
    r.zz = fl_mul(v, a.zz, P);
    r.zzz = fl_mul(w, a.zzz, P);
    return r;
}

// a + b for two XYZZ accumulators (add-2008-s), complete, lazy; same invariants in and out as xyzzl_madd.
//   U1, U2, S1, S2 < 1.1p ;  P = U2 + 2p - U1 < 3.1p ;  R = S2 + 2p - S1 < 3.1p ;  rest as in the mixed addition.
template <int NL, int B>
__device__ __forceinline__ XyzzL<NL, B> xyzzl_add(const XyzzL<NL, B>& a, const XyzzL<NL, B>& b, const FLParams<NL, B>& P) {
    if (fl_all_zero(a.zz)) return b;
    if (fl_all_zero(b.zz)) return a;
    const FL<NL, B> u1 = fl_mul(a.x, b.zz, P);
    const FL<NL, B> u2 = fl_mul(b.x, a.zz, P);
    const FL<NL, B> s1 = fl_mul(a.y, b.zzz, P);
    const FL<NL, B> s2 = fl_mul(b.y, a.zzz, P);
    FL<NL, B> p = fl_sub(u2, u1, P.c2);
    fl_norm(p);
    FL<NL, B> r = fl_sub(s2, s1, P.c2);
    fl_norm(r);
    const FL<NL, B> pp = fl_sqr(p, P);
    if (fl_is_zero_mod_p_lt3p(pp, P)) {
        const FL<NL, B> rc = fl_canon_small(r, P, 4);
        if (fl_all_zero(rc)) return xyzzl_dbl(a, P);
        return xyzzl_inf<NL, B>();
    }
    XyzzL<NL, B> o;
    const FL<NL, B> ppp = fl_mul(p, pp, P);
    const FL<NL, B> qq = fl_mul(u1, pp, P);
    const FL<NL, B> sub = fl_add(ppp, fl_add(qq, qq));
    o.x = fl_sub(fl_sqr(r, P), sub, P.c4);
    fl_norm(o.x);
    FL<NL, B> t = fl_sub(qq, o.x, P.c8);
    fl_norm(t);
    o.y = fl_sub(fl_mul(r, t, P), fl_mul(s1, ppp, P), P.c2);
    fl_norm(o.y);
    o.zz = fl_mul(fl_mul(a.zz, b.zz, P), pp, P);
    o.zzz = fl_mul(fl_mul(a.zzz, b.zzz, P), ppp, P);
    return o;
}


// a += b without the exceptional same-x cases: returns false (a untouched) when they occur.
template <int NL, int B>
__device__ __forceinline__ bool xyzzl_add_fast(XyzzL<NL, B>& a, const XyzzL<NL, B>& b, const FLParams<NL, B>& P) {
    if (fl_all_zero(b.zz)) return true;
    if (fl_all_zero(a.zz)) { a = b; return true; }
    const FL<NL, B> u1 = fl_mul(a.x, b.zz, P);
    const FL<NL, B> u2 = fl_mul(b.x, a.zz, P);
    const FL<NL, B> s1 = fl_mul(a.y, b.zzz, P);
    const FL<NL, B> s2 = fl_mul(b.y, a.zzz, P);
    FL<NL, B> p = fl_sub(u2, u1, P.c2);
    fl_norm(p);
    FL<NL, B> r = fl_sub(s2, s1, P.c2);
    fl_norm(r);
    const FL<NL, B> pp = fl_sqr(p, P);
    if (fl_is_zero_mod_p_lt3p(pp, P)) return false;
    const FL<NL, B> ppp = fl_mul(p, pp, P);
    const FL<NL, B> qq = fl_mul(u1, pp, P);
    const FL<NL, B> sub = fl_add(ppp, fl_add(qq, qq));
    FL<NL, B> x3 = fl_sub(fl_sqr(r, P), sub, P.c4);
    fl_norm(x3);
    FL<NL, B> t = fl_sub(qq, x3, P.c8);
    fl_norm(t);
    FL<NL, B> y3 = fl_sub(fl_mul(r, t, P), fl_mul(s1, ppp, P), P.c2);
    fl_norm(y3);
    a.zz = fl_mul(fl_mul(a.zz, b.zz, P), pp, P);
    a.zzz = fl_mul(fl_mul(a.zzz, b.zzz, P), ppp, P);
    a.x = x3;
    a.y = y3;
    return true;
}
