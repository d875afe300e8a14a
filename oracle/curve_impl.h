/* oracle/curve_impl.h — TEST INFRASTRUCTURE ONLY (CPU oracle), not part of the product path.
 *
 * PARITY UNPINNED: see oracle/plonk_oracle.c header.
 *
 * Short-Weierstrass (a = 0) G1 arithmetic in Jacobian coordinates and the windowed Pippenger MSM,
 * restating the published algorithms of ark-ec 0.3.0 (reference dependency, Cargo.lock:99-102):
 *   models/short_weierstrass_jacobian.rs  add_assign_mixed (madd-2007-bl), add_assign (add-2007-bl),
 *                                         double_in_place (dbl-2009-l), From<Projective> for Affine
 *   msm/variable_base.rs                  VariableBaseMSM::multi_scalar_mul       
 * Reference call sites: /root/reference/src/worker.rs:122,179-182; dispatcher.rs:236-240,1052;
 * dispatcher2.rs:887-892.
 *
 * Include with QS = suffix of the base-field instantiation (4 or 6).
 */
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define QN(name) CAT(name, QS)
#define QFE QN(fe)
#define QCTX QN(fctx)
#define AFF QN(aff)
#define JAC QN(jac)

typedef struct { QFE x, y; int inf; } AFF;
typedef struct { QFE x, y, z; } JAC;

static inline int QN(jac_is_zero)(const JAC *p) { return QN(fe_is_zero)(&p->z); }
static inline void QN(jac_set_zero)(const QCTX *F, JAC *p) { p->x = F->one; p->y = F->one; memset(&p->z, 0, sizeof p->z); }

static void QN(jac_double)(const QCTX *F, JAC *p) {
    if (QN(jac_is_zero)(p)) return;
    QFE a, b, c, d, e, f, t;
    QN(fe_sqr)(F, &a, &p->x);
    QN(fe_sqr)(F, &b, &p->y);
    QN(fe_sqr)(F, &c, &b);
    QN(fe_add)(F, &t, &p->x, &b); QN(fe_sqr)(F, &t, &t);
    QN(fe_sub)(F, &t, &t, &a); QN(fe_sub)(F, &t, &t, &c); QN(fe_dbl)(F, &d, &t);
    QN(fe_dbl)(F, &e, &a); QN(fe_add)(F, &e, &e, &a);
    QN(fe_sqr)(F, &f, &e);
    QN(fe_mul)(F, &t, &p->y, &p->z); QN(fe_dbl)(F, &p->z, &t);          /* z3 = 2 y z */
    QN(fe_sub)(F, &t, &f, &d); QN(fe_sub)(F, &p->x, &t, &d);             /* x3 = f - 2d */
    QN(fe_sub)(F, &t, &d, &p->x); QN(fe_mul)(F, &t, &e, &t);
    QN(fe_dbl)(F, &c, &c); QN(fe_dbl)(F, &c, &c); QN(fe_dbl)(F, &c, &c); /* 8c */
    QN(fe_sub)(F, &p->y, &t, &c);
}

static void QN(jac_add_mixed)(const QCTX *F, JAC *p, const AFF *q) {
    if (q->inf) return;
    if (QN(jac_is_zero)(p)) { p->x = q->x; p->y = q->y; p->z = F->one; return; }
    QFE z1z1, u2, s2, h, hh, i, j, r, v, t;
    QN(fe_sqr)(F, &z1z1, &p->z);
    QN(fe_mul)(F, &u2, &q->x, &z1z1);
    QN(fe_mul)(F, &s2, &q->y, &p->z); QN(fe_mul)(F, &s2, &s2, &z1z1);
    if (QN(fe_eq)(&p->x, &u2) && QN(fe_eq)(&p->y, &s2)) { QN(jac_double)(F, p); return; }
    QN(fe_sub)(F, &h, &u2, &p->x);
    QN(fe_sqr)(F, &hh, &h);
    QN(fe_dbl)(F, &i, &hh); QN(fe_dbl)(F, &i, &i);
    QN(fe_mul)(F, &j, &h, &i);
    QN(fe_sub)(F, &r, &s2, &p->y); QN(fe_dbl)(F, &r, &r);
    QN(fe_mul)(F, &v, &p->x, &i);
    /* z3 = (z+h)^2 - z1z1 - hh */
    QN(fe_add)(F, &t, &p->z, &h); QN(fe_sqr)(F, &t, &t);
    QN(fe_sub)(F, &t, &t, &z1z1); QN(fe_sub)(F, &p->z, &t, &hh);
    /* x3 = r^2 - j - 2v */
    QN(fe_sqr)(F, &t, &r); QN(fe_sub)(F, &t, &t, &j); QN(fe_sub)(F, &t, &t, &v); QN(fe_sub)(F, &p->x, &t, &v);
    /* y3 = r (v - x3) - 2 y j */
    QN(fe_sub)(F, &t, &v, &p->x); QN(fe_mul)(F, &t, &r, &t);
    QN(fe_mul)(F, &j, &p->y, &j); QN(fe_dbl)(F, &j, &j);
    QN(fe_sub)(F, &p->y, &t, &j);
}

static void QN(jac_add)(const QCTX *F, JAC *p, const JAC *q) {
    if (QN(jac_is_zero)(p)) { *p = *q; return; }
    if (QN(jac_is_zero)(q)) return;
    QFE z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t;
    QN(fe_sqr)(F, &z1z1, &p->z);
    QN(fe_sqr)(F, &z2z2, &q->z);
    QN(fe_mul)(F, &u1, &p->x, &z2z2);
    QN(fe_mul)(F, &u2, &q->x, &z1z1);
    QN(fe_mul)(F, &s1, &p->y, &q->z); QN(fe_mul)(F, &s1, &s1, &z2z2);
    QN(fe_mul)(F, &s2, &q->y, &p->z); QN(fe_mul)(F, &s2, &s2, &z1z1);
    if (QN(fe_eq)(&u1, &u2) && QN(fe_eq)(&s1, &s2)) { QN(jac_double)(F, p); return; }
    QN(fe_sub)(F, &h, &u2, &u1);
    QN(fe_dbl)(F, &i, &h); QN(fe_sqr)(F, &i, &i);
    QN(fe_mul)(F, &j, &h, &i);
    QN(fe_sub)(F, &r, &s2, &s1); QN(fe_dbl)(F, &r, &r);
    QN(fe_mul)(F, &v, &u1, &i);
    /* z3 = ((z1+z2)^2 - z1z1 - z2z2) h */
    QN(fe_add)(F, &t, &p->z, &q->z); QN(fe_sqr)(F, &t, &t);
    QN(fe_sub)(F, &t, &t, &z1z1); QN(fe_sub)(F, &t, &t, &z2z2); QN(fe_mul)(F, &p->z, &t, &h);
    QN(fe_sqr)(F, &t, &r); QN(fe_sub)(F, &t, &t, &j); QN(fe_sub)(F, &t, &t, &v); QN(fe_sub)(F, &p->x, &t, &v);
    QN(fe_sub)(F, &t, &v, &p->x); QN(fe_mul)(F, &t, &r, &t);
    QN(fe_mul)(F, &s1, &s1, &j); QN(fe_dbl)(F, &s1, &s1);
    QN(fe_sub)(F, &p->y, &t, &s1);
}

/* From<GroupProjective> for GroupAffine: z=0 -> infinity; else (X/Z^2, Y/Z^3). */
static void QN(jac_to_affine)(const QCTX *F, AFF *a, const JAC *p) {
    if (QN(jac_is_zero)(p)) { memset(a, 0, sizeof *a); a->y = F->one; a->inf = 1; return; }
    QFE zi, zi2, zi3;
    QN(fe_inv)(F, &zi, &p->z);
    QN(fe_sqr)(F, &zi2, &zi);
    QN(fe_mul)(F, &zi3, &zi2, &zi);
    QN(fe_mul)(F, &a->x, &p->x, &zi2);
    QN(fe_mul)(F, &a->y, &p->y, &zi3);
    a->inf = 0;
}

static inline uint64_t QN(scalar_bits)(const uint64_t *s, int start, int c) {   /* (s >> start) mod 2^c, s = 4 limbs */
    int limb = start >> 6, off = start & 63;
    uint64_t v = s[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
    return v & ((1ull << c) - 1);
}

/* ark-ec 0.3.0 VariableBaseMSM::multi_scalar_mul.  scalars: canonical 4xu64.  Threads over windows
 * (upstream: rayon par_iter over window_starts). */
static void QN(msm)(const QCTX *F, int scalar_bits, const AFF *bases, const uint64_t *scalars, size_t size,
                    JAC *out, int threads) {
    int lg = 0; while (((size_t)1 << lg) < size) lg++;                 /* ark_std::log2 = ceil */
    int c = size < 32 ? 3 : (lg * 69 / 100) + 2;                        /* ln_without_floats + 2 */
    int nwin = (scalar_bits + c - 1) / c;
    JAC *wsum = (JAC *)malloc(sizeof(JAC) * nwin);
    size_t nb = ((size_t)1 << c) - 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int w = 0; w < nwin; w++) {
        int w_start = w * c;
        JAC res; QN(jac_set_zero)(F, &res);
        JAC *buckets = (JAC *)malloc(sizeof(JAC) * nb);
        for (size_t i = 0; i < nb; i++) QN(jac_set_zero)(F, &buckets[i]);
        for (size_t i = 0; i < size; i++) {
            const uint64_t *s = scalars + 4 * i;
            if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;            /* zero scalars filtered */
            if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) {
                if (w_start == 0) QN(jac_add_mixed)(F, &res, &bases[i]);
            } else {
                uint64_t d = QN(scalar_bits)(s, w_start, c);
                if (d) QN(jac_add_mixed)(F, &buckets[d - 1], &bases[i]);
            }
        }
        JAC running; QN(jac_set_zero)(F, &running);
        for (size_t i = nb; i-- > 0;) {
            QN(jac_add)(F, &running, &buckets[i]);
            QN(jac_add)(F, &res, &running);
        }
        free(buckets);
        wsum[w] = res;
    }
    JAC total; QN(jac_set_zero)(F, &total);
    for (int w = nwin - 1; w >= 1; w--) {
        QN(jac_add)(F, &total, &wsum[w]);
        for (int k = 0; k < c; k++) QN(jac_double)(F, &total);
    }
    JAC lowest = wsum[0];
    QN(jac_add)(F, &lowest, &total);
    *out = lowest;
    free(wsum);
}

/* double-and-add k*P (k canonical 4 limbs) — independent check of the MSM */
static void QN(scalar_mul)(const QCTX *F, const AFF *P, const uint64_t *k, JAC *out) {
    JAC acc; QN(jac_set_zero)(F, &acc);
    for (int i = 255; i >= 0; i--) {
        QN(jac_double)(F, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) QN(jac_add_mixed)(F, &acc, P);
    }
    *out = acc;
}

#undef QFE
#undef QCTX
#undef AFF
#undef JAC
