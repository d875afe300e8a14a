/* oracle/field_impl.h — TEST INFRASTRUCTURE ONLY (CPU oracle), not part of the product path.
 *
 * PARITY UNPINNED: see oracle/plonk_oracle.c header.
 *
 * Montgomery prime-field arithmetic on NL x u64 little-endian limbs, restating the published
 * algorithm of ark-ff 0.3.0 `Fp256` / `Fp384` (reference dependency, Cargo.lock:149-152; used at
 * /root/reference/src/worker.rs:79,93,113 `Fr::pow`, and by every arkworks call on the hot path):
 * R = 2^(64*NL), elements stored as a*R mod p, always fully reduced to [0,p)  (SURVEY Appendix A.3).
 *
 * Include with NL and SUF defined; every name gets suffix SUF (template-by-include, plain C).
 */
#ifndef NL
#error "define NL (number of u64 limbs) and SUF before including"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)
#define FE FN(fe)
#define FCTX FN(fctx)

typedef struct { uint64_t l[NL]; } FE;

typedef struct {
    FE p;            /* modulus */
    uint64_t inv;    /* -p^{-1} mod 2^64 */
    FE one;          /* R mod p   (Montgomery 1) */
    FE r2;           /* R^2 mod p */
    int bits;        /* MODULUS_BITS */
} FCTX;

static inline int FN(fe_is_zero)(const FE *a) {
    uint64_t t = 0;
    for (int i = 0; i < NL; i++) t |= a->l[i];
    return t == 0;
}
static inline int FN(fe_eq)(const FE *a, const FE *b) {
    uint64_t t = 0;
    for (int i = 0; i < NL; i++) t |= a->l[i] ^ b->l[i];
    return t == 0;
}
/* a >= b on raw limbs */
static inline int FN(raw_geq)(const uint64_t *a, const uint64_t *b) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline uint64_t FN(raw_add)(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    u128 c = 0;
    for (int i = 0; i < NL; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t FN(raw_sub)(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    uint64_t br = 0;
    for (int i = 0; i < NL; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        r[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static inline void FN(fe_add)(const FCTX *F, FE *r, const FE *a, const FE *b) {
    uint64_t c = FN(raw_add)(r->l, a->l, b->l);
    if (c || FN(raw_geq)(r->l, F->p.l)) FN(raw_sub)(r->l, r->l, F->p.l);
}
static inline void FN(fe_sub)(const FCTX *F, FE *r, const FE *a, const FE *b) {
    if (FN(raw_sub)(r->l, a->l, b->l)) FN(raw_add)(r->l, r->l, F->p.l);
}
static inline void FN(fe_dbl)(const FCTX *F, FE *r, const FE *a) { FN(fe_add)(F, r, a, a); }
static inline void FN(fe_neg)(const FCTX *F, FE *r, const FE *a) {
    if (FN(fe_is_zero)(a)) { *r = *a; return; }
    FN(raw_sub)(r->l, F->p.l, a->l);
}
/* CIOS Montgomery multiplication: r = a*b*R^{-1} mod p, fully reduced. */
static inline void FN(fe_mul)(const FCTX *F, FE *r, const FE *a, const FE *b) {
    uint64_t t[NL + 2];
    for (int i = 0; i < NL + 2; i++) t[i] = 0;
    for (int i = 0; i < NL; i++) {
        u128 c = 0;
        for (int j = 0; j < NL; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c; c >>= 64;
        }
        c += t[NL]; t[NL] = (uint64_t)c; t[NL + 1] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->p.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < NL; j++) {
            c += (u128)m * F->p.l[j] + t[j];
            t[j - 1] = (uint64_t)c; c >>= 64;
        }
        c += t[NL]; t[NL - 1] = (uint64_t)c;
        t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
    }
    if (t[NL] || FN(raw_geq)(t, F->p.l)) FN(raw_sub)(t, t, F->p.l);
    for (int i = 0; i < NL; i++) r->l[i] = t[i];
}
static inline void FN(fe_sqr)(const FCTX *F, FE *r, const FE *a) { FN(fe_mul)(F, r, a, a); }

/* a^e, e a u64 (Fr::pow([e]) of worker.rs:79,93,113) */
static void FN(fe_pow_u64)(const FCTX *F, FE *r, const FE *a, uint64_t e) {
    FE acc = F->one, b = *a;
    while (e) {
        if (e & 1) FN(fe_mul)(F, &acc, &acc, &b);
        FN(fe_sqr)(F, &b, &b);
        e >>= 1;
    }
    *r = acc;
}
/* a^e, e NL limbs */
static void FN(fe_pow)(const FCTX *F, FE *r, const FE *a, const uint64_t *e) {
    FE acc = F->one;
    for (int i = NL * 64 - 1; i >= 0; i--) {
        FN(fe_sqr)(F, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) FN(fe_mul)(F, &acc, &acc, a);
    }
    *r = acc;
}
/* Fermat inverse (0 -> 0) */
static void FN(fe_inv)(const FCTX *F, FE *r, const FE *a) {
    uint64_t e[NL], two[NL];
    for (int i = 0; i < NL; i++) two[i] = 0;
    two[0] = 2;
    FN(raw_sub)(e, F->p.l, two);
    FN(fe_pow)(F, r, a, e);
}
static inline void FN(fe_from_mont)(const FCTX *F, FE *r, const FE *a) {   /* into_repr */
    FE o; for (int i = 0; i < NL; i++) o.l[i] = 0; o.l[0] = 1;
    FN(fe_mul)(F, r, a, &o);
}
static inline void FN(fe_to_mont)(const FCTX *F, FE *r, const FE *a) {     /* from_repr */
    FN(fe_mul)(F, r, a, &F->r2);
}
static void FN(fe_from_u64)(const FCTX *F, FE *r, uint64_t v) {
    FE t; for (int i = 0; i < NL; i++) t.l[i] = 0; t.l[0] = v;
    FN(fe_to_mont)(F, r, &t);
}

/* Derive every Montgomery constant from the modulus alone (no hand-typed constants to get wrong). */
static void FN(fctx_init)(FCTX *F, const uint64_t *p) {
    for (int i = 0; i < NL; i++) F->p.l[i] = p[i];
    /* inv = -p^{-1} mod 2^64 by Newton iteration */
    uint64_t x = 1;
    for (int i = 0; i < 7; i++) x *= 2 - p[0] * x;
    F->inv = (uint64_t)0 - x;
    /* bits */
    int bits = 0;
    for (int i = NL - 1; i >= 0 && !bits; i--)
        if (p[i]) bits = 64 * i + 64 - __builtin_clzll(p[i]);
    F->bits = bits;
    /* one = 2^(64 NL) mod p by 64*NL modular doublings of 1;  r2 by 64*NL more */
    FE t; for (int i = 0; i < NL; i++) t.l[i] = 0; t.l[0] = 1;
    for (int i = 0; i < 64 * NL; i++) FN(fe_add)(F, &t, &t, &t);
    F->one = t;
    for (int i = 0; i < 64 * NL; i++) FN(fe_add)(F, &t, &t, &t);
    F->r2 = t;
}

#undef FE
#undef FCTX
