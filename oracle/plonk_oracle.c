/* oracle/plonk_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / reported CPU baseline.  The product path (distributed_plonk_amd/) never
 * links, imports or calls it.
 *
 * PARITY UNPINNED.  The reference (/root/reference) holds no golden vectors, KATs or fixtures for
 * this path (every reference test is differential against arkworks in-process),
 * its arithmetic lives in un-vendored crates (ark-ff / ark-poly / ark-ec 0.3.0, ark-bls12-381 0.3.0,
 * ark-bn254 0.3.0 — Cargo.lock:66-80,99-102,149-152,190-193) and no Rust toolchain exists here, so
 * the reference cannot be run to produce vectors.  This file restates those crates' published
 * algorithms and the reference's in-tree orchestration.  What pins it instead: exact integer math
 * (any correct implementation with the same p, R, omega, g is bit-identical on reduced residues and
 * on affine points), an independent pure-Python big-int statement (oracle/bigint_ref.py), O(N^2)
 * DFTs and double-and-add checks, tests/golden/ vectors generated from bigint_ref.py — and, since
 * round 6, THIRD-PARTY code that this repository neither wrote nor ships: sympy's number-theoretic
 * transform (sympy.discrete.transforms.ntt / intt; its primitive_root is arkworks' GENERATOR 5 / 7,
 * hence the same omega) and sympy's elliptic-curve group law, run live against this file
 * (tests/test_oracle_thirdparty.py: all four transform modes for 2 ... 2^10 points on both Fr, the
 * 4-step and distributed decompositions, P + Q / 2P / P - P / k * P / MSMs on both curves) and
 * committed as a second fixture set (tests/golden/sympy_*.json, tools/gen_golden_sympy.py) that the
 * GPU golden test consumes.  Still "unpinned" by the task's definition: none of it is arkworks output.
 *
 * Restated reference functions (file:line under /root/reference/src):
 *   orc_ntt               ark-poly Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place
 *                         as called at worker.rs:82,84,105,107,398; dispatcher.rs:594,632,667;
 *                         dispatcher2.rs:507  
 *   orc_fourstep          playground.rs:21-80
 *   orc_fft1_helper       worker.rs:66-94
 *   orc_fft2_helper       worker.rs:96-115
 *   orc_exchange_pack     worker.rs:327-330
 *   orc_exchange_scatter  worker.rs:432-435
 *   orc_distributed_fft   dispatcher2.rs:732-787 (+ worker.rs:187-381,412-438)
 *   orc_msm               ark-ec VariableBaseMSM::multi_scalar_mul as called at worker.rs:179-182
 *   orc_sharded_msm       dispatcher.rs:218-238 / dispatcher2.rs:870-890
 *   orc_commit_polynomial worker.rs:117-123
 *   orc_round1            worker.rs:383-408 (blinders supplied explicitly; the reference draws them
 *                         from thread_rng)
 *   orc_quotient_evals    dispatcher2.rs:362-504 (coset evaluations of the quotient polynomial)
 *
 * Layout contract (utils.rs:27-43): Fr = 4xu64 LE Montgomery; scalars = 4xu64 LE canonical;
 * Fq = 4xu64 (BN254) / 6xu64 (BLS12-381) Montgomery; Jacobian = X||Y||Z.
 *
 * Build: gcc -O3 -fopenmp -shared -fPIC plonk_oracle.c -o libplonk_oracle.so   (oracle/build.py)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef unsigned __int128 u128;

#define NL 4
#define SUF 4
#include "field_impl.h"
#undef NL
#undef SUF
#undef FN
#define NL 6
#define SUF 6
#include "field_impl.h"
#undef NL
#undef SUF
#undef FN

#define QS 4
#include "curve_impl.h"
#undef QS
#undef QN
#define QS 6
#include "curve_impl.h"
#undef QS
#undef QN

/* ------------------------------------------------------------------------------------------------
 * Curve / field parameters (only moduli, generators and curve constants are
 * typed in — every Montgomery constant and root of unity is derived at init).
 * ---------------------------------------------------------------------------------------------- */
enum { CURVE_BN254 = 0, CURVE_BLS12_381 = 1 };

typedef struct {
    int ready;
    fctx4 fr;
    uint64_t fr_gen;          /* multiplicative generator (coset shift): 5 / 7 */
    int two_adicity;          /* 28 / 32 */
    fe4 two_adic_root;        /* g^((p-1)/2^s), Montgomery */
    int fq_limbs;             /* 4 / 6 */
    fctx4 fq4;
    fctx6 fq6;
} curve_t;

static curve_t g_curves[2];

static const uint64_t BN254_FR_P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t BN254_FQ_P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t BLS_FR_P[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const uint64_t BLS_FQ_P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                     0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};

static curve_t *get_curve(int id) {
    curve_t *C = &g_curves[id];
    if (C->ready) return C;
#pragma omp critical(orc_curve_init)
    if (!C->ready) {
        if (id == CURVE_BN254) {
            fctx_init4(&C->fr, BN254_FR_P); C->fr_gen = 5; C->two_adicity = 28;
            C->fq_limbs = 4; fctx_init4(&C->fq4, BN254_FQ_P);
        } else {
            fctx_init4(&C->fr, BLS_FR_P); C->fr_gen = 7; C->two_adicity = 32;
            C->fq_limbs = 6; fctx_init6(&C->fq6, BLS_FQ_P);
        }
        /* two_adic_root = g^((p-1) >> s) */
        uint64_t e[4]; fe4 g;
        for (int i = 0; i < 4; i++) e[i] = C->fr.p.l[i];
        e[0] -= 1;
        int s = C->two_adicity;
        for (int i = 0; i < 4; i++) {
            uint64_t lo = e[i] >> s;
            uint64_t hi = (i + 1 < 4) ? e[i + 1] << (64 - s) : 0;
            e[i] = lo | hi;
        }
        fe_from_u644(&C->fr, &g, C->fr_gen);
        fe_pow4(&C->fr, &C->two_adic_root, &g, e);
        C->ready = 1;
    }
    return C;
}

/* ark-ff FftField::get_root_of_unity(n = 2^log_n) */
static int root_of_unity(const curve_t *C, int log_n, fe4 *w) {
    if (log_n > C->two_adicity) return -1;
    *w = C->two_adic_root;
    for (int i = 0; i < C->two_adicity - log_n; i++) fe_sqr4(&C->fr, w, w);
    return 0;
}

typedef struct {
    size_t size; int log_size;
    fe4 group_gen, group_gen_inv, size_inv, gen, gen_inv;
} domain_t;

static int domain_new(const curve_t *C, int log_n, domain_t *D) {
    D->size = (size_t)1 << log_n; D->log_size = log_n;
    if (root_of_unity(C, log_n, &D->group_gen)) return -1;       /* DomainCreationError */
    fe_inv4(&C->fr, &D->group_gen_inv, &D->group_gen);
    fe4 n; fe_from_u644(&C->fr, &n, (uint64_t)D->size);
    fe_inv4(&C->fr, &D->size_inv, &n);
    fe_from_u644(&C->fr, &D->gen, C->fr_gen);
    fe_inv4(&C->fr, &D->gen_inv, &D->gen);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * ark-poly 0.3.0 radix-2 NTT (Appendix A.2): forward = Gentleman-Sande butterflies on in-order
 * input then bit-reversal ("derange"); inverse = derange, Cooley-Tukey butterflies with
 * group_gen_inv, then * size_inv.
 * ---------------------------------------------------------------------------------------------- */
static void derange(fe4 *v, int log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 1; i < n; i++) {
        size_t r = 0, x = i;
        for (int b = 0; b < log_n; b++) { r = (r << 1) | (x & 1); x >>= 1; }
        if (i < r) { fe4 t = v[i]; v[i] = v[r]; v[r] = t; }
    }
}

static fe4 *powers_table(const fctx4 *F, const fe4 *w, size_t count) {
    fe4 *t = (fe4 *)malloc(sizeof(fe4) * (count ? count : 1));
    fe4 acc = F->one;
    for (size_t i = 0; i < count; i++) { t[i] = acc; fe_mul4(F, &acc, &acc, w); }
    return t;
}

static void ntt_forward(const fctx4 *F, fe4 *v, int log_n, const fe4 *root, int threads) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) return;
    fe4 *roots = powers_table(F, root, n / 2);
    /* io_helper: gap = n/2 .. 1 ; twiddle index step = n/(2 gap) */
    for (size_t gap = n / 2; gap >= 1; gap >>= 1) {
        size_t step = n / (2 * gap), nchunks = n / (2 * gap);
        if (nchunks >= (size_t)threads * 4 || threads == 1) {
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1 && n >= 4096)
            for (size_t ch = 0; ch < nchunks; ch++) {
                fe4 *base = &v[ch * 2 * gap];
                for (size_t k = 0; k < gap; k++) {
                    fe4 *lo = base + k, *hi = lo + gap, d;
                    fe_sub4(F, &d, lo, hi);
                    fe_add4(F, lo, lo, hi);
                    fe_mul4(F, hi, &d, &roots[k * step]);
                }
            }
        } else {
            for (size_t ch = 0; ch < nchunks; ch++) {
                fe4 *base = &v[ch * 2 * gap];
#pragma omp parallel for schedule(static) num_threads(threads)
                for (size_t k = 0; k < gap; k++) {
                    fe4 *lo = base + k, *hi = lo + gap, d;
                    fe_sub4(F, &d, lo, hi);
                    fe_add4(F, lo, lo, hi);
                    fe_mul4(F, hi, &d, &roots[k * step]);
                }
            }
        }
    }
    free(roots);
    derange(v, log_n);
}

static void ntt_inverse_core(const fctx4 *F, fe4 *v, int log_n, const fe4 *root_inv, int threads) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) return;
    derange(v, log_n);
    fe4 *roots = powers_table(F, root_inv, n / 2);
    for (size_t gap = 1; gap < n; gap <<= 1) {
        size_t step = n / (2 * gap), nchunks = n / (2 * gap);
        if (nchunks >= (size_t)threads * 4 || threads == 1) {
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1 && n >= 4096)
            for (size_t ch = 0; ch < nchunks; ch++) {
                fe4 *base = &v[ch * 2 * gap];
                for (size_t k = 0; k < gap; k++) {
                    fe4 *lo = base + k, *hi = lo + gap, t, s2;
                    fe_mul4(F, &t, hi, &roots[k * step]);
                    fe_sub4(F, &s2, lo, &t);
                    fe_add4(F, lo, lo, &t);
                    *hi = s2;
                }
            }
        } else {
            for (size_t ch = 0; ch < nchunks; ch++) {
                fe4 *base = &v[ch * 2 * gap];
#pragma omp parallel for schedule(static) num_threads(threads)
                for (size_t k = 0; k < gap; k++) {
                    fe4 *lo = base + k, *hi = lo + gap, t, s2;
                    fe_mul4(F, &t, hi, &roots[k * step]);
                    fe_sub4(F, &s2, lo, &t);
                    fe_add4(F, lo, lo, &t);
                    *hi = s2;
                }
            }
        }
    }
    free(roots);
}

static void distribute_powers(const fctx4 *F, fe4 *v, size_t n, const fe4 *g) {
    fe4 pw = F->one;
    for (size_t i = 0; i < n; i++) { fe_mul4(F, &v[i], &v[i], &pw); fe_mul4(F, &pw, &pw, g); }
}

static void domain_fft(const curve_t *C, const domain_t *D, fe4 *v, int is_inv, int is_coset, int threads) {
    const fctx4 *F = &C->fr;
    if (!is_inv) {
        if (is_coset) distribute_powers(F, v, D->size, &D->gen);
        ntt_forward(F, v, D->log_size, &D->group_gen, threads);
    } else {
        ntt_inverse_core(F, v, D->log_size, &D->group_gen_inv, threads);
#pragma omp parallel for schedule(static) num_threads(threads) if (D->size >= 4096)
        for (size_t i = 0; i < D->size; i++) fe_mul4(F, &v[i], &v[i], &D->size_inv);
        if (is_coset) distribute_powers(F, v, D->size, &D->gen_inv);
    }
}

int orc_ntt(int curve, uint64_t *v, int log_n, int is_inv, int is_coset, int threads) {
    curve_t *C = get_curve(curve); domain_t D;
    if (domain_new(C, log_n, &D)) return -1;
    domain_fft(C, &D, (fe4 *)v, is_inv, is_coset, threads > 0 ? threads : 1);
    return 0;
}

/* O(N^2) DFT, for pinning orc_ntt on tiny sizes: out[k] = sum_j v[j] w^(jk) */
int orc_naive_dft(int curve, const uint64_t *v, uint64_t *out, int log_n, int is_inv) {
    curve_t *C = get_curve(curve); domain_t D; const fctx4 *F = &C->fr;
    if (domain_new(C, log_n, &D)) return -1;
    const fe4 *in = (const fe4 *)v; fe4 *o = (fe4 *)out;
    const fe4 *w = is_inv ? &D.group_gen_inv : &D.group_gen;
    for (size_t k = 0; k < D.size; k++) {
        fe4 wk, pw = F->one, acc; memset(&acc, 0, sizeof acc);
        fe_pow_u644(F, &wk, w, k);
        for (size_t j = 0; j < D.size; j++) {
            fe4 t; fe_mul4(F, &t, &in[j], &pw); fe_add4(F, &acc, &acc, &t);
            fe_mul4(F, &pw, &pw, &wk);
        }
        if (is_inv) fe_mul4(F, &acc, &acc, &D.size_inv);
        o[k] = acc;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * The reference's 2-D decomposition.
 * ---------------------------------------------------------------------------------------------- */
static void split_rc(int log_n, size_t *r, size_t *c) {        /* worker.rs:143-144 */
    *r = (size_t)1 << (log_n >> 1);
    *c = ((size_t)1 << log_n) / *r;
}

/* worker.rs:66-94.  v: one row of c elements, i = global row index. */
int orc_fft1_helper(int curve, uint64_t *v_, uint64_t i, int log_n, int is_inv, int is_coset) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr; domain_t D, Dc;
    size_t r, c; split_rc(log_n, &r, &c);
    int log_c = log_n - (log_n >> 1);
    if (domain_new(C, log_n, &D) || domain_new(C, log_c, &Dc)) return -1;
    fe4 *v = (fe4 *)v_;
    if (is_coset && !is_inv)
        for (size_t j = 0; j < c; j++) { fe4 t; fe_pow_u644(F, &t, &D.gen, i + j * r); fe_mul4(F, &v[j], &v[j], &t); }
    domain_fft(C, &Dc, v, is_inv, 0, 1);
    const fe4 *shift = is_inv ? &D.group_gen_inv : &D.group_gen;
    for (size_t j = 0; j < c; j++) { fe4 t; fe_pow_u644(F, &t, shift, i * j); fe_mul4(F, &v[j], &v[j], &t); }
    return 0;
}

/* worker.rs:96-115.  v: one column of r elements, i = global column index. */
int orc_fft2_helper(int curve, uint64_t *v_, uint64_t i, int log_n, int is_inv, int is_coset) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr; domain_t D, Dr;
    size_t r, c; split_rc(log_n, &r, &c);
    if (domain_new(C, log_n, &D) || domain_new(C, log_n >> 1, &Dr)) return -1;
    fe4 *v = (fe4 *)v_;
    domain_fft(C, &Dr, v, is_inv, 0, 1);
    if (is_coset && is_inv)
        for (size_t j = 0; j < r; j++) { fe4 t; fe_pow_u644(F, &t, &D.gen_inv, i + j * c); fe_mul4(F, &v[j], &v[j], &t); }
    return 0;
}

/* worker.rs:327-330: block sent to a peer = rows[.][col_start..col_end], row-major. */
void orc_exchange_pack(const uint64_t *rows, size_t n_rows, size_t c, size_t col_start, size_t col_end, uint64_t *out) {
    const fe4 *R = (const fe4 *)rows; fe4 *o = (fe4 *)out;
    size_t nc = col_end - col_start;
    for (size_t a = 0; a < n_rows; a++)
        for (size_t b = 0; b < nc; b++) o[a * nc + b] = R[a * c + col_start + b];
}
/* worker.rs:432-435: cols[i % nc][from_row_start + i / nc] = v[i]. cols: nc x r row-major. */
void orc_exchange_scatter(uint64_t *cols, size_t nc, size_t r, size_t from_row_start, const uint64_t *v, size_t len) {
    fe4 *Cc = (fe4 *)cols; const fe4 *V = (const fe4 *)v;
    for (size_t i = 0; i < len; i++) Cc[(i % nc) * r + from_row_start + i / nc] = V[i];
}

/* dispatcher2.rs:732-787 with S in-process workers.  v: N elements natural order, in place. */
int orc_distributed_fft(int curve, uint64_t *v_, int log_n, int S, int is_inv, int is_coset) {
    size_t r, c; split_rc(log_n, &r, &c);
    size_t N = (size_t)1 << log_n;
    if (r % S || c % S) return -2;
    fe4 *v = (fe4 *)v_;
    fe4 *t = (fe4 *)malloc(sizeof(fe4) * N), *cols = (fe4 *)malloc(sizeof(fe4) * N);
    fe4 *blk = (fe4 *)malloc(sizeof(fe4) * (r / S) * (c / S));
    /* :754  t[b][a] = coeffs[a*r + b] */
    for (size_t b = 0; b < r; b++) for (size_t a = 0; a < c; a++) t[b * c + a] = v[a * r + b];
    int rc = 0;
    for (size_t b = 0; b < r; b++) rc |= orc_fft1_helper(curve, (uint64_t *)&t[b * c], b, log_n, is_inv, is_coset);
    /* exchange: worker s holds rows [s r/S,(s+1) r/S); worker d receives cols [d c/S,(d+1) c/S) */
    for (int s = 0; s < S; s++)
        for (int d = 0; d < S; d++) {
            size_t rs = s * r / S, re = (s + 1) * r / S, cs = d * c / S, ce = (d + 1) * c / S;
            orc_exchange_pack((uint64_t *)&t[rs * c], re - rs, c, cs, ce, (uint64_t *)blk);
            orc_exchange_scatter((uint64_t *)&cols[cs * r], ce - cs, r, rs, (uint64_t *)blk, (re - rs) * (ce - cs));
        }
    for (size_t i = 0; i < c; i++) rc |= orc_fft2_helper(curve, (uint64_t *)&cols[i * r], i, log_n, is_inv, is_coset);
    /* :780-786  out[j*c + i] = u[i][j] */
    for (size_t i = 0; i < c; i++) for (size_t j = 0; j < r; j++) v[j * c + i] = cols[i * r + j];
    free(t); free(cols); free(blk);
    return rc;
}

/* playground.rs:21-80 */
int orc_fourstep(int curve, uint64_t *v_, int log_n, int is_inv, int is_coset) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr; domain_t D, Dr, Dc;
    size_t r, c; split_rc(log_n, &r, &c);
    size_t N = (size_t)1 << log_n;
    if (domain_new(C, log_n, &D) || domain_new(C, log_n >> 1, &Dr) || domain_new(C, log_n - (log_n >> 1), &Dc)) return -1;
    fe4 *v = (fe4 *)v_;
    if (is_coset && !is_inv) distribute_powers(F, v, N, &D.gen);
    fe4 *t = (fe4 *)malloc(sizeof(fe4) * N), *g = (fe4 *)malloc(sizeof(fe4) * N);
    for (size_t b = 0; b < r; b++) for (size_t a = 0; a < c; a++) t[b * c + a] = v[a * r + b];
    const fe4 *w = is_inv ? &D.group_gen_inv : &D.group_gen;
    for (size_t i = 0; i < r; i++) {
        domain_fft(C, &Dc, &t[i * c], is_inv, 0, 1);
        for (size_t j = 0; j < c; j++) { fe4 tw; fe_pow_u644(F, &tw, w, i * j); fe_mul4(F, &t[i * c + j], &t[i * c + j], &tw); }
    }
    for (size_t j = 0; j < c; j++) for (size_t i = 0; i < r; i++) g[j * r + i] = t[i * c + j];
    for (size_t j = 0; j < c; j++) domain_fft(C, &Dr, &g[j * r], is_inv, 0, 1);
    for (size_t j = 0; j < c; j++) for (size_t i = 0; i < r; i++) v[i * c + j] = g[j * r + i];
    if (is_coset && is_inv) distribute_powers(F, v, N, &D.gen_inv);
    free(t); free(g);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Fr helpers (element-wise; used to pin the device field arithmetic).
 * op: 0 mul, 1 add, 2 sub, 3 to_mont(a), 4 from_mont(a) [into_repr], 5 inverse(a), 6 square(a)
 * field: 0 = Fr, 1 = Fq of the curve
 * ---------------------------------------------------------------------------------------------- */
int orc_field_op(int curve, int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
    curve_t *C = get_curve(curve);
    if (field == 0 || C->fq_limbs == 4) {
        const fctx4 *F = field == 0 ? &C->fr : &C->fq4;
        const fe4 *A = (const fe4 *)a, *B = (const fe4 *)b; fe4 *O = (fe4 *)out;
        for (size_t i = 0; i < n; i++) switch (op) {
            case 0: fe_mul4(F, &O[i], &A[i], &B[i]); break;
            case 1: fe_add4(F, &O[i], &A[i], &B[i]); break;
            case 2: fe_sub4(F, &O[i], &A[i], &B[i]); break;
            case 3: fe_to_mont4(F, &O[i], &A[i]); break;
            case 4: fe_from_mont4(F, &O[i], &A[i]); break;
            case 5: fe_inv4(F, &O[i], &A[i]); break;
            case 6: fe_sqr4(F, &O[i], &A[i]); break;
            default: return -1;
        }
    } else {
        const fctx6 *F = &C->fq6;
        const fe6 *A = (const fe6 *)a, *B = (const fe6 *)b; fe6 *O = (fe6 *)out;
        for (size_t i = 0; i < n; i++) switch (op) {
            case 0: fe_mul6(F, &O[i], &A[i], &B[i]); break;
            case 1: fe_add6(F, &O[i], &A[i], &B[i]); break;
            case 2: fe_sub6(F, &O[i], &A[i], &B[i]); break;
            case 3: fe_to_mont6(F, &O[i], &A[i]); break;
            case 4: fe_from_mont6(F, &O[i], &A[i]); break;
            case 5: fe_inv6(F, &O[i], &A[i]); break;
            case 6: fe_sqr6(F, &O[i], &A[i]); break;
            default: return -1;
        }
    }
    return 0;
}

/* Field constants out (for cross-checking generated product headers): which: 0 modulus, 1 R (one),
 * 2 R^2, 3 two-adic root (Fr only), 4 root of unity of order 2^arg (Fr only). Returns limb count. */
int orc_field_const(int curve, int field, int which, int arg, uint64_t *out) {
    curve_t *C = get_curve(curve);
    if (field == 1 && C->fq_limbs == 6) {
        const fe6 *s = which == 0 ? &C->fq6.p : which == 1 ? &C->fq6.one : which == 2 ? &C->fq6.r2 : NULL;
        if (!s) return -1;
        memcpy(out, s, sizeof *s); return 6;
    }
    const fctx4 *F = field == 0 ? &C->fr : &C->fq4;
    fe4 t;
    switch (which) {
        case 0: t = F->p; break;
        case 1: t = F->one; break;
        case 2: t = F->r2; break;
        case 3: if (field) return -1; t = C->two_adic_root; break;
        case 4: if (field || root_of_unity(C, arg, &t)) return -1; break;
        default: return -1;
    }
    memcpy(out, &t, sizeof t); return 4;
}
uint64_t orc_field_inv64(int curve, int field) {
    curve_t *C = get_curve(curve);
    if (field == 0) return C->fr.inv;
    return C->fq_limbs == 4 ? C->fq4.inv : C->fq6.inv;
}

/* ------------------------------------------------------------------------------------------------
 * Seeded synthetic inputs (the reference uses unseeded thread_rng: dispatcher.rs:187-200).
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* ark-ff Fp::rand: draw limbs, mask the top REPR_SHAVE_BITS, accept if < p; the accepted raw limbs
 * ARE the Montgomery representation (Appendix A.3).  Element i uses its own stream (seed, i) so the
 * device generator can reproduce it in parallel. */
void orc_rand_fr(int curve, uint64_t seed, size_t n, uint64_t *out) {
    curve_t *C = get_curve(curve);
    int shave = 256 - C->fr.bits;
    for (size_t i = 0; i < n; i++) {
        uint64_t s = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(i + 1));
        uint64_t l[4];
        do {
            for (int k = 0; k < 4; k++) l[k] = splitmix64(&s);
            l[3] &= (~0ull) >> shave;
        } while (raw_geq4(l, C->fr.p.l));
        memcpy(out + 4 * i, l, 32);
    }
}

/* ------------------------------------------------------------------------------------------------
 * G1 / MSM.  bases: n x (2*fq_limbs) u64 = x||y Montgomery, plus inf[n] flags (0/1).
 * ---------------------------------------------------------------------------------------------- */
#define LOAD_BASES(Q, AFFT)                                                           \
    AFFT *B = (AFFT *)malloc(sizeof(AFFT) * (n ? n : 1));                             \
    for (size_t i = 0; i < n; i++) {                                                  \
        memcpy(&B[i].x, bases + (2 * Q) * i, 8 * Q);                                  \
        memcpy(&B[i].y, bases + (2 * Q) * i + Q, 8 * Q);                              \
        B[i].inf = inf ? inf[i] : 0;                                                  \
    }

int orc_msm(int curve, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars, size_t n,
            uint64_t *out_jac, int threads) {
    curve_t *C = get_curve(curve);
    if (threads < 1) threads = 1;
    if (C->fq_limbs == 4) {
        LOAD_BASES(4, aff4)
        jac4 r; msm4(&C->fq4, C->fr.bits, B, scalars, n, &r, threads);
        memcpy(out_jac, &r, sizeof r); free(B);
    } else {
        LOAD_BASES(6, aff6)
        jac6 r; msm6(&C->fq6, C->fr.bits, B, scalars, n, &r, threads);
        memcpy(out_jac, &r, sizeof r); free(B);
    }
    return 0;
}

/* naive sum_i s_i * P_i by double-and-add (tiny n only) */
int orc_msm_naive(int curve, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars, size_t n, uint64_t *out_jac) {
    curve_t *C = get_curve(curve);
    if (C->fq_limbs == 4) {
        LOAD_BASES(4, aff4)
        jac4 acc, t; jac_set_zero4(&C->fq4, &acc);
        for (size_t i = 0; i < n; i++) { scalar_mul4(&C->fq4, &B[i], scalars + 4 * i, &t); jac_add4(&C->fq4, &acc, &t); }
        memcpy(out_jac, &acc, sizeof acc); free(B);
    } else {
        LOAD_BASES(6, aff6)
        jac6 acc, t; jac_set_zero6(&C->fq6, &acc);
        for (size_t i = 0; i < n; i++) { scalar_mul6(&C->fq6, &B[i], scalars + 4 * i, &t); jac_add6(&C->fq6, &acc, &t); }
        memcpy(out_jac, &acc, sizeof acc); free(B);
    }
    return 0;
}

/* a + b on Jacobian triples (dispatcher.rs:236-238 reduce) */
int orc_jac_add(int curve, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    curve_t *C = get_curve(curve);
    if (C->fq_limbs == 4) { jac4 x, y; memcpy(&x, a, sizeof x); memcpy(&y, b, sizeof y); jac_add4(&C->fq4, &x, &y); memcpy(out, &x, sizeof x); }
    else { jac6 x, y; memcpy(&x, a, sizeof x); memcpy(&y, b, sizeof y); jac_add6(&C->fq6, &x, &y); memcpy(out, &x, sizeof x); }
    return 0;
}

/* Commitment(commitment.into()) (dispatcher2.rs:892): out_xy = x||y Montgomery; returns infinity flag */
int orc_jac_to_affine(int curve, const uint64_t *jac, uint64_t *out_xy) {
    curve_t *C = get_curve(curve);
    if (C->fq_limbs == 4) {
        jac4 p; aff4 a; memcpy(&p, jac, sizeof p); jac_to_affine4(&C->fq4, &a, &p);
        memcpy(out_xy, &a.x, 32); memcpy(out_xy + 4, &a.y, 32); return a.inf;
    } else {
        jac6 p; aff6 a; memcpy(&p, jac, sizeof p); jac_to_affine6(&C->fq6, &a, &p);
        memcpy(out_xy, &a.x, 48); memcpy(out_xy + 6, &a.y, 48); return a.inf;
    }
}

int orc_on_curve(int curve, const uint64_t *xy) {
    curve_t *C = get_curve(curve);
    if (C->fq_limbs == 4) {
        const fctx4 *F = &C->fq4; fe4 x, y, l, r, b;
        memcpy(&x, xy, 32); memcpy(&y, xy + 4, 32);
        fe_sqr4(F, &l, &y); fe_sqr4(F, &r, &x); fe_mul4(F, &r, &r, &x);
        fe_from_u644(F, &b, 3); fe_add4(F, &r, &r, &b);
        return fe_eq4(&l, &r);
    } else {
        const fctx6 *F = &C->fq6; fe6 x, y, l, r, b;
        memcpy(&x, xy, 48); memcpy(&y, xy + 6, 48);
        fe_sqr6(F, &l, &y); fe_sqr6(F, &r, &x); fe_mul6(F, &r, &r, &x);
        fe_from_u646(F, &b, 4); fe_add6(F, &r, &r, &b);
        return fe_eq6(&l, &r);
    }
}

/* Generator (Montgomery x||y). */
static const uint64_t BLS_GX[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                                   0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
static const uint64_t BLS_GY[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                                   0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
void orc_generator(int curve, uint64_t *out_xy) {
    curve_t *C = get_curve(curve);
    if (C->fq_limbs == 4) {
        fe4 x, y; fe_from_u644(&C->fq4, &x, 1); fe_from_u644(&C->fq4, &y, 2);
        memcpy(out_xy, &x, 32); memcpy(out_xy + 4, &y, 32);
    } else {
        fe6 x, y; memcpy(&x, BLS_GX, 48); memcpy(&y, BLS_GY, 48);
        fe_to_mont6(&C->fq6, &x, &x); fe_to_mont6(&C->fq6, &y, &y);
        memcpy(out_xy, &x, 48); memcpy(out_xy + 6, &y, 48);
    }
}

/* k*P, k canonical 4 limbs; P affine x||y; result Jacobian */
int orc_scalar_mul(int curve, const uint64_t *xy, const uint64_t *k, uint64_t *out_jac) {
    curve_t *C = get_curve(curve);
    if (C->fq_limbs == 4) { aff4 P; jac4 r; memcpy(&P.x, xy, 32); memcpy(&P.y, xy + 4, 32); P.inf = 0; scalar_mul4(&C->fq4, &P, k, &r); memcpy(out_jac, &r, sizeof r); }
    else { aff6 P; jac6 r; memcpy(&P.x, xy, 48); memcpy(&P.y, xy + 6, 48); P.inf = 0; scalar_mul6(&C->fq6, &P, k, &r); memcpy(out_jac, &r, sizeof r); }
    return 0;
}

/* n affine points: like dispatcher.rs:190-196 — `unique` random points k_j*G (k_j seeded), tiled up
 * to n.  unique == n gives all-distinct bases. */
int orc_gen_bases(int curve, uint64_t seed, size_t unique, size_t n, uint64_t *out_xy) {
    curve_t *C = get_curve(curve);
    int Q = C->fq_limbs;
    uint64_t g[12], k[4], jac[18];
    orc_generator(curve, g);
    uint64_t *sc = (uint64_t *)malloc(32 * unique);
    orc_rand_fr(curve, seed, unique, sc);
#pragma omp parallel for schedule(dynamic, 16) private(k, jac)
    for (size_t j = 0; j < unique; j++) {
        /* raw limbs drawn < p are used directly as the canonical scalar */
        memcpy(k, sc + 4 * j, 32);
        orc_scalar_mul(curve, g, k, jac);
        orc_jac_to_affine(curve, jac, out_xy + 2 * Q * j);
    }
    free(sc);
    for (size_t i = unique; i < n; i++) memcpy(out_xy + 2 * Q * i, out_xy + 2 * Q * (i % unique), 16 * Q);
    return 0;
}

/* dispatcher.rs:218-238: contiguous shards + reduce */
int orc_sharded_msm(int curve, const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars, size_t n, int S,
                    uint64_t *out_jac, int threads) {
    curve_t *C = get_curve(curve);
    int Q = C->fq_limbs;
    uint64_t acc[18], part[18];
    for (int i = 0; i < S; i++) {
        size_t lo = i * n / S, hi = (i + 1) * n / S;
        orc_msm(curve, bases + 2 * Q * lo, inf ? inf + lo : NULL, scalars + 4 * lo, hi - lo, part, threads);
        if (i == 0) memcpy(acc, part, 24 * Q); else orc_jac_add(curve, acc, part, acc);
    }
    memcpy(out_jac, acc, 24 * Q);
    return 0;
}

/* worker.rs:117-123 */
int orc_commit_polynomial(int curve, const uint64_t *bases, const uint8_t *inf, size_t n_bases,
                          const uint64_t *coeffs_mont, size_t n_coeffs, uint64_t *out_jac, int threads) {
    curve_t *C = get_curve(curve);
    if (n_coeffs > n_bases) n_coeffs = n_bases;     /* MSM takes min(len) */
    uint64_t *sc = (uint64_t *)calloc(n_bases ? n_bases : 1, 32);
    for (size_t i = 0; i < n_coeffs; i++) fe_from_mont4(&C->fr, (fe4 *)(sc + 4 * i), (const fe4 *)(coeffs_mont + 4 * i));
    int rc = orc_msm(curve, bases, inf, sc, n_bases, out_jac, threads);
    free(sc);
    return rc;
}

/* worker.rs:383-408 with the two blinding coefficients supplied (b0 + b1 X) * (X^n - 1) + ifft(evals).
 * evals: n Fr Montgomery in, poly_out: n+2 coefficients Montgomery out; commitment Jacobian out. */
int orc_round1(int curve, const uint64_t *bases, const uint8_t *inf, size_t n_bases, const uint64_t *evals, int log_n,
               const uint64_t *blind2, uint64_t *poly_out, uint64_t *out_jac, int threads) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr;
    size_t n = (size_t)1 << log_n;
    fe4 *p = (fe4 *)poly_out;
    memcpy(p, evals, 32 * n);
    memset(&p[n], 0, 64);
    if (orc_ntt(curve, poly_out, log_n, 1, 0, threads)) return -1;
    const fe4 *b = (const fe4 *)blind2;
    /* mul_by_vanishing_poly: b(X) * X^n - b(X) */
    fe_sub4(F, &p[0], &p[0], &b[0]); fe_sub4(F, &p[1], &p[1], &b[1]);
    fe_add4(F, &p[n], &p[n], &b[0]); fe_add4(F, &p[n + 1], &p[n + 1], &b[1]);
    return orc_commit_polynomial(curve, bases, inf, n_bases, poly_out, n + 2, out_jac, threads);
}

/* dispatcher2.rs:362-504: coset evaluations of the TurboPlonk quotient polynomial, pointwise over the m = 8n
 * points x_i = g * w_m^i.  Inputs are the coset-FFT outputs the reference computes at :382-432 (Montgomery Fr):
 * sel[13][m] in the order q_lc[4], q_mul[2], q_hash[4], q_o, q_c, q_ecc (:443-456), sig[5][m], wire[5][m],
 * z[m] (permutation product polynomial), pi[m] (public input).  alpha/beta/gamma: transcript challenges,
 * k[5]: vk.k (coset representatives).  The local coset_ifft of :507 is orc_ntt(inv, coset). */
int orc_quotient_evals(int curve, int log_n, const uint64_t *sel, const uint64_t *sig, const uint64_t *wire, const uint64_t *z_,
                       const uint64_t *pi_, const uint64_t *alpha_, const uint64_t *beta_, const uint64_t *gamma_, const uint64_t *k_,
                       uint64_t *out_, int threads) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr;
    const int log_m = log_n + 3;
    domain_t Dm;
    if (domain_new(C, log_m, &Dm)) return -1;
    const size_t n = (size_t)1 << log_n, m = (size_t)1 << log_m, ratio = m / n;
    const fe4 *S = (const fe4 *)sel, *G = (const fe4 *)sig, *Wv = (const fe4 *)wire, *Z = (const fe4 *)z_, *PI = (const fe4 *)pi_;
    const fe4 alpha = *(const fe4 *)alpha_, beta = *(const fe4 *)beta_, gamma = *(const fe4 *)gamma_;
    const fe4 *K = (const fe4 *)k_;
    fe4 *out = (fe4 *)out_;
    fe4 nf, ninv, a2n;                                     /* alpha^2 / n  (:363) */
    fe_from_u644(F, &nf, (uint64_t)n); fe_inv4(F, &ninv, &nf);
    fe_sqr4(F, &a2n, &alpha); fe_mul4(F, &a2n, &a2n, &ninv);
    fe4 *xs = (fe4 *)malloc(sizeof(fe4) * m);               /* eval_points (:366-369) */
    xs[0] = Dm.gen;
    for (size_t i = 1; i < m; i++) fe_mul4(F, &xs[i], &xs[i - 1], &Dm.group_gen);
    fe4 zh_inv[8];                                          /* 1 / Z_H(x_i), i < m/n (:372-379) */
    for (size_t i = 0; i < ratio; i++) {
        fe4 t; fe_pow_u644(F, &t, &xs[i], (uint64_t)n); fe_sub4(F, &t, &t, &F->one); fe_inv4(F, &zh_inv[i], &t);
    }
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t i = 0; i < m; i++) {
        const fe4 x = xs[i];
        const fe4 a = Wv[0 * m + i], b = Wv[1 * m + i], c = Wv[2 * m + i], d = Wv[3 * m + i], e = Wv[4 * m + i];
        fe4 ab, cd, t, gate, p5;
        fe_mul4(F, &ab, &a, &b); fe_mul4(F, &cd, &c, &d);
        fe_add4(F, &gate, &S[11 * m + i], &PI[i]);                                  /* q_c + pub_input */
        fe_mul4(F, &t, &S[0 * m + i], &a); fe_add4(F, &gate, &gate, &t);            /* q_lc */
        fe_mul4(F, &t, &S[1 * m + i], &b); fe_add4(F, &gate, &gate, &t);
        fe_mul4(F, &t, &S[2 * m + i], &c); fe_add4(F, &gate, &gate, &t);
        fe_mul4(F, &t, &S[3 * m + i], &d); fe_add4(F, &gate, &gate, &t);
        fe_mul4(F, &t, &S[4 * m + i], &ab); fe_add4(F, &gate, &gate, &t);           /* q_mul */
        fe_mul4(F, &t, &S[5 * m + i], &cd); fe_add4(F, &gate, &gate, &t);
        fe_mul4(F, &t, &S[12 * m + i], &ab); fe_mul4(F, &t, &t, &cd); fe_mul4(F, &t, &t, &e); fe_add4(F, &gate, &gate, &t);  /* q_ecc*ab*cd*e */
        const fe4 *ws[4] = {&a, &b, &c, &d};
        for (int j = 0; j < 4; j++) {                                               /* q_hash[j] * w^5 */
            fe_sqr4(F, &p5, ws[j]); fe_sqr4(F, &p5, &p5); fe_mul4(F, &p5, &p5, ws[j]);
            fe_mul4(F, &t, &S[(6 + j) * m + i], &p5); fe_add4(F, &gate, &gate, &t);
        }
        fe_mul4(F, &t, &S[10 * m + i], &e); fe_sub4(F, &gate, &gate, &t);           /* - q_o*e */
        /* permutation check (:479-495) */
        fe4 acc1 = Z[i], acc2 = Z[(i + ratio) % m];
        for (int j = 0; j < 5; j++) {
            fe4 tj, u;
            fe_add4(F, &tj, &Wv[j * m + i], &gamma);
            fe_mul4(F, &u, &K[j], &x); fe_mul4(F, &u, &u, &beta); fe_add4(F, &u, &tj, &u); fe_mul4(F, &acc1, &acc1, &u);
            fe_mul4(F, &u, &G[j * m + i], &beta); fe_add4(F, &u, &tj, &u); fe_mul4(F, &acc2, &acc2, &u);
        }
        fe4 perm; fe_sub4(F, &perm, &acc1, &acc2); fe_mul4(F, &perm, &alpha, &perm);
        /* (z(x)-1) * alpha^2 / (n (x-1))  (:497-503) */
        fe4 l1, den; fe_sub4(F, &l1, &Z[i], &F->one); fe_mul4(F, &l1, &a2n, &l1);
        fe_sub4(F, &den, &x, &F->one); fe_inv4(F, &den, &den); fe_mul4(F, &l1, &l1, &den);
        fe4 r; fe_add4(F, &r, &gate, &perm); fe_mul4(F, &r, &zh_inv[i % ratio], &r); fe_add4(F, &r, &r, &l1);
        out[i] = r;
    }
    free(xs);
    return 0;
}

/* ---- permutation grand product, dispatcher2.rs:329-344.
 * wires[5][n] (= witness[wire_variables[i][j]]), id_perm[5n] (extended_id_permutation), perm_idx[5n] (perm_i*n+perm_j),
 * all Fr Montgomery; out[n], out[0] = 1.  Returns -2 on a zero denominator (the reference panics there). */
int orc_perm_product(int curve, size_t n, const uint64_t *wires_, const uint64_t *id_perm_, const uint64_t *perm_idx,
                     const uint64_t *beta_, const uint64_t *gamma_, uint64_t *out_) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr;
    const fe4 *W = (const fe4 *)wires_, *ID = (const fe4 *)id_perm_;
    const fe4 beta = *(const fe4 *)beta_, gamma = *(const fe4 *)gamma_;
    fe4 *out = (fe4 *)out_;
    fe4 zero; memset(&zero, 0, sizeof zero);
    out[0] = F->one;
    if (n < 2) return 0;
    /* a[j] / b[j] per gate; the divisions use Montgomery's batch-inversion trick per 1024-gate chunk (field
     * arithmetic is exact, so the quotients are the reference's a / b) */
    fe4 *A = (fe4 *)malloc(sizeof(fe4) * n), *Bv = (fe4 *)malloc(sizeof(fe4) * n);
    int bad = 0;
#pragma omp parallel for schedule(static)
    for (size_t j = 0; j < n - 1; j++) {
        fe4 a = F->one, b = F->one, t, u;
        for (int i = 0; i < 5; i++) {
            fe_add4(F, &t, &W[i * n + j], &gamma);
            fe_mul4(F, &u, &beta, &ID[i * n + j]); fe_add4(F, &u, &t, &u); fe_mul4(F, &a, &a, &u);
            fe_mul4(F, &u, &beta, &ID[perm_idx[i * n + j]]); fe_add4(F, &u, &t, &u); fe_mul4(F, &b, &b, &u);
        }
        A[j] = a; Bv[j] = b;
        if (!memcmp(&b, &zero, sizeof b)) {
#pragma omp atomic write
            bad = 1;
        }
    }
    if (bad) { free(A); free(Bv); return -2; }
    const size_t CH = 1024, nch = (n - 1 + CH - 1) / CH;
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < nch; c++) {
        size_t lo = c * CH, hi = lo + CH < n - 1 ? lo + CH : n - 1;
        fe4 pre[1024], run = F->one, inv, t;
        for (size_t j = lo; j < hi; j++) { pre[j - lo] = run; fe_mul4(F, &run, &run, &Bv[j]); }
        fe_inv4(F, &inv, &run);
        for (size_t j = hi; j-- > lo;) {
            fe_mul4(F, &t, &inv, &pre[j - lo]);      /* 1 / b[j] */
            fe_mul4(F, &inv, &inv, &Bv[j]);
            fe_mul4(F, &A[j], &A[j], &t);            /* a[j] / b[j] */
        }
    }
    for (size_t j = 0; j + 1 < n; j++) fe_mul4(F, &out[j + 1], &out[j], &A[j]);
    free(A); free(Bv);
    return 0;
}

/* ---- round 4/5 polynomial ops (coefficients Fr Montgomery) */
/* DensePolynomial::evaluate, dispatcher2.rs:545-555 */
int orc_poly_eval(int curve, const uint64_t *coeffs_, size_t len, const uint64_t *z_, uint64_t *out_) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr;
    const fe4 *c = (const fe4 *)coeffs_; const fe4 z = *(const fe4 *)z_;
    fe4 acc; memset(&acc, 0, sizeof acc);
    for (size_t i = len; i-- > 0;) { fe_mul4(F, &acc, &acc, &z); fe_add4(F, &acc, &acc, &c[i]); }
    memcpy(out_, &acc, 32);
    return 0;
}
/* out[i] = sum_k coeff[k] * polys[k][i]  (i < lens[k]); out_len = max lens.  dispatcher2.rs:566-633,646-649 */
int orc_poly_lincomb(int curve, size_t k, const uint64_t *const *polys, const size_t *lens, const uint64_t *coeffs_, uint64_t *out_, size_t out_len) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr;
    fe4 *out = (fe4 *)out_;
    memset(out, 0, 32 * out_len);
    for (size_t t = 0; t < k; t++) {
        const fe4 *q = (const fe4 *)polys[t]; const fe4 cf = ((const fe4 *)coeffs_)[t];
        size_t L = lens[t] < out_len ? lens[t] : out_len;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < L; i++) { fe4 u; fe_mul4(F, &u, &q[i], &cf); fe_add4(F, &out[i], &out[i], &u); }
    }
    return 0;
}
/* quotient of poly / (X - z), remainder dropped: dispatcher2.rs:651-666 (q_{i-1} = c_i + z q_i).  out: len-1 coefficients. */
int orc_poly_div_linear(int curve, const uint64_t *coeffs_, size_t len, const uint64_t *z_, uint64_t *out_) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr;
    const fe4 *c = (const fe4 *)coeffs_; const fe4 z = *(const fe4 *)z_;
    fe4 *q = (fe4 *)out_;
    if (len < 2) return 0;
    fe4 carry; memset(&carry, 0, sizeof carry);
    for (size_t i = len - 1; i >= 1; i--) {
        fe4 t; fe_mul4(F, &t, &carry, &z); fe_add4(F, &carry, &c[i], &t);
        q[i - 1] = carry;
    }
    return 0;
}
/* (sum_i b_i X^i)(X^n - 1) + poly, in place; poly has n + k coefficients.  dispatcher2.rs:311-312,347-348 */
int orc_blind(int curve, uint64_t *poly_, size_t n, const uint64_t *blinders_, size_t k) {
    curve_t *C = get_curve(curve); const fctx4 *F = &C->fr;
    fe4 *p = (fe4 *)poly_; const fe4 *b = (const fe4 *)blinders_;
    for (size_t i = 0; i < k; i++) { fe_sub4(F, &p[i], &p[i], &b[i]); fe_add4(F, &p[n + i], &p[n + i], &b[i]); }
    return 0;
}

int orc_max_threads(void) { return omp_get_max_threads(); }
