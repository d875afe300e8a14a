/* plonk_hip.h — C ABI of the MI355X-native MSM + NTT hot path (libplonk_hip.so).
 *
 * Drop-in boundary for MengLing-L/distributed_plonk's worker: one entry point per Cap'n Proto method
 * of `interface PlonkSlave` @0..@6 and `interface PlonkPeer` @0 (reference
 * src/hello_world.capnp:16-24,49-50; server src/worker.rs:125-439; clients src/dispatcher.rs:50-175,
 * src/dispatcher2.rs:961-1086), plus the three third-party operator calls the worker makes
 * (VariableBaseMSM::multi_scalar_mul, Radix2EvaluationDomain::{fft,ifft}_in_place, Fr::pow loops), and — further down — the
 * O(n) loops the dispatcher runs itself in `Prover::prove` (src/dispatcher2.rs:329-344, 435-504, 545-688: SURVEY.md §8f) as
 * device-resident calls, with the coset-class primitives a multi-rank prover is built from.
 * Plain pointers and sizes only; the caller owns every host buffer; the library owns device memory
 * inside the context.  INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * Byte layouts are the reference's raw memory layouts (src/utils.rs:27-43):
 *   Fr            4 x u64 little-endian, Montgomery form (R = 2^256), fully reduced
 *   MSM scalar    4 x u64 little-endian, canonical (`into_repr()`)
 *   Fq            4 x u64 (BN254) / 6 x u64 (BLS12-381), Montgomery
 *   G1 projective X || Y || Z (Jacobian, Montgomery), infinity = Z == 0
 *   G1 affine     PLONK_BASES_XY: x || y, infinity encoded as x = y = 0 (not on either curve)
 *                 PLONK_BASES_ARK: arkworks' in-memory `GroupAffine {x, y, infinity: bool}` with the
 *                 struct padded to 8 bytes (72 B BN254 / 104 B BLS12-381) — what `init` puts on the wire
 *
 * Every function returns 0 (PLONK_OK) or a negative error code; nothing throws across the boundary.
 * The reference panics (`.unwrap()`) on malformed input (worker.rs:131,253,303); here ranges, sizes
 * and domain two-adicity are validated and reported.  A context is bound to one GPU and is not
 * thread-safe (the reference worker is a single-threaded tokio LocalSet, worker.rs:441-448); DIFFERENT contexts may be
 * driven from different host threads concurrently (two contexts on one GPU overlap independent commitments).
 */
#ifndef PLONK_HIP_H
#define PLONK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plonk_ctx plonk_ctx;

enum { PLONK_BN254 = 0, PLONK_BLS12_381 = 1 };
enum { PLONK_BASES_XY = 0, PLONK_BASES_ARK = 1 };
enum {
    PLONK_OK = 0,
    PLONK_ERR_ARG = -1,       /* bad argument (range, size, null) */
    PLONK_ERR_DOMAIN = -2,    /* DomainCreationError: log2(size) > two-adicity, or not a power of two */
    PLONK_ERR_HIP = -3,       /* HIP runtime error (no device, out of memory, launch failure) */
    PLONK_ERR_STATE = -4,     /* call order violated (unknown task id, rows missing, ...) */
    PLONK_ERR_EXCHANGE = -5   /* the exchange callback failed */
};

/* src/utils.rs:3-8 / hello_world.capnp:8-13 */
typedef struct { uint64_t row_start, row_end, col_start, col_end; } plonk_fft_workload;
/* src/utils.rs:21-25 / hello_world.capnp:3-6 */
typedef struct { uint64_t start, end; } plonk_msm_workload;

/* Worker<->worker transport for fft2Prepare (replaces the per-exchange TCP + Cap'n Proto
 * connections of worker.rs:303-338).  Called once per fft2_prepare with device pointers:
 * `send` holds n_ranks blocks of `bytes_per_peer` bytes (block p goes to rank p), `recv` receives
 * n_ranks blocks (block p came from rank p).  `stream` is the hipStream_t the buffers are ordered on.
 * An RCCL all-to-all (ncclSend/ncclRecv group, or torch.distributed.all_to_all_single) implements it.
 * Return 0 on success. */
typedef int (*plonk_exchange_fn)(void* user, const void* send, void* recv, size_t bytes_per_peer,
                                 int n_ranks, void* stream);

/* ---- in-library transport: RCCL over xGMI (replaces the worker<->worker TCP + Cap'n Proto links of worker.rs:280-345,412-438
 * and the `result: Data` replies the dispatcher adds up, dispatcher.rs:236-238).  One process per GPU.  Rank 0 obtains an id with
 * plonk_comm_unique_id and ships its 128 bytes to the other ranks out of band (the reference's config/network.json + TCP play
 * that role); every rank then calls plonk_comm_init — a collective: it returns when all `world` ranks have joined.  With a
 * communicator attached, plonk_fft2_prepare(ctx, id, NULL, NULL) performs the all-to-all itself (grouped ncclSend/ncclRecv on the
 * context's stream); the callback form stays for tests and for hosts that bring their own transport.  Two contexts of one
 * process (two streams) need two communicators, created in the same order on every rank, and every rank must issue its
 * collectives in the same order.  The library chains every collective it issues on a GPU behind the previous one (an event wait on
 * the issuing stream), whatever communicator it belongs to: collectives of different communicators are never in flight together
 * (RCCL gives no progress guarantee for that), they still overlap the other contexts' compute. */
#define PLONK_COMM_ID_BYTES 128
int plonk_comm_unique_id(void* out_id);
int plonk_comm_init(plonk_ctx* ctx, const void* id, int rank, int world);
int plonk_comm_destroy(plonk_ctx* ctx);
/* rank / world as RCCL reports them (ncclCommUserRank / ncclCommCount), and the RCCL version code (ncclGetVersion) */
int plonk_comm_info(plonk_ctx* ctx, int* rank, int* world, int* rccl_version);
/* plonk_exchange_fn backed by the context's communicator: pass as `exchange` with user = ctx (what a NULL callback selects) */
int plonk_exchange_rccl(void* user, const void* send, void* recv, size_t bytes_per_peer, int n_ranks, void* stream);
/* diagnostic plonk_exchange_fn (bench.py --simulate-ranks: one rank's share of an n_ranks job on ONE GPU): blocks 1 .. n_ranks-1 of
 * `send` are copied to the same blocks of `recv` on `stream`, device to device — the bytes a real exchange would move, through the
 * same buffers, at HBM instead of xGMI speed.  Results are meaningless; `user` is ignored. */
int plonk_exchange_standin(void* user, const void* send, void* recv, size_t bytes_per_peer, int n_ranks, void* stream);
/* device buffers, ordered on the context's stream, not synchronised: block p of d_send -> rank p / every rank's d_send -> block
 * `rank` of everybody's d_recv.  The two data-path collectives of the coset-class prover.
 * THREAD RULE (all collectives of this library, plonk_fft2_prepare's exchange included): the collectives of one DEVICE must come from
 * ONE host thread — the first that issues one on that device owns it until the device's last communicator is destroyed; a call from
 * another thread returns PLONK_ERR_STATE instead of risking a cross-rank deadlock (two threads cannot promise the same issue order on
 * every rank).  A host whose runtime migrates one logical task between OS threads sets the environment variable
 * PLONK_COMM_ANY_THREAD=1 and serialises its collectives itself.
 * ORDER CHECK (diagnostic): with PLONK_COMM_CHECK_ORDER=1 in every rank's environment (or plonk_set_option(ctx, "comm_check_order", 1) on
 * every rank: the switch is process-wide) each collective is preceded by an all-gather of a 24-byte tag — the communicator's ordinal among
 * the device's communicators in creation order, the kind of collective, its byte count, the device's collective count — over the device's
 * FIRST communicator, and every rank returns PLONK_ERR_STATE when the tags differ: ranks that enter their collectives in different orders
 * (two tasks finished in different orders, a different kind or size on one rank) get an error naming both sides instead of a silent
 * deadlock inside RCCL.  Costs a host round trip per collective: for the first runs on a new machine, not for production. */
int plonk_comm_alltoall_dev(plonk_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_peer);
int plonk_comm_allgather_dev(plonk_ctx* ctx, const void* d_send, void* d_recv, size_t bytes);
/* host buffers (partial commitment points, 96 / 144 B each): out receives world * bytes.  Synchronises. */
int plonk_comm_allgather_host(plonk_ctx* ctx, const void* in, size_t bytes, void* out);

/* ---- lifetime ------------------------------------------------------------------------------- */
/* State::new, worker.rs:455-472. */
int plonk_create(plonk_ctx** out, int device, int curve);
void plonk_destroy(plonk_ctx* ctx);
const char* plonk_last_error(void);
/* HIP stream all work of this context is ordered on (for callers that share device buffers). */
void* plonk_stream(plonk_ctx* ctx);
int plonk_sync(plonk_ctx* ctx);

/* ---- PlonkSlave @0 init — worker.rs:126-157, client dispatcher.rs:50-68 ----------------------
 * Stores the SRS bases on the device and fixes the two evaluation domains (n and the quotient
 * domain m); either size may be 0 (dispatcher.rs:215 passes 0,0 for the MSM test).
 * The bases are CHECKED: every one must be a point of the curve (coordinates below the modulus, y^2 = x^3 + b), the infinity this
 * layout encodes ((0, 0) in PLONK_BASES_XY; any x, y with the flag byte = 1 in PLONK_BASES_ARK), and an ark flag byte must be 0 or 1 —
 * else PLONK_ERR_ARG naming the first offending index, and the context is left WITHOUT bases.  The reference reinterprets
 * `[G1Affine]` memory (utils.rs:27-43, worker.rs:136-141) and `GroupAffine {x, y, infinity}` is a default-repr Rust struct whose field
 * order is not guaranteed: swapped coordinates, a shifted stride or different padding fail here instead of yielding commitments that are
 * garbage with PLONK_OK.  ~1 ms per 2^24 points; plonk_set_option(ctx, "check_bases", 0) skips it. */
int plonk_init(plonk_ctx* ctx, const void* bases, size_t n_bases, int base_layout,
               size_t domain_size, size_t quot_domain_size);

/* ---- PlonkSlave @1 varMsm — worker.rs:159-185, client dispatcher.rs:70-92 --------------------
 * out_jacobian = sum_i scalars[i] * bases[start + i], i < min(end - start, n_scalars). */
int plonk_var_msm(plonk_ctx* ctx, const plonk_msm_workload* workload, const uint64_t* scalars,
                  size_t n_scalars, uint64_t* out_jacobian);

/* ---- PlonkSlave @2 fftInit — worker.rs:187-233 ----------------------------------------------- */
int plonk_fft_init(plonk_ctx* ctx, uint64_t id, const plonk_fft_workload* workloads, size_t n_workloads,
                   size_t me, int is_quot, int is_inv, int is_coset);
/* ---- PlonkSlave @3 fft1 — worker.rs:235-278 (helper :66-94).  One decimated row of c elements;
 * i is the LOCAL row index (global row = i + workloads[me].row_start, worker.rs:267). */
int plonk_fft1(plonk_ctx* ctx, uint64_t id, uint64_t i, const uint64_t* v, size_t len);
/* ---- PlonkSlave @4 fft2Prepare + PlonkPeer @0 fftExchange — worker.rs:280-345, 412-438 --------
 * Row pass on every local row, then the all-to-all block transpose.  Collective: every rank calls
 * it for the same id.  Transport precedence: a non-NULL `exchange` callback; else the context's RCCL communicator (plonk_comm_init)
 * when the task has exactly as many workloads as the communicator has ranks; a single-workload task stays local (nothing to
 * exchange) whether or not the context joined a communicator; anything else is PLONK_ERR_ARG. */
int plonk_fft2_prepare(plonk_ctx* ctx, uint64_t id, plonk_exchange_fn exchange, void* user);
/* ---- PlonkSlave @5 fft2 — worker.rs:347-381 (helper :96-115).  out_cols receives num_cols
 * columns of r elements each (column-major blobs, as the reply of worker.rs:366-376); the task is
 * removed (worker.rs:378). */
int plonk_fft2(plonk_ctx* ctx, uint64_t id, uint64_t* out_cols);
/* ---- PlonkSlave @6 round1 — worker.rs:383-408.  evals: n = domain_size Fr; blinders: the two
 * coefficients (b0, b1) of the degree-1 blinding polynomial the reference draws from thread_rng
 * (worker.rs:400), supplied explicitly so results are reproducible.  The blinded polynomial
 * (n + 2 coefficients) stays resident (`state.wire`, worker.rs:58); its commitment is returned. */
int plonk_round1(plonk_ctx* ctx, const uint64_t* evals, size_t n, const uint64_t* blinders,
                 uint64_t* out_commit_jacobian);
int plonk_get_wire(plonk_ctx* ctx, uint64_t* out_coeffs, size_t n_coeffs);

/* ---- third-party operator boundary (ark-poly / ark-ec calls the worker makes) ----------------
 * Whole-vector transform, dispatcher.rs:594,632,667 / dispatcher2.rs:507: host buffer of n = 2^k
 * Fr transformed in place ({fft,ifft,coset_fft,coset_ifft}_in_place). */
int plonk_ntt(plonk_ctx* ctx, uint64_t* v, size_t n, int is_inv, int is_coset);
/* commit_polynomial, worker.rs:117-123 / dispatcher2.rs:835-893: Montgomery coefficients ->
 * into_repr -> zero-pad to the SRS -> MSM.  Host buffers. */
int plonk_commit(plonk_ctx* ctx, const uint64_t* coeffs_mont, size_t n_coeffs, uint64_t* out_jacobian);
/* G1Projective addition / normalisation used by the dispatcher's reduce (dispatcher.rs:236-238) and
 * `Commitment(commitment.into())` (dispatcher2.rs:892).  Host-side, tiny. */
int plonk_g1_add(int curve, const uint64_t* a_jac, const uint64_t* b_jac, uint64_t* out_jac);
int plonk_g1_to_affine(int curve, const uint64_t* jac, uint64_t* out_xy, int* is_infinity);
/* Keccak-f[1600] in place on a 200-byte state (lane (x, y) at byte 8 * (x + 5 y), little-endian): the permutation under merlin / STROBE-128,
 * i.e. under the reference's Fiat-Shamir transcript (dispatcher2.rs:44-154; merlin 3.0.0, Cargo.toml:40).  Host-side, tiny: a host whose language
 * has no fast Keccak (the Python mirror) runs its transcript's ~25 permutations per proof through this. */
int plonk_keccak_f1600(uint8_t* state200);
/* ip_transpose, transpose.rs:413 (rows x cols -> cols x rows of Fr), host buffer. */
int plonk_transpose(plonk_ctx* ctx, uint64_t* v, size_t rows, size_t cols);

/* ---- device-resident variants (no host staging; pointers are HBM addresses on ctx's device) ---
 * d_in is used as workspace and destroyed unless d_in == d_out (then an internal scratch is used). */
int plonk_ntt_dev(plonk_ctx* ctx, void* d_in, void* d_out, size_t n, int is_inv, int is_coset);
int plonk_msm_dev(plonk_ctx* ctx, size_t start, size_t end, const void* d_scalars, uint64_t* out_jacobian);
int plonk_commit_dev(plonk_ctx* ctx, const void* d_coeffs_mont, size_t n_coeffs, uint64_t* out_jacobian);
/* The shard [start, start + count) of commit_polynomial (dispatcher2.rs:870-890 hands each worker such a range): d_coeffs_mont points
 * at coefficient `start`; out = sum_{i < count} into_repr(coeff[start + i]) * bases[start + i].  The shards' points add up to the
 * commitment. */
int plonk_commit_range_dev(plonk_ctx* ctx, const void* d_coeffs_mont, size_t start, size_t count, uint64_t* out_jacobian);
/* k commit_polynomial calls against the same key in ONE set of kernel launches — the independent commitments of a prover round
 * (dispatcher2.rs:313-321: five wire polynomials; :519-531: five quotient parts; :690-697: two opening proofs).  Polynomial i has
 * n_coeffs[i] Montgomery coefficients at d_coeffs_mont[i] (point at coefficient `start`), paired with bases [start, start + n_coeffs[i]);
 * out_jacobians receives k Jacobian triples.  Same points as k calls of plonk_commit_range_dev(.., start, n_coeffs[i], ..): the scalar
 * vectors become extra windows of one Pippenger problem (one sort, one bucket accumulation, one reduction), which removes the
 * per-MSM launch gaps, wave tails and host round trips that bound small MSMs. */
int plonk_commit_many_dev(plonk_ctx* ctx, size_t k, const void* const* d_coeffs_mont, const size_t* n_coeffs, size_t start, uint64_t* out_jacobians);
/* All local rows at once, row-major [num_rows][c] in HBM (replaces num_rows fft1 calls).  d_rows is CONSUMED: the buffer must stay
 * alive until plonk_fft2_prepare returns, which runs the row pass with d_rows as its inter-pass workspace (c > 2^9) and leaves
 * garbage in it — like plonk_ntt_dev's d_in. */
int plonk_fft1_dev(plonk_ctx* ctx, uint64_t id, void* d_rows);
/* The same for the rows of a ZERO-PADDED vector (the reference pads n + 2 / n + 3 coefficients to the 8n-point domain before it
 * decimates, dispatcher2.rs:746, 754): d_rows is [num_rows][row_len] — the leading row_len coefficients of every decimated row, the
 * rest of each row being zero by construction and never materialised (row b of the padded 8n vector has c/8 leading entries, one
 * more for b < 3).  Forward transforms only; the row pass then runs as 2^k independent (c/2^k)-point transforms per row.  The
 * buffer is NOT modified and must stay alive until plonk_fft2_prepare returns. */
int plonk_fft1_dev_compact(plonk_ctx* ctx, uint64_t id, const void* d_rows, size_t row_len);
/* Result of fft2 left in HBM.  layout 0: [num_cols][r] (the reference's reply); layout 1:
 * [r][num_cols] (natural order restricted to this rank's columns: element (j, i) = X[(i + col_start) + j*c]). */
int plonk_fft2_dev(plonk_ctx* ctx, uint64_t id, void* d_out, int layout);
int plonk_transpose_dev(plonk_ctx* ctx, const void* d_in, void* d_out, size_t rows, size_t cols);

/* ---- next row: quotient polynomial coset evaluations — dispatcher2.rs:362-504 --------------
 * Device pointers to the m = quot_domain_size coset evaluations the prover obtained from its 25 coset-FFTs
 * (dispatcher2.rs:382-432), in the selector order of :443-456: q_lc[0..3], q_mul[0..1], q_hash[0..3], q_o, q_c, q_ecc. */
typedef struct {
    const void* selectors[13];
    const void* sigmas[5];
    const void* wires[5];
    const void* perm;        /* permutation product polynomial z */
    const void* pub_input;
} plonk_quotient_inputs;
/* d_out[i] = 1/Z_H(x_i) * (gate(x_i) + alpha * perm(x_i)) + alpha^2/n * (z(x_i) - 1)/(x_i - 1), x_i = g * w_m^i, exactly as
 * the loop at dispatcher2.rs:435-504.  alpha, beta, gamma: transcript challenges; k: vk.k[0..5] — all Fr, Montgomery,
 * host pointers.  Uses the domains fixed by plonk_init.  The quotient's coefficient form is then
 * plonk_ntt_dev(d_out, ..., m, is_inv = 1, is_coset = 1) (dispatcher2.rs:507).
 * NO ALIASING: d_out must not overlap any input vector (a lane reads `perm` at its own index and at the shifted index of z(wX), and the
 * split formulation writes d_out before it reads wires, sigmas and perm) — an overlapping call is PLONK_ERR_ARG. */
int plonk_quotient_evals_dev(plonk_ctx* ctx, const plonk_quotient_inputs* in, const uint64_t* alpha, const uint64_t* beta,
                             const uint64_t* gamma, const uint64_t* k, void* d_out);

/* The same on ONE coset class: all inputs hold the m/G evaluations at the points x_j, j = class_offset + class_stride * k
 * (plonk_coset_eval_dev with shift g * w_m^class_offset), d_out[k] is the quotient evaluation at that point.  G = class_stride
 * must be a power of two dividing m/n, so z(w x) — point j + m/n — stays inside the class: rank-local with no halo. */
int plonk_quotient_evals_class_dev(plonk_ctx* ctx, const plonk_quotient_inputs* in, const uint64_t* alpha, const uint64_t* beta,
                                   const uint64_t* gamma, const uint64_t* k, uint32_t class_stride, uint32_t class_offset, void* d_out);

/* ---- next row: permutation grand product — dispatcher2.rs:329-344 ---------------------------
 * d_out[0] = 1, d_out[j+1] = d_out[j] * prod_i (w_i[j] + gamma + beta*id[i*n+j]) / prod_i (w_i[j] + gamma + beta*id[perm[i*n+j]]),
 * j < n-1: the `product_vec` the reference builds gate by gate on the host.  d_wires[i]: n wire values
 * witness[wire_variables[i][j]] (Fr, Montgomery); d_id_perm: the 5n values of extended_id_permutation; d_perm_idx: 5n u64,
 * perm_i * n + perm_j of wire_permutation[i*n+j]; beta, gamma: host.  PLONK_ERR_ARG if an index is >= 5n or a denominator
 * is zero (the reference panics: Fp division unwraps the inverse).  Synchronises the context's stream. */
int plonk_perm_product_dev(plonk_ctx* ctx, const void* const d_wires[5], const void* d_id_perm, const void* d_perm_idx,
                           const uint64_t* beta, const uint64_t* gamma, size_t n, void* d_out);
/* A worker's slice of that vector, up to the product of the gates before it: d_out[t] = prod over gates first <= j < first + t of the
 * same ratio, t < count (first + count <= n; gate n-1 contributes 1 as in the reference's loop, which stops at n-2).  With
 * count = slice + 1 the last value is the slice's total: G workers exchange those 32-byte totals, and worker s multiplies its slice by the
 * totals of the workers before it (class_prover.py) — the reference's serial dispatcher loop distributed like its FFTs
 * (dispatcher2.rs:329-344 next to :732-787).  The pointers are to the WHOLE vectors (n / 5n entries).  Same errors; synchronises. */
int plonk_perm_product_range_dev(plonk_ctx* ctx, const void* const d_wires[5], const void* d_id_perm, const void* d_perm_idx,
                                 const uint64_t* beta, const uint64_t* gamma, size_t n, size_t first, size_t count, void* d_out);

/* ---- next row: round 4/5 polynomial operations — dispatcher2.rs:545-555,566-633,646-688 ------
 * Coefficient vectors are device pointers to Fr (Montgomery); scalars (points, coefficients, blinders) are host Fr. */
/* out = poly(point): DensePolynomial::evaluate (:545-555).  len < 2^30.  Synchronises. */
int plonk_poly_eval_dev(plonk_ctx* ctx, const void* d_poly, size_t len, const uint64_t* point, uint64_t* out);
/* d_out[i] = sum_{t<k} coeffs[t] * d_polys[t][i] over the terms with i < lens[t], i < out_len (k <= 32): the scalar*poly sums
 * that build lin_poly and batch_poly (:566-633,646-649).  d_out must not alias an input. */
int plonk_poly_lincomb_dev(plonk_ctx* ctx, size_t k, const void* const* d_polys, const size_t* lens, const uint64_t* coeffs,
                           void* d_out, size_t out_len);
/* d_out[0..len-1) = quotient of poly / (X - point), remainder dropped: the synthetic-division loops of :651-666,672-688. */
int plonk_poly_div_linear_dev(plonk_ctx* ctx, const void* d_poly, size_t len, const uint64_t* point, void* d_out);
/* *degree = index of the highest non-zero coefficient, -1 for the zero polynomial: DensePolynomial::degree() after the
 * trimming of from_coefficients_vec — the WrongQuotientPolyDegree check of :511-518 without a host copy.  Synchronises. */
int plonk_poly_degree_dev(plonk_ctx* ctx, const void* d_poly, size_t len, int64_t* degree);
/* d_poly (n + k coefficients, the top k already valid, normally zero) += (sum_{i<k} blinders[i] X^i) * (X^n - 1):
 * DensePolynomial::rand(k-1).mul_by_vanishing_poly(domain) + poly (:311-312 k = 2, :347-348 k = 3).  k <= 4, any n >= 1 (a domain smaller
 * than the mask included). */
int plonk_blind_dev(plonk_ctx* ctx, void* d_poly, size_t n, const uint64_t* blinders, size_t k);

/* ---- evaluation on / interpolation from an ARBITRARY coset (building block of coset-class parallelism) ------
 * d_out[k] = poly(shift * w_size^k), k < size; size a power of two, len <= 8*size (coefficients beyond `size` fold back, since
 * X^size = shift^size on the coset).  shift = Fr::multiplicative_generator(), size = m: quot_domain.coset_fft (dispatcher2.rs:387-424)
 * of the zero-padded vector.  shift = g * w_m^s, size = m/G: the evaluations at the points of index s, s+G, s+2G, ... of that
 * same coset FFT — rank s's share of every round-3 vector with no communication.
 * Zero-padding aware: the `size - len` zero coefficients the reference appends (dispatcher2.rs:746) are never loaded, multiplied or
 * stored — with len <= size/2 the transform runs as 2^k independent (size/2^k)-point transforms of the same coefficients on the
 * sub-cosets (shift * w_size^q) * <w_(size/2^k)> whose results are interleaved into natural order by the last pass' stores.
 * d_poly is not modified; d_out must not alias it. */
int plonk_coset_eval_dev(plonk_ctx* ctx, const void* d_poly, size_t len, size_t size, const uint64_t* shift, void* d_out);
/* E = iNTT_size(d_evals) (d_evals is destroyed), d_out[t] = scale * shift^-(i0+t) * E[(i0+t) mod size], t < count.
 * shift = g, scale = 1, i0 = 0, count = size: quot_domain.coset_ifft (dispatcher2.rs:507).  With scale = 1/G and shift = g * w_m^s,
 * the sum over s < G of these vectors is coefficient i0+t of the polynomial interpolating all G cosets. */
int plonk_coset_interp_dev(plonk_ctx* ctx, void* d_evals, size_t size, const uint64_t* shift, const uint64_t* scale, size_t i0, size_t count,
                           void* d_out);

/* `classes` vectors of `size` values (d_in: class-major, class s at d_in + s * in_stride; in_stride = 0: size) -> natural order:
 * d_out[t * classes + s] = scale * d_in[s * in_stride + t'], t' = t, or (size - t) mod size with reverse != 0; scale = NULL: 1.
 * classes in {1, 2, 4, 8}; not in place.  What the dispatcher does
 * with the column replies of a distributed transform (dispatcher2.rs:776-787: concatenate + transpose), for transforms split by
 * residue class: a size-n iFFT over G workers is, on worker s, plonk_coset_eval_dev(evaluations, n, n / G, shift = w_n^-s) — the
 * evaluations read as coefficients, folded G-fold onto n / G points — then an all-gather, then this call with reverse = 1,
 * scale = 1 / n: coefficient s + G t of the interpolant is 1/n times value (n/G - t) mod n/G of class s. */
int plonk_class_interleave_dev(plonk_ctx* ctx, const void* d_in, size_t classes, size_t size, size_t in_stride, int reverse, const uint64_t* scale,
                               void* d_out);

/* ---- device memory + synthetic inputs (bench / tests; the reference uses thread_rng) ---------- */
/* Give back every cache the context can rebuild on demand (exchange buffers of finished FFT tasks, NTT factor planes and tables derived per
 * problem size, MSM workspace, scratch): a worker's State outlives a circuit (worker.rs:42-59) and would otherwise keep the last problem's
 * tens of GiB.  The SRS, the domains, open FFT tasks and the communicator stay.  Synchronises the context's stream. */
int plonk_trim(plonk_ctx* ctx);
int plonk_dev_alloc(plonk_ctx* ctx, size_t bytes, void** out);
int plonk_dev_free(plonk_ctx* ctx, void* p);
int plonk_memcpy_h2d(plonk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int plonk_memcpy_d2h(plonk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int plonk_memcpy_d2d(plonk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
int plonk_memcpy_d2d_async(plonk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);   /* ordered on the context's stream, not synchronised */
int plonk_memset_dev(plonk_ctx* ctx, void* d_dst, int byte, size_t bytes);
/* n uniform Fr (Montgomery limbs drawn like ark-ff's Fp::rand: mask + rejection), element i from
 * stream (seed, i). */
int plonk_synth_fr(plonk_ctx* ctx, uint64_t seed, void* d_out, size_t n);
/* SRS-like bases in PLONK_BASES_XY layout.  unique > 0: `unique` points k_j*G tiled to n (the
 * distribution of dispatcher.rs:190-196); unique == 0: n pairwise-distinct points A[i%4096] + B[i/4096]. */
int plonk_synth_bases(plonk_ctx* ctx, uint64_t seed, size_t unique, size_t n, void* d_out);
/* A KZG commit key with a KNOWN trapdoor: d_out[i] = tau^i * G, i < n (PLONK_BASES_XY).  The reference's universal_setup
 * (dispatcher2.rs:1278) draws tau and forgets it; keeping it makes commit(f) = f(tau) * G, so a proof can be verified in G1 without
 * a pairing (oracle/verifier_ref.py) — how bench.py checks a whole 2^24-gate proof.  tau: host Fr, Montgomery. */
int plonk_synth_srs(plonk_ctx* ctx, const uint64_t* tau, size_t n, void* d_out);
/* A random SATISFIED TurboPlonk instance of n = 2^k gates, generated in HBM (the reference's generate_circuit, dispatcher2.rs:1214-1270,
 * needs jellyfish's circuit builder; north_star asks for synthetic random circuits).  Column i of gate j reads variable P_i(j) for
 * seeded bijections P_i, so the copy constraints are n cycles of length 5 between pseudo-random gates; witness values are uniform
 * per variable; 12 selectors are uniform and q_c is solved so that the gate equation of dispatcher2.rs:465-477 holds; the first
 * num_inputs gates carry uniform public inputs.  Outputs (device): wires [5][n] Fr, selector EVALUATIONS [13][n] and sigma
 * EVALUATIONS [5][n] on the n-point domain (coefficient form = plonk_ntt_dev(.., is_inv = 1)), id_perm [5n] = k_i * w^j
 * (extended_id_permutation), perm_idx [5n] u64 (perm_i * n + perm_j), pub_input [n].  k: vk.k[0..5], host Fr. */
int plonk_synth_circuit(plonk_ctx* ctx, uint64_t seed, size_t n, size_t num_inputs, const uint64_t* k, void* d_wires, void* d_selector_evals,
                        void* d_sigma_evals, void* d_id_perm, void* d_perm_idx, void* d_pub_input);
/* Use n_bases points already in HBM (PLONK_BASES_XY) as the SRS without a host round trip; they are
 * re-encoded into the library's resident form (x || y as canonical R'-Montgomery residues in 32-bit words: 64 B / 96 B per point), the caller keeps its buffer. */
int plonk_init_dev(plonk_ctx* ctx, const void* d_bases_xy, size_t n_bases, size_t domain_size,
                   size_t quot_domain_size);
/* element-wise field ops on the device, for pinning the arithmetic layer.
 * field: 0 Fr, 1 Fq.  op: 0 mul, 1 add, 2 sub, 3 to_mont, 4 from_mont, 5 inverse, 6 square.  Host buffers. */
int plonk_debug_field_op(plonk_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b,
                         uint64_t* out, size_t n);
/* tuning knobs, each stored in the context it is set on (other contexts — also ones driven from other host threads — keep theirs):
 * key "msm_window" (bits, 0 = auto), "ntt_max_log_r" (<= 9), "msm_slice_log" (8..26: MSMs above 2^value points are
 * computed slice by slice and the partial points added; default 26, for tests of the slicing path), "msm_batch_max"
 * (scalar vectors per launch set of plonk_commit_many_dev, default 32; 1 = one MSM at a time), "msm_acc_persist" (workgroups per CU of the
 * persistent bucket accumulation, default 4; 0 = one lane per bucket over the whole grid; < 0 = an absolute grid, for tests),
 * "msm_precompute" (fixed-base window table built at the next init: 0 off = default, 1 when the cost model predicts a gain, 2 whenever a usable
 * shape exists — a pinned width that is unusable for the SRS at hand falls back to no table, never to an error),
 * "msm_table_c" (0 or 4..21) / "msm_table_sets" / "msm_table_budget_mib" (the table's window width, bucket sets per scalar and memory budget;
 * 0 = the plan's choice), "msm_sort_stage_cap" (tests: caps the LDS staging buffer of the level-2 sort), "msm_reduce_grid" (default 0: the window
 * reduction as tree sums over the bucket grid instead of the running-sum pyramid — faster for one small MSM alone, not beside another context's
 * accumulation: profiles/r04_pin_nop_experiment.txt), "msm_fused_order" (the bucket-size histogram inside the level-2 sort: 1 = for launches of
 * >= 2^23 points (default), 2 = always, 0 = never), "msm_fused_y3", "ntt_shoup" (both curves, default 1: precomputed-quotient butterflies in
 * the NTT passes; 0 = Montgomery butterflies), "check_bases" (default 1: plonk_init* verifies that every base is a curve point), "quotient_fuse" (6 = default: the compact kernel; 0-5, 7: other formulations).
 * INTEGRATION.md §6 has the table. */
int plonk_set_option(plonk_ctx* ctx, const char* key, int64_t value);
/* Timing of the kernels launched by the last plonk_*_dev call on this context, measured with HIP
 * events on the context's stream (milliseconds). */
int plonk_last_kernel_ms(plonk_ctx* ctx, double* out_ms);

/* Per-kernel timing with HIP events recorded on the context's stream around every kernel launch
 * (off by default).  Names: "ntt_pass_kernel", "ntt_pass_kernel<9>" (per in-LDS size),
 * "msm_digits_kernel", "msm_sort", "msm_bucket_order", "msm_accumulate_kernel", "msm_accumulate_redo_kernel", "msm_heavy",
 * "msm_reduce", "quotient_evals_kernel", "perm_terms_kernel", "perm_scan_num", "perm_scan_den_final",
 * "poly_eval_kernel", "poly_lincomb_kernel", "poly_scale_kernel", "poly_div_scan", "rccl_alltoall", "rccl_allgather" (the collective
 * as the stream saw it, waiting for peers included).  total_ms / launches accumulate until reset. */
int plonk_profile_enable(plonk_ctx* ctx, int on);
int plonk_profile_reset(plonk_ctx* ctx);
int plonk_profile_get(plonk_ctx* ctx, const char* name, double* total_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* PLONK_HIP_H */
