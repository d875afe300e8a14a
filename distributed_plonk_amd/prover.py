"""Host-side mirror of the reference prover's five rounds (`Prover::prove`, /root/reference/src/dispatcher2.rs:192-713)
with every vector resident in HBM: each step the reference runs on the dispatcher's CPU (or ships to workers) is one
call into libplonk_hip.so through PlonkWorker, and only commitments (affine points), evaluations and challenges cross
the host boundary.

Out of scope here: the merlin transcript.  Challenges come from a caller-supplied function
`challenge(label, proof_so_far) -> Fr limbs`, blinding polynomials from the caller (the reference draws them from `prng`).

MI355X-first: all 25 coset evaluation vectors of round 3 (25 x 8n x 32 B = 100 GiB at n = 2^24) stay on the device; the
18 that belong to the proving key (selectors, sigmas) can be cached across proofs (`cache_key_cosets`, 72 GiB at 2^24).
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Optional

import numpy as np

from . import fr as _fr
from .transcript import PlonkTranscript
from .worker import PlonkWorker

NUM_WIRE_TYPES = 5
NUM_SELECTORS = 13          # q_lc[4], q_mul[2], q_hash[4], q_o, q_c, q_ecc (dispatcher2.rs:443-456)


class WrongQuotientPolyDegree(Exception):        # SnarkError::WrongQuotientPolyDegree, dispatcher2.rs:513-517
    def __init__(self, got: int, expected: int):
        super().__init__(f"WrongQuotientPolyDegree({got}, {expected})")
        self.got, self.expected = got, expected


class FiatShamir:
    """Challenge source backed by the reference's transcript: appends what `Prover::prove` appends before each challenge
    (dispatcher2.rs:323, 327-328, 356, 361, 533, 543, 555, 634).  Use as the `challenge` argument of Prover.prove*."""

    def __init__(self, transcript: PlonkTranscript):
        self.t = transcript
        self.drawn: Dict[str, np.ndarray] = {}

    def __call__(self, label: str, proof: dict) -> np.ndarray:
        t = self.t
        if label == "beta":
            t.append_commitments(b"witness_poly_comms", proof["wires_poly_comms"])
        elif label == "alpha":
            t.append_commitment(b"perm_poly_comms", proof["prod_perm_poly_comm"])
        elif label == "zeta":
            t.append_commitments(b"quot_poly_comms", proof["split_quot_poly_comms"])
        elif label == "v":
            t.append_proof_evaluations(proof["wires_evals"], proof["wire_sigma_evals"], proof["perm_next_eval"])
        c = t.get_and_append_challenge(label.encode())
        self.drawn[label] = c
        return c


class Prover:
    """One GPU.  `worker.init(ck, n, 8n)` must have been called with the commit key padded to a multiple of 32 points as
    dispatcher2.rs:207-208 does.  The reference pads with `G1Affine::zero()` = (0, 1, infinity = true): pass such a key in the
    PLONK_BASES_ARK layout (the infinity flag travels), or, in the PLONK_BASES_XY layout, encode the padding points as x = y = 0 —
    XY has no flag byte, and (0, 1) there would be read as a finite (off-curve) point."""

    def __init__(self, worker: PlonkWorker, log_n: int, cache_key_cosets: bool = False, quotient_mode: str = "coset8n",
                 commit_helper: Optional[PlonkWorker] = None, fft_helper: Optional[PlonkWorker] = None):
        """commit_helper: a second context on the same GPU, `init`-ed with the same commit key: independent commitments of a
        round are then issued from two host threads on two streams (the sort / reduction phases and the wave tail of one MSM
        overlap the bucket accumulation of the other: ~9 % per commitment at 2^24 points).
        fft_helper (quotient_mode "coset8n" without cache_key_cosets only): another context on the same GPU, `init`-ed for the same
        domains.  18 of round 3's 25 coset FFTs are of proving-key polynomials and depend on nothing a proof draws; with a helper
        they are issued on its stream by a host thread of their own at the START of the proof, beside the transforms and
        commitments of rounds 1 and 2, and round 3 only joins them.  Same proof bytes.  Measured per proof, same lease
        (profiles/r05_opening_measurements.txt): 63.3 -> 57.6 ms at 2^20 BN254, 291.6 -> 282.0 ms at 2^22 BLS12-381.  The 18 * 8n
        evaluation vector (77 GB at 2^24) is then live from the START of the proof, beside the buffers of rounds 1 and 2, instead of
        from round 3 on: the peak of a proof does not change (round 3 holds it either way), only when it is reached.
        quotient_mode "coset8n": round 3 exactly as the reference does it (25 coset FFTs over the 8n-point domain, one coset
        iFFT).  "classes6": the same quotient polynomial from 6n evaluations — see _quotient_poly_classes."""
        self.w = worker
        self.f = _fr.FIELDS[worker.curve_name]
        self._gen_limbs = self.f.to_limbs(self.f.generator)          # Fr::multiplicative_generator(), the coset shift of the quotient domain
        self.log_n, self.n, self.m = log_n, 1 << log_n, 8 << log_n
        self.cache_key_cosets = cache_key_cosets
        if quotient_mode not in ("coset8n", "classes6"):
            raise ValueError(quotient_mode)
        if quotient_mode == "classes6" and self.n < 16:
            raise ValueError("classes6 needs n >= 16: below that 5n+7 >= 6n-1 and the degree check of dispatcher2.rs:511-518 is vacuous")
        self.quotient_mode = quotient_mode
        self.commit_helper = commit_helper
        self.fft_helper = fft_helper
        self._key_ffts = None                 # (thread, 18 device pointers, errors) while the key's coset FFTs run beside rounds 1-2
        self._cls = None
        self._key = None
        self._bufs = []
        self._ws: Dict[str, object] = {}      # named work buffers, allocated by the first proof and reused (hipMalloc of ~100 GiB takes seconds)
        self.timings: Dict[str, float] = {}
        self.last_polys: Dict[str, object] = {}

    # ------------------------------------------------------------------ device buffers
    def _alloc(self, n_fr: int):
        b = self.w.alloc(max(n_fr, 1) * 32)
        self._bufs.append(b)
        return b

    def _free(self, bufs):
        for b in bufs:
            if b in self._bufs:
                self._bufs.remove(b)
            b.free()

    def _work(self, name: str, n_fr: int):
        b = self._ws.get(name)
        if b is None or b.nbytes < max(n_fr, 1) * 32:
            if b is not None:
                self._free([b])
            b = self._ws[name] = self._alloc(n_fr)
        return b

    def close(self):
        self._free(list(self._bufs))
        self._ws.clear()
        self._key = None

    def load_key(self, selectors: np.ndarray, sigmas: np.ndarray, k: np.ndarray):
        """ProvingKey polynomials in coefficient form: selectors (13,n,4), sigmas (5,n,4); k = vk.k (5,4)."""
        n = self.n
        assert selectors.shape == (NUM_SELECTORS, n, 4) and sigmas.shape == (NUM_WIRE_TYPES, n, 4)
        sel = self._alloc(NUM_SELECTORS * n).upload(selectors)
        sig = self._alloc(NUM_WIRE_TYPES * n).upload(sigmas)
        self.load_key_dev([sel.ptr + i * n * 32 for i in range(NUM_SELECTORS)], [sig.ptr + i * n * 32 for i in range(NUM_WIRE_TYPES)], k)

    def load_key_dev(self, sel_ptrs, sig_ptrs, k: np.ndarray):
        """Same with the 13 + 5 coefficient vectors (n Fr each) already in HBM; the caller keeps them alive."""
        n, m = self.n, self.m
        k = np.ascontiguousarray(k, dtype=np.uint64)
        assert len(sel_ptrs) == NUM_SELECTORS and len(sig_ptrs) == NUM_WIRE_TYPES and k.shape == (NUM_WIRE_TYPES, 4)
        self._key = dict(sel=[int(x) for x in sel_ptrs], sig=[int(x) for x in sig_ptrs], k=k, cos=None, vk=None)
        if self.cache_key_cosets and self.quotient_mode == "classes6":
            cls = self._class_setup()
            cos = self._alloc(18 * cls["ncls"] * n)
            ptrs = [[cos.ptr + (j * cls["ncls"] + c) * n * 32 for c in range(cls["ncls"])] for j in range(18)]
            for j, src in enumerate(self._key["sel"] + self._key["sig"]):
                for c in range(cls["ncls"]):
                    self.w.coset_eval_dev(src, n, n, cls["shift"][c], ptrs[j][c])
            self._key["cos"] = ptrs
        elif self.cache_key_cosets:
            cos = self._alloc(18 * m)
            ptrs = [cos.ptr + j * m * 32 for j in range(18)]
            for j, src in enumerate(self._key["sel"] + self._key["sig"]):
                self._coset_fft(src, n, ptrs[j])
            self._key["cos"] = ptrs

    def verifying_key(self) -> dict:
        """The parts of jf-plonk's VerifyingKey the transcript absorbs (dispatcher2.rs:56-91): commitments of the 13 selector
        and 5 sigma polynomials (18 MSMs, computed once per key)."""
        key = self._key
        if key["vk"] is None:
            key["vk"] = dict(domain_size=self.n, k=key["k"], selector_comms=[self._commit(ptr, self.n) for ptr in key["sel"]],
                             sigma_comms=[self._commit(ptr, self.n) for ptr in key["sig"]])
        return key["vk"]

    def fiat_shamir(self, public_inputs: np.ndarray) -> FiatShamir:
        """A fresh transcript with the verifying key and the public inputs absorbed (dispatcher2.rs:238-241).
        public_inputs: (num_inputs, 4) — `circuit.public_input()`, NOT padded to n."""
        vk = self.verifying_key()
        t = PlonkTranscript(self.w.curve_name)
        pi = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
        t.append_vk_and_pub_input(vk["domain_size"], pi.shape[0], list(vk["k"]), vk["selector_comms"], vk["sigma_comms"], list(pi))
        return FiatShamir(t)

    # ------------------------------------------------------------------ building blocks
    def _coset_fft(self, d_src: int, length: int, d_dst: int):
        """Self::fft(.., quot_domain, coeffs, true, false, true): the coset FFT of the coefficients zero-padded to m
        (dispatcher2.rs:387-424, 746).  plonk_coset_eval_dev with shift = g is that transform without ever materialising the
        zeros: 8 interleaved n-point transforms of the n+2 / n+3 coefficients."""
        self.w.coset_eval_dev(d_src, length, self.m, self._gen_limbs, d_dst)

    def _commit(self, d_poly: int, length: int):
        """commit_polynomial (dispatcher2.rs:835-893) -> affine (xy limbs, is_infinity)."""
        return self.w.g1_to_affine(self.w.commit_dev(d_poly, length))

    def _commit_many(self, items):
        """The independent commitments of a round [(device pointer, length), ...] -> affine points, in order: ONE Pippenger problem
        per context (plonk_commit_many_dev: the polynomials become extra windows of the same sort / accumulation / reduction),
        the round split over two contexts when a commit_helper is set."""
        if len(items) < 2:
            return [self._commit(ptr, ln) for ptr, ln in items]
        if self.commit_helper is None:
            return [self.w.g1_to_affine(j) for j in self.w.commit_many_dev(items)]
        import threading
        lanes = [self.w, self.commit_helper]
        self.w.sync()                                   # the helper's stream reads what this context's stream wrote
        out = [None] * len(items)
        errs = []

        def run(lane):
            try:
                mine = list(range(lane, len(items), 2))
                for i, j in zip(mine, lanes[lane].commit_many_dev([items[i] for i in mine])):
                    out[i] = j
            except BaseException as ex:     # noqa: BLE001 - re-raised below
                errs.append(ex)

        th = [threading.Thread(target=run, args=(lane,)) for lane in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        return [self.w.g1_to_affine(j) for j in out]

    def _degree(self, d_poly: int, length: int) -> int:
        return self.w.poly_degree_dev(d_poly, length)

    def _interpolate_many(self, alloc, pairs):
        """domain.ifft of rounds 1-3 (dispatcher2.rs:300-309, 345-346, 426): pairs [(d_evals, d_coeffs)], n evaluations -> n coefficients
        each; the evaluations are not modified (ntt_dev consumes its input, hence the copy)."""
        n = self.n
        d_tmp = self._work("interp_tmp", n)              # ONE temporary for the three calls of a proof (rounds 1, 2, 3), not a numbered buffer per call
        for src, dst in pairs:
            self.w.memcpy_d2d(d_tmp.ptr, src, n * 32)
            self.w.ntt_dev(d_tmp.ptr, dst, n, True, False)

    def _perm_product(self, alloc, wev, d_id: int, d_idx: int, beta, gamma) -> int:
        """the product vector of dispatcher2.rs:329-344 -> device pointer to its n values"""
        d_prod = alloc(self.n)
        self.w.perm_product_dev(wev, d_id, d_idx, beta, gamma, self.n, d_prod.ptr)
        return d_prod.ptr

    def _download(self, d_ptr: int, n_fr: int) -> np.ndarray:
        out = np.empty((n_fr, 4), dtype=np.uint64)
        import ctypes as C
        from ._ffi import check
        check(self.w.lib.plonk_memcpy_d2h(self.w.ctx, out.ctypes.data_as(C.c_void_p), d_ptr, out.nbytes))
        return out

    def _evaluate_many(self, polys, points):
        """DensePolynomial::evaluate of round 4 (:545-555): polys [(device pointer, length)], points [Fr limbs] -> [Fr limbs]."""
        return [self.w.poly_eval_dev(ptr, ln, pt) for (ptr, ln), pt in zip(polys, points)]

    def _openings(self, alloc, lin_terms, lin_coeffs, batch_terms, batch_coeffs, perm_poly, zeta, zeta_w, PP: int, keep: bool) -> dict:
        """Round 5 after the challenges (:566-697): lin_poly = sum lin_coeffs * lin_terms, batch_poly = lin_poly + sum batch_coeffs *
        batch_terms, the two divisions by (X - zeta) / (X - zeta w) and their commitments.  -> {"comms": [opening, shifted_opening],
        "opening_poly" / "shifted_opening_poly": (ptr, len), and with keep: "lin_poly" / "batch_poly" as host arrays}."""
        w, f = self.w, self.f
        d_lin = alloc(PP)
        w.poly_lincomb_dev(lin_terms, f.vec_to_limbs(lin_coeffs), d_lin.ptr, PP)
        d_batch = alloc(PP)
        w.poly_lincomb_dev([(d_lin.ptr, PP)] + list(batch_terms), f.vec_to_limbs([1] + list(batch_coeffs)), d_batch.ptr, PP)
        d_wit = alloc(2 * PP)
        w.poly_div_linear_dev(d_batch.ptr, PP, zeta, d_wit.ptr)
        w.poly_div_linear_dev(perm_poly[0], perm_poly[1], zeta_w, d_wit.ptr + PP * 32)
        out = dict(comms=self._commit_many([(d_wit.ptr, PP - 1), (d_wit.ptr + PP * 32, PP - 1)]),
                   opening_poly=(d_wit.ptr, PP - 1), shifted_opening_poly=(d_wit.ptr + PP * 32, PP - 1))
        if keep:
            out["lin_poly"], out["batch_poly"] = d_lin.download((PP, 4)), d_batch.download((PP, 4))
        return out

    # ---- the quotient from 6n evaluations instead of 8n
    def _class_setup(self):
        """The quotient has degree 5n+7 < 6n, so its values on SIX of the eight cosets  h_s * H_n  (h_s = g * w_m^s) of the
        reference's 8n-point domain determine it.  Writing t(X) = sum_{u<6} X^{un} t_u(X), deg t_u < n: on coset s, X^n is the
        constant c_s = h_s^n, so the size-n interpolant on that coset is P_s = sum_u c_s^u t_u — a 6x6 Vandermonde system per
        coefficient index, inverted once on the host."""
        if self._cls is None:
            f, n, m = self.f, self.n, self.m
            p = f.p
            ncls = 6
            w_m = f.root_of_unity(m)
            h = [f.generator * pow(w_m, s, p) % p for s in range(ncls)]
            c = [pow(x, n, p) for x in h]
            # inverse of V[s][u] = c_s^u by Gauss-Jordan mod p
            V = [[pow(c[s], u, p) for u in range(ncls)] + [1 if s == t else 0 for t in range(ncls)] for s in range(ncls)]
            for col in range(ncls):
                piv = next(r for r in range(col, ncls) if V[r][col])
                V[col], V[piv] = V[piv], V[col]
                inv = pow(V[col][col], -1, p)
                V[col] = [x * inv % p for x in V[col]]
                for r in range(ncls):
                    if r != col and V[r][col]:
                        fac = V[r][col]
                        V[r] = [(x - fac * y) % p for x, y in zip(V[r], V[col])]
            vinv = [row[ncls:] for row in V]                                   # vinv[u][s]
            self._cls = dict(ncls=ncls, shift=[f.to_limbs(x) for x in h], one=f.to_limbs(1),
                             vinv=[f.vec_to_limbs(row) for row in vinv])
        return self._cls

    def _quotient_poly_classes(self, alloc, tick, wire_polys, perm_poly, pi_poly, alpha, beta, gamma) -> int:
        w, n, m, key = self.w, self.n, self.m, self._key
        cls = self._class_setup()
        C = cls["ncls"]
        t0 = time.perf_counter()
        if key["cos"] is None:
            d_kc = alloc(18 * C * n)
            kc = [[d_kc.ptr + (j * C + c) * n * 32 for c in range(C)] for j in range(18)]
            for j, src in enumerate(key["sel"] + key["sig"]):
                for c in range(C):
                    w.coset_eval_dev(src, n, n, cls["shift"][c], kc[j][c])
        else:
            kc = key["cos"]
        d_c = alloc(7 * C * n)
        cw = [[d_c.ptr + (j * C + c) * n * 32 for c in range(C)] for j in range(7)]
        for j, (ptr, ln) in enumerate(list(wire_polys) + [perm_poly, pi_poly]):
            for c in range(C):
                w.coset_eval_dev(ptr, ln, n, cls["shift"][c], cw[j][c])         # n+2 / n+3 coefficients fold onto n
        tick("round3_coset_ffts", t0)
        t0 = time.perf_counter()
        d_qev = alloc(n)
        d_P = alloc(C * n)
        for c in range(C):
            w.quotient_evals_dev([kc[j][c] for j in range(13)], [kc[13 + j][c] for j in range(5)], [cw[j][c] for j in range(5)], cw[5][c], cw[6][c],
                                 alpha, beta, gamma, key["k"], d_qev.ptr, class_stride=m // n, class_offset=c)
            w.coset_interp_dev(d_qev.ptr, n, cls["shift"][c], cls["one"], 0, n, d_P.ptr + c * n * 32)      # P_c = t mod (X^n - c_c)
        d_quot = alloc(m)
        w.memset_dev(d_quot.ptr + C * n * 32, 0, (m - C * n) * 32)
        terms = [(d_P.ptr + c * n * 32, n) for c in range(C)]
        for u in range(C):
            w.poly_lincomb_dev(terms, cls["vinv"][u], d_quot.ptr + u * n * 32, n)     # t_u = sum_s Vinv[u][s] P_s
        tick("round3_quotient", t0)
        return d_quot.ptr

    def _key_ffts_start(self, alloc):
        """fft_helper: the 18 key coset FFTs on the helper's stream, from a thread of their own (see __init__)"""
        if self.fft_helper is None or self.quotient_mode != "coset8n" or self._key["cos"] is not None:
            return
        import threading
        m, key, h, gen = self.m, self._key, self.fft_helper, self._gen_limbs
        d_kc = self._work("key_cosets", 18 * m)         # a name of its own: the numbered work buffers of the rounds do not shift with the mode
        kc = [d_kc.ptr + j * m * 32 for j in range(18)]
        self.w.sync()                                   # the helper's stream reads the key polynomials this context's stream wrote
        errs = []

        def run():
            try:
                for j, src in enumerate(key["sel"] + key["sig"]):
                    h.coset_eval_dev(src, self.n, m, gen, kc[j])
                h.sync()
            except BaseException as ex:     # noqa: BLE001 - re-raised by _key_ffts_join
                errs.append(ex)

        th = threading.Thread(target=run)
        th.start()
        self._key_ffts = (th, kc, errs)

    def _key_ffts_join(self):
        th, kc, errs = self._key_ffts
        self._key_ffts = None
        th.join()
        if errs:
            raise errs[0]
        return kc

    def _quotient_poly(self, alloc, tick, wire_polys, perm_poly, pi_poly, alpha, beta, gamma) -> int:
        """Round 3 between the challenges and the split commitments (dispatcher2.rs:362-509): 25 coset FFTs over the 8n domain,
        the pointwise quotient evaluation, one coset iFFT.  Returns a device pointer to the m quotient coefficients."""
        if self.quotient_mode == "classes6":
            return self._quotient_poly_classes(alloc, tick, wire_polys, perm_poly, pi_poly, alpha, beta, gamma)
        w, n, m, key = self.w, self.n, self.m, self._key
        t0 = time.perf_counter()
        if self._key_ffts is not None:
            kc = self._key_ffts_join()
        elif key["cos"] is None:
            d_kc = alloc(18 * m)
            kc = [d_kc.ptr + j * m * 32 for j in range(18)]
            for j, src in enumerate(key["sel"] + key["sig"]):
                self._coset_fft(src, n, kc[j])
        else:
            kc = key["cos"]
        d_c = alloc(7 * m)
        cw = [d_c.ptr + j * m * 32 for j in range(7)]
        for j, (ptr, ln) in enumerate(list(wire_polys) + [perm_poly, pi_poly]):
            self._coset_fft(ptr, ln, cw[j])                               # :406-429
        tick("round3_coset_ffts", t0)
        t0 = time.perf_counter()
        d_qev = alloc(m)
        w.quotient_evals_dev(kc[0:13], kc[13:18], cw[0:5], cw[5], cw[6], alpha, beta, gamma, key["k"], d_qev.ptr)
        d_quot = alloc(m)
        w.ntt_dev(d_qev.ptr, d_quot.ptr, m, True, True)                   # :507
        tick("round3_quotient", t0)
        return d_quot.ptr

    # ------------------------------------------------------------------ the five rounds
    def prove(self, wires: np.ndarray, id_perm: np.ndarray, perm_idx: np.ndarray, pub_input: np.ndarray, blinders: dict,
              challenge: Callable[[str, dict], np.ndarray], check_degree: bool = True, keep: bool = False) -> dict:
        """wires (5,n,4): witness[wire_variables[i][j]]; id_perm (5n,4): extended_id_permutation; perm_idx (5n,) u64:
        perm_i*n+perm_j; pub_input (n,4) evaluations (zero-padded); blinders {"wires": (5,2,4), "perm": (3,4)}.
        Returns the fields of `Proof` (dispatcher2.rs:699-710) with commitments as (xy, is_inf)."""
        n = self.n
        up = []
        try:
            d_wev = self._alloc(5 * n).upload(np.ascontiguousarray(wires, dtype=np.uint64)); up.append(d_wev)
            d_id = self._alloc(5 * n).upload(np.ascontiguousarray(id_perm, dtype=np.uint64)); up.append(d_id)
            d_idx = self._alloc((5 * n + 3) // 4).upload(np.ascontiguousarray(perm_idx, dtype=np.uint64)); up.append(d_idx)
            d_pi = self._alloc(n).upload(np.ascontiguousarray(pub_input, dtype=np.uint64)); up.append(d_pi)
            return self.prove_dev([d_wev.ptr + i * n * 32 for i in range(5)], d_id.ptr, d_idx.ptr, d_pi.ptr, blinders, challenge,
                                  check_degree=check_degree, keep=keep)
        finally:
            self._free(up)

    def prove_dev(self, wev, d_id: int, d_idx: int, d_pi: int, blinders: dict, challenge: Callable[[str, dict], np.ndarray],
                  check_degree: bool = True, keep: bool = False) -> dict:
        """Same as prove() with the circuit data already in HBM: wev = 5 device pointers (n Fr each), d_id (5n Fr),
        d_idx (5n u64), d_pi (n Fr); none of them is modified."""
        assert self._key is not None, "load_key first"
        w, f, n, m, key = self.w, self.f, self.n, self.m, self._key
        p = f.p
        L, I = f.to_limbs, f.from_limbs
        T = self.timings
        T.clear()
        proof: dict = {}
        counter = [0]

        def alloc(cnt):
            counter[0] += 1
            return self._work(f"prove{counter[0]}", cnt)

        def tick(name, t0):
            w.sync()
            T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

        self._key_ffts_start(alloc)
        try:
            return self._rounds(wev, d_id, d_idx, d_pi, blinders, challenge, check_degree, keep, alloc, tick, proof)
        finally:
            if self._key_ffts is not None:              # an exception before round 3: do not leave the helper's thread behind
                self._key_ffts[0].join()
                self._key_ffts = None

    def _rounds(self, wev, d_id, d_idx, d_pi, blinders, challenge, check_degree, keep, alloc, tick, proof) -> dict:
        w, f, n, m, key = self.w, self.f, self.n, self.m, self._key
        p = f.p
        L, I = f.to_limbs, f.from_limbs
        # ---- Round 1 (:296-322): wire polynomials and their commitments
        t0 = time.perf_counter()
        WP = n + 2
        d_wp = alloc(5 * WP)
        wp = [d_wp.ptr + i * WP * 32 for i in range(5)]
        for i in range(5):                                            # the interpolation writes coefficients 0 .. n-1; only the two above them must be zero
            w.memset_dev(wp[i] + n * 32, 0, (WP - n) * 32)            # before the blinding adds into them (a whole-vector memset was 2.5 GiB per 2^24 proof)
        self._interpolate_many(alloc, [(wev[i], wp[i]) for i in range(5)])
        for i in range(5):
            w.blind_dev(wp[i], n, blinders["wires"][i])
        proof["wires_poly_comms"] = self._commit_many([(wp[i], WP) for i in range(5)])
        tick("round1", t0)
        # ---- Round 2 (:325-357): permutation product polynomial
        t0 = time.perf_counter()
        beta, gamma = challenge("beta", proof), challenge("gamma", proof)
        d_prod_ptr = self._perm_product(alloc, wev, d_id, d_idx, beta, gamma)
        dbg_prod = self._download(d_prod_ptr, n) if keep else None
        PP = n + 3
        d_pp = alloc(PP)
        w.memset_dev(d_pp.ptr + n * 32, 0, (PP - n) * 32)
        self._interpolate_many(alloc, [(d_prod_ptr, d_pp.ptr)])
        w.blind_dev(d_pp.ptr, n, blinders["perm"])
        proof["prod_perm_poly_comm"] = self._commit(d_pp.ptr, PP)
        tick("round2", t0)
        # ---- Round 3 (:360-533): quotient polynomial
        t0 = time.perf_counter()
        alpha = challenge("alpha", proof)
        d_pi_poly = alloc(n)
        self._interpolate_many(alloc, [(d_pi, d_pi_poly.ptr)])            # :426
        d_quot_ptr = self._quotient_poly(alloc, tick, [(wp[i], WP) for i in range(5)], (d_pp.ptr, PP), (d_pi_poly.ptr, n), alpha, beta, gamma)
        t0 = time.perf_counter()
        expected = NUM_WIRE_TYPES * (n + 1) + 2
        if check_degree:
            deg = self._degree(d_quot_ptr, m)
            if deg != expected:
                raise WrongQuotientPolyDegree(deg, expected)
        split = []
        for off in range(0, expected + 1, n + 2):                       # coeffs.chunks(n + 2)  (:519-523)
            split.append((d_quot_ptr + off * 32, min(n + 2, expected + 1 - off)))
        proof["split_quot_poly_comms"] = self._commit_many(split)
        tick("round3_commit", t0)
        # ---- Round 4 (:536-555): evaluations at zeta
        t0 = time.perf_counter()
        zeta = challenge("zeta", proof)
        z = I(zeta)
        zeta_w = L(z * f.root_of_unity(n))
        polys4 = [(wp[i], WP) for i in range(5)] + [(key["sig"][i], n) for i in range(4)] + [(d_pp.ptr, PP)]
        ev = self._evaluate_many(polys4, [zeta] * 9 + [zeta_w])
        proof["wires_evals"], proof["wire_sigma_evals"], proof["perm_next_eval"] = ev[0:5], ev[5:9], ev[9]
        tick("round4", t0)
        # ---- Round 5 (:558-690): linearisation polynomial, batched opening, shifted opening
        t0 = time.perf_counter()
        al, be, ga = I(alpha), I(beta), I(gamma)
        a, b, c, d, e = (I(x) for x in proof["wires_evals"])
        sg = [I(x) for x in proof["wire_sigma_evals"]]
        kk = [I(x) for x in key["k"]]
        vanish = (pow(z, n, p) - 1) % p
        ab, cd = a * b % p, c * d % p
        polys = [(ptr, n) for ptr in key["sel"]]
        coeffs = [a, b, c, d, ab, cd, pow(a, 5, p), pow(b, 5, p), pow(c, 5, p), pow(d, 5, p), (-e) % p, 1, ab * cd % p * e % p]
        l1 = vanish * f.inv(n * (z - 1) % p) % p
        acc = al
        for wv, k_ in zip((a, b, c, d, e), kk):
            acc = acc * ((wv + be * k_ % p * z + ga) % p) % p
        polys.append((d_pp.ptr, PP))
        coeffs.append((acc + al * al % p * l1) % p)
        acc = al * be % p * I(proof["perm_next_eval"]) % p
        for wv, s in zip((a, b, c, d), sg):
            acc = acc * ((wv + be * s + ga) % p) % p
        polys.append((key["sig"][4], n))
        coeffs.append((-acc) % p)
        z_n2 = (vanish + 1) * z % p * z % p
        cq = 1
        for ptr, ln in split:
            polys.append((ptr, ln))
            coeffs.append((-vanish) * cq % p)
            cq = cq * z_n2 % p
        v = I(challenge("v", proof))
        # batch_poly = lin_poly + v w_0 + ... + v^5 w_4 + v^6 sigma_0 + ... + v^9 sigma_3 (:646-649), as ONE list of scalar * polynomial
        # terms: lin_poly's own terms first (coefficient v^0 = 1)
        bterms = [(wp[i], WP) for i in range(5)] + [(key["sig"][i], n) for i in range(4)]
        bcoef = [pow(v, i + 1, p) for i in range(len(bterms))]
        opening = self._openings(alloc, polys, coeffs, bterms, bcoef, (d_pp.ptr, PP), zeta, zeta_w, PP, keep)
        proof["opening_proof"], proof["shifted_opening_proof"] = opening["comms"]
        tick("round5", t0)
        # where the committed polynomials live until the next proof reuses the work buffers (post-hoc checks: bench.py re-derives
        # commitments and evaluations of a finished 2^24-gate proof from them with the CPU oracle)
        self.last_polys = dict(wire_polys=[(wp[i], WP) for i in range(5)], perm_poly=(d_pp.ptr, PP), split_quot_polys=list(split),
                               opening_poly=opening.get("opening_poly"), shifted_opening_poly=opening.get("shifted_opening_poly"))
        if keep:
            proof["_debug"] = dict(perm_product=dbg_prod, perm_poly=d_pp.download((PP, 4)),
                                   quot_poly=self._download(d_quot_ptr, expected + 1), lin_poly=opening["lin_poly"],
                                   batch_poly=opening["batch_poly"])
        return proof
