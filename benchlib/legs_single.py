"""The single-GPU run (rank 0, N == 1): the proof the headline times (SingleProof) and the legs AFTER and OUTSIDE the timed region — the oracle
as CHECKER of what was just timed, the SURVEY §8f "next" rows measured on their own, the timed proof handed to a verifier, its variants."""
import time

from .common import HBM_PEAK_GBS
from .legs_multi import TAU_SEED

HELPER_MAX_LOG_N = 22          # largest size at which the proof's key coset FFTs go to a third context by default (proof_helper_wanted)


def verify_single(b):
    """-> the `verification` dict (every value must be True).  The oracle is used here and only here: as the checker."""
    from oracle import checks, oracle as O
    args, np, w, n, m, nbig = b.args, b.np, b.w, b.n, b.m, b.nbig
    buf_n, buf_m, scal, polys = b.buf_n, b.buf_m, b.scal, b.polys
    verification = {}
    cid = O.CURVE_IDS[args.curve]
    f_ = __import__("distributed_plonk_amd.fr", fromlist=["FIELDS"]).FIELDS[args.curve]
    # (1) a whole ROUND of five commitments of the timed configuration (same bases, the step's first five DISTINCT scalar vectors,
    #     same two-lane code path: both contexts run a BATCHED problem, three and two vectors, at full size), each against
    #     the exact expected point from small oracle MSMs of its aggregated scalars (oracle/checks.py)
    got5 = b.commits_finish(b.commits_start(5), all_parts=True)
    ok = True
    for j, got in enumerate(got5):
        sc = O.from_mont(cid, scal[j % len(scal)].download((n, 4)))
        want = (checks.msm_expected_distinct(cid, 0x5EED, sc) if args.bases == "distinct"
                else checks.msm_expected_tiled(cid, 0x5EED, min(n, 1 << 11), sc))
        e_, ei = O.jac_to_affine(cid, want)
        g_, gi = w.g1_to_affine(got)
        ok &= bool(gi == ei and np.array_equal(g_, e_))
        del sc
    verification["commit_round_of_5_distinct_vectors_vs_oracle_exact"] = ok
    # (2) 8n coset FFTs as timed, three different polynomials of the step: sampled outputs against Horner evaluations by an
    #     unrelated kernel (plonk_poly_eval_dev, itself oracle-checked in tests/); for the last one the coset iFFT must also
    #     return the zero-padded coefficients everywhere
    CH = 1 << 22
    if not nbig:
        pass                                   # --n-domain-only: there is no 8n transform to check
    elif b.padded:
        w_m = f_.root_of_unity(m)
        ok = True
        picks = sorted({0, len(polys) // 2, len(polys) - 1})
        for pi_ in picks:
            buf_p = polys[pi_]
            w.coset_eval_dev(buf_p.ptr, b.poly_len, m, b.gen_limbs, buf_m[0][0].ptr)
            for k_ in (0, 1, 8, 9, (12345 + pi_) % m, (5 * n + 3) % m, m - 1):
                x_ = f_.to_limbs(f_.generator * pow(w_m, k_, f_.p) % f_.p)
                ok &= bool(np.array_equal(buf_m[0][0].download((1, 4), byte_offset=k_ * 32)[0], w.poly_eval_dev(buf_p.ptr, b.poly_len, x_)))
        verification["coset_fft_samples_vs_poly_eval_3_polys"] = ok
        w.ntt_dev(buf_m[0][0].ptr, buf_m[0][1].ptr, m, True, True)
        back = buf_m[0][1]
        ok = bool(np.array_equal(back.download((b.poly_len, 4)), buf_p.download((b.poly_len, 4))))
        for off in range(b.poly_len * 32, m * 32, CH * 32):
            nb = min(CH * 32, m * 32 - off)
            ok &= not back.download((nb // 8,), byte_offset=off).any()
        verification["coset_fft_round_trip_every_element"] = ok
    else:
        w.synth_fr(0xBADC0DE, buf_m[0][0].ptr, m)
        keep = w.alloc(m * 32)
        w.memcpy_d2d(keep.ptr, buf_m[0][0].ptr, m * 32)
        w.ntt_dev(buf_m[0][0].ptr, buf_m[0][1].ptr, m, False, True)
        w.ntt_dev(buf_m[0][1].ptr, buf_m[0][0].ptr, m, True, True)
        ok = True
        for off in range(0, m, CH):
            cnt = min(CH, m - off)
            ok &= bool(np.array_equal(buf_m[0][0].download((cnt, 4), byte_offset=off * 32), keep.download((cnt, 4), byte_offset=off * 32)))
        keep.free()
        verification["coset_fft_round_trip_every_element"] = ok
    # (3) a size-n iNTT as timed: NTT(iNTT(x)) == x everywhere (and against the oracle itself when n is small enough)
    w.synth_fr(0xD15EA5E, buf_n[0][0].ptr, n)
    ref = buf_n[0][0].download((n, 4))
    w.ntt_dev(buf_n[0][0].ptr, buf_n[0][1].ptr, n, True, False)
    w.ntt_dev(buf_n[0][1].ptr, buf_n[0][0].ptr, n, False, False)
    verification["intt_n_round_trip_every_element"] = bool(np.array_equal(buf_n[0][0].download((n, 4)), ref))
    if n <= (1 << 20):                     # small enough for the oracle to transform directly
        buf_n[0][0].upload(ref)
        w.ntt_dev(buf_n[0][0].ptr, buf_n[0][1].ptr, n, True, False)
        verification["intt_n_vs_oracle"] = bool(np.array_equal(buf_n[0][1].download((n, 4)), O.ntt(cid, ref, True, False, threads=O.max_threads())))
    return verification


def quotient_row(b):
    """next row, measured on its own, NOT part of `value`: quotient coset evaluations over 8n points"""
    np, w, m = b.np, b.w, b.m
    vecs = []
    try:
        for _ in range(25):              # inside the try: an allocation that fails half-way must not leak the buffers before it (100 GiB at 2^24)
            vecs.append(w.alloc(m * 32))
        for j, buf in enumerate(vecs):
            w.synth_fr(0xABC + j, buf.ptr, m)
        ch = np.arange(32, dtype=np.uint64).reshape(8, 4) + 3
        ptr = [buf.ptr for buf in vecs]
        w.profile_enable(True)
        for it in range(2):
            w.profile_reset()
            w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], b.buf_m[0][1].ptr)
            w.sync()
        qms, _ = w.profile_get("quotient_evals_kernel")
        w.profile_enable(False)
    finally:
        for buf in vecs:
            buf.free()
    alg = 27.0 * 32 * m                     # 26 vector reads (z twice) + 1 write per point
    return {"quotient_evals_kernel": {"points": m, "ms": round(qms, 3), "bound": "hbm", "achieved": round(alg / qms / 1e6, 1),
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / qms / 1e6 / HBM_PEAK_GBS, 4),
                                      "algorithmic_bytes": alg, "reference": "dispatcher2.rs:435-504"}}


def _check_proof(b, pv, inst, vk, pub, proof, fs, TAU):
    """the proof just timed, checked: (a) accepted by the pairing-free verifier for the trapdoor SRS (oracle/verifier_ref.py: pure-Python
    integers, its own Fiat-Shamir) — every one of the 13 + 18 commitments, the 10 evaluations and both openings enter that equation;
    (b) three of the proof's commitments re-derived as f(tau)*G and three evaluations re-derived by the CPU oracle's Horner from the
    polynomials the prover holds; (c) a flipped evaluation must be rejected."""
    from distributed_plonk_amd.transcript import PlonkTranscript
    from oracle import bigint_ref as B_, oracle as O, verifier_ref as V_
    args, np, n = b.args, b.np, b.n
    fld = __import__("distributed_plonk_amd.fr", fromlist=["FIELDS"]).FIELDS[args.curve]
    pver = {}
    cid = O.CURVE_IDS[args.curve]
    cv = B_.CURVES[args.curve]
    t0 = time.perf_counter()
    res = V_.verify(cv, vk, pub, proof, TAU, transcript=PlonkTranscript(args.curve))
    pver["accepted_by_verifier"] = True
    pver["verifier_and_prover_drew_the_same_challenges"] = all(np.array_equal(res["challenges"][k_], fs.drawn[k_]) for k_ in fs.drawn)
    bad = [x.copy() for x in proof["wires_evals"]]
    bad[1][0] ^= np.uint64(1)
    try:
        V_.verify(cv, vk, pub, dict(proof, wires_evals=bad), TAU, transcript=PlonkTranscript(args.curve))
        pver["flipped_evaluation_rejected"] = False
    except V_.VerificationError:
        pver["flipped_evaluation_rejected"] = True
    tau_l, zeta_l = fld.to_limbs(TAU), fs.drawn["zeta"]
    zeta_w = fld.to_limbs(fld.from_limbs(zeta_l) * fld.root_of_unity(n) % fld.p)
    lp = pv.last_polys
    ok_c = ok_e = True
    for (ptr, ln), comm, ev_pt, ev_want in ((lp["wire_polys"][2], proof["wires_poly_comms"][2], zeta_l, proof["wires_evals"][2]),
                                            (lp["perm_poly"], proof["prod_perm_poly_comm"], zeta_w, proof["perm_next_eval"]),
                                            (lp["split_quot_polys"][4], proof["split_quot_poly_comms"][4], None, None),
                                            ((inst.sig_ptrs[1], n), vk["sigma_comms"][1], zeta_l, proof["wire_sigma_evals"][1])):
        poly = pv._download(ptr, ln)
        f_tau = O.from_mont(cid, O.poly_eval(cid, poly, tau_l).reshape(1, 4))[0]
        want = O.jac_to_affine(cid, O.scalar_mul(cid, O.generator(cid), f_tau))
        ok_c &= bool(want[1] == comm[1] and np.array_equal(want[0], comm[0]))
        if ev_pt is not None:
            ok_e &= bool(np.array_equal(O.poly_eval(cid, poly, ev_pt), ev_want))
        del poly
    pver["commitments_equal_f_of_tau_times_G_by_cpu_horner_4_checked"] = ok_c
    pver["evaluations_equal_cpu_horner_3_checked"] = ok_e
    pver["check_s"] = round(time.perf_counter() - t0, 1)
    return pver


def _small_rows(b, inst, fs, consts):
    """the O(n) rows on their own (HIP events inside the library), with their algorithmic HBM bytes"""
    np, w, n = b.np, b.w, b.n
    ch = {k_: fs.drawn[k_] for k_ in ("beta", "gamma", "alpha", "zeta", "v")}
    w.profile_enable(True)
    w.profile_reset()
    out_n = w.alloc((n + 3) * 32)
    w.perm_product_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, ch["beta"], ch["gamma"], n, out_n.ptr)
    w.poly_eval_dev(inst.wev[0], n, ch["zeta"])
    w.poly_lincomb_dev([(ptr_, n) for ptr_ in inst.sel_ptrs + inst.sig_ptrs] + [(inst.wev[0], n), (inst.wev[1], n)], np.tile(consts[:4], (5, 1)), out_n.ptr, n)
    w.poly_div_linear_dev(inst.wev[0], n, ch["zeta"], out_n.ptr)
    w.sync()

    def row(names, alg_bytes, ref):
        ms = sum(w.profile_get(k_)[0] for k_ in names)
        return {"ms": round(ms, 3), "bound": "hbm", "achieved": round(alg_bytes / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg_bytes / ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes": alg_bytes, "reference": ref}

    small = {
        "perm_product": row(["perm_terms_kernel", "perm_scan_num", "perm_scan_den_final"], n * (16 * 32 + 5 * 8.0), "dispatcher2.rs:329-344"),
        "poly_eval": row(["poly_eval_kernel"], n * 32.0, "dispatcher2.rs:545-555"),
        "poly_lincomb_20_terms": row(["poly_lincomb_kernel"], n * 21 * 32.0, "dispatcher2.rs:566-633"),
        "poly_div_linear": row(["poly_div_kernels"], n * 64.0, "dispatcher2.rs:651-666"),
    }
    w.profile_enable(False)
    out_n.free()
    return small


def proof_helper_wanted(args) -> bool:
    """Prover(fft_helper=...): the 18 proving-key coset FFTs of round 3 on a third context beside rounds 1 and 2 (same proof bytes).  Same-lease
    A/Bs (profiles/r05_opening_measurements.txt, r05_helper_2p24.txt): 2^20 BN254 63.3 -> 57.6 ms per proof, 2^22 BLS12-381 291.6 -> 282.0 ms.
    PLONK_BENCH_PROOF_HELPER=0 / 1 overrides (A/B runs)."""
    import os
    return os.environ.get("PLONK_BENCH_PROOF_HELPER", "1" if args.log_n <= HELPER_MAX_LOG_N else "0") == "1"


class _Helper:
    """the third context: the step's transform context when the run has one (--overlap-phases), else a temporary one"""

    def __init__(self, b, inst):
        from distributed_plonk_amd.worker import PlonkWorker
        self.own = None
        if b.wt is not b.w:
            self.ctx = b.wt
            return
        self.own = self.ctx = PlonkWorker(me=b.rank, device=b.local_rank, curve=b.args.curve)
        self.ctx.init_dev(inst.d_ck.ptr, inst.key_size, b.n, 8 * b.n)
        self.ctx.sync()

    def close(self):
        if self.own is not None:
            self.own.close()


def _variants(b, inst, vk, pub, bl, proof, full=True):
    """variants of the same rounds (identical proofs).  `full`: the quotient from 6 cosets of H_n instead of the 8n-point domain, and/or the 18
    proving-key evaluation vectors kept resident across proofs (72 / 54 GiB at 2^24).  PLONK_BENCH_HELPER_AB=1 adds the headline's rounds with the
    third context switched the other way (A/B runs of proof_helper_wanted's choice)."""
    from distributed_plonk_amd.prover import Prover
    np, w, n = b.np, b.w, b.n
    variants = {}
    same_as_headline_proof = lambda pr: bool(all(np.array_equal(pr[k_][0], proof[k_][0]) for k_ in ("opening_proof", "shifted_opening_proof"))
                                             and np.array_equal(np.stack(pr["wires_evals"]), np.stack(proof["wires_evals"])))
    helper = None

    def fft_helper():
        nonlocal helper
        helper = _Helper(b, inst)
        return helper.ctx

    import os
    todo = []
    if os.environ.get("PLONK_BENCH_HELPER_AB") == "1":
        todo.append(("key_coset_ffts_inside_round_3", lambda: {}) if proof_helper_wanted(b.args) else
                    ("key_coset_ffts_beside_rounds_1_2", lambda: dict(fft_helper=fft_helper())))
    if full:
        todo += [("resident_key_cosets", lambda: dict(cache_key_cosets=True)),
                 ("six_cosets", lambda: dict(quotient_mode="classes6")),
                 ("six_cosets_resident_key", lambda: dict(quotient_mode="classes6", cache_key_cosets=True))]
    for vname, kw_of in todo:
        try:
            pvc = Prover(w, b.args.log_n, commit_helper=b.workers[1], **kw_of())
            pvc.load_key_dev(inst.sel_ptrs, inst.sig_ptrs, inst.k)
            pvc._key["vk"] = vk                                          # same key: the 18 commitments are not repeated
            t_v = pr = None
            for it in range(2):
                fsv = pvc.fiat_shamir(pub)
                t0 = time.perf_counter()
                pr = pvc.prove_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, inst.d_pi.ptr, bl, fsv, check_degree=True)
                t_v = (time.perf_counter() - t0) * 1e3
            variants[vname] = {"ms": round(t_v, 2), "constraints_per_s": round(n / t_v * 1e3, 1),
                               "rounds_ms": {k_: round(v_, 2) for k_, v_ in pvc.timings.items()},
                               "same_proof_as_the_verified_one": same_as_headline_proof(pr)}
            pvc.close()
        except Exception as ex:     # noqa: BLE001 - a variant is a side note of a side leg
            variants[vname] = {"error": str(ex)}
    if helper is not None:
        helper.close()
    return variants


class SingleProof:
    """The run's REAL proof (N == 1): a random SATISFIED 2^log_n-gate TurboPlonk circuit generated in HBM, its proving key, a trapdoor SRS, and
    the prover of distributed_plonk_amd/prover.py.  `prove()` is one pass of `Prover::prove` (dispatcher2.rs:296-712): transcript set-up with
    the verifying key and the public inputs (:238-241), rounds 1-5 with the merlin challenges, quotient-degree check ON.  bench.py times W + K
    of them as the headline and hands the last one to the verifier AFTER the timed region (`check`)."""

    def __init__(self, b):
        from distributed_plonk_amd.prover import Prover
        from distributed_plonk_amd.synthetic import SyntheticInstance
        args, np, w, workers = b.args, b.np, b.w, b.workers
        self.b = b
        fld = __import__("distributed_plonk_amd.fr", fromlist=["FIELDS"]).FIELDS[args.curve]
        self.TAU = TAU_SEED % fld.p      # the trapdoor this run publishes
        t0 = time.perf_counter()
        self.inst = inst = SyntheticInstance(w, args.log_n, seed=0xC1AC, num_inputs=3, tau=self.TAU, helpers=workers[1:2])
        for x in workers[:2]:
            x.sync()
        self.t_gen = (time.perf_counter() - t0) * 1e3
        self.consts = consts = np.arange(64, dtype=np.uint64).reshape(16, 4) + 11
        self.bl = {"wires": consts[5:15].reshape(5, 2, 4), "perm": consts[12:15]}
        self.helper = _Helper(b, inst) if proof_helper_wanted(args) else None
        self.pv = Prover(w, args.log_n, commit_helper=workers[1], fft_helper=self.helper.ctx if self.helper else None)
        self.pv.load_key_dev(inst.sel_ptrs, inst.sig_ptrs, inst.k)
        self.pub = inst.public_inputs()
        t0 = time.perf_counter()
        self.vk = self.pv.verifying_key()                                           # preprocess: 18 commitments, once per key
        self.t_vk = (time.perf_counter() - t0) * 1e3
        self.proof = self.fs = None
        self.prove()                     # set-up, never timed: the first proof allocates the work buffers (hipMalloc of ~120 GiB at 2^24 takes seconds)

    def prove(self):
        pv, inst = self.pv, self.inst
        self.fs = pv.fiat_shamir(self.pub)
        self.proof = pv.prove_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, inst.d_pi.ptr, self.bl, self.fs, check_degree=True)
        return pv.timings

    @property
    def overlapped(self):
        return self.helper is not None

    def check(self):
        b = self.b
        if b.args.no_verify:
            return None, {}
        try:
            pver = _check_proof(b, self.pv, self.inst, self.vk, self.pub, self.proof, self.fs, self.TAU)
        except Exception as ex:     # noqa: BLE001 - a failed check must be visible, never fatal
            pver = {"error": repr(ex)}
        return bool(pver) and "error" not in pver and all(v_ for k_, v_ in pver.items() if k_ != "check_s"), pver

    def close_prover(self):
        if self.pv is not None:
            self.pv.close()
            self.pv = None
        if self.helper is not None:
            self.helper.close()
            self.helper = None

    def close(self):
        self.close_prover()
        if self.inst is not None:
            self.inst.close()
            self.inst = None


def proof_rows(b, P, proof_ms, rounds_ms, with_small_rows=True, with_variants=True, with_quotient_row=False):
    """After the timed proofs: the last proof handed to the verifier (the reference's own end-to-end test, dispatcher2.rs:1273-1295, at the run's
    size), the O(n) rows measured on their own, the quotient kernel on its own, the same-proof variants.
    -> (rows dict, prover_rounds dict)"""
    n = b.n
    prover_verified, pver = P.check()
    small = _small_rows(b, P.inst, P.fs, P.consts) if with_small_rows else {}
    helper_on = P.helper is not None
    proof, inst, vk, pub, bl = P.proof, P.inst, P.vk, P.pub, P.bl
    P.close_prover()                     # its work buffers (~125 GiB at 2^24) go before the quotient row (100 GiB) and the variants allocate theirs
    if with_quotient_row:
        try:
            small.update(quotient_row(b))
        except Exception as ex:     # noqa: BLE001 - a row of its own: its failure must not cost the proof's verdict
            small["quotient_evals_kernel"] = {"error": str(ex)}
    variants = _variants(b, inst, vk, pub, bl, proof, full=with_variants)
    row = {
        "n": n, "ms": None if proof_ms is None else round(proof_ms, 2), "constraints_per_s": None if proof_ms is None else round(n / proof_ms * 1e3, 1),
        "rounds_ms": rounds_ms,
        "prover_verified": prover_verified, "prover_verification": pver,
        "key_coset_ffts": "on a third context beside rounds 1 and 2" if helper_on else "inside round 3",
        "setup_ms": {"circuit_key_and_trapdoor_srs_generation": round(P.t_gen, 1), "verifying_key_18_commitments": round(P.t_vk, 1)},
        "variants": variants,
        "reference": "dispatcher2.rs:296-712 (rounds 1-5: 13 commitments, 7 NTT(n), 26 NTT(8n), permutation product, quotient, "
                     "10 evaluations, linearisation, 2 openings), end-to-end test dispatcher2.rs:1273-1295",
        "note": "a random SATISFIED TurboPlonk circuit generated in HBM (plonk_synth_circuit: uniform witness and selectors, q_c solved per "
                "gate, copy constraints = n cycles of length 5 between pseudo-random gates), commit key tau^i*G with a published trapdoor "
                "(plonk_synth_srs), challenges from the merlin transcript (host Python, ~7 ms inside the timed proof), "
                "WrongQuotientPolyDegree check ON.  `ms` = the run's headline (the average of the timed proofs).  Variants produce the same proof: "
                "resident_key_cosets skips the 18 selector/sigma coset NTTs per proof (proving-key data); six_cosets interpolates the degree-(5n+7) "
                "quotient from 6n evaluations"}
    return small, row
