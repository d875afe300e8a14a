"""The program of one bench.py run, in order: the op-mix step, the proof loop that takes the headline over, the legs that check and extend the
line, the CPU baseline last.  One function per stage; every stage after the first measurement runs through run_leg (an exception costs the
stage's fields, never the line) and, where a collective could hang, under the line's watchdog."""
import os

from . import headline, legs_multi as M
from .line import ResultLine, run_leg

OPMIX_STEPS, OPMIX_WARMUP = 5, 2      # the op-mix step beside a proof headline: a short measurement of its own (`op_mix`), not the timed K steps


def _single_gpu_legs(b, out, args, P, proof_ms, rounds_ms):
    """rank 0, N == 1, after the timed region: the next rows, the timed proof handed to the verifier, its variants"""
    from . import legs_single as L
    next_rows = {}
    if P is None and args.next_rows == "all":
        next_rows.update(run_leg(None, "quotient_row", None, lambda: L.quotient_row(b), error=lambda ex: {"quotient_evals_kernel": {"error": str(ex)}}))
    if P is not None:
        full = args.next_rows == "all"
        res = run_leg(None, "prover_rounds", None, lambda: L.proof_rows(b, P, proof_ms, rounds_ms, with_small_rows=full, with_variants=full, with_quotient_row=full),
                      error=lambda ex: ({}, {"error": repr(ex)}))
        next_rows.update(res[0])
        next_rows["prover_rounds"] = res[1]
        run_leg(None, "proof_close", None, P.close)
    return next_rows or None


def _proof_fields(out, next_rows, class_row):
    """the REAL proof at top level (BASELINE's metric is "proof-gen ms"): the five rounds of dispatcher2.rs:296-712 on the 8n route
    with the proving key NOT resident — the reference's work — on the satisfied synthetic circuit, verified.  Since round 6 it IS the
    headline (`value` = n / proof_ms); the SURVEY §8d op mix of rounds 1-5 is `op_mix`.  Same-proof variants beside it, labelled."""
    pr = (next_rows or {}).get("prover_rounds") or {}
    if "ms" in pr and pr["ms"] is not None:
        out["proof_ms"] = pr["ms"]
        out["proof_constraints_per_s"] = pr["constraints_per_s"]
    if "prover_verified" in pr:
        out["prover_verified"] = pr.get("prover_verified")
        names = {"resident_key_cosets": "8n_route_key_coset_vectors_resident_in_HBM", "six_cosets": "six_coset_quotient_key_not_resident",
                 "six_cosets_resident_key": "six_coset_quotient_key_resident",
                 "key_coset_ffts_beside_rounds_1_2": "8n_route_key_not_resident_key_coset_ffts_on_a_third_context_beside_rounds_1_2",
                 "key_coset_ffts_inside_round_3": "8n_route_key_not_resident_key_coset_ffts_inside_round_3"}
        out["proof_variants_ms"] = dict({"8n_route_key_not_resident (the reference's work; = proof_ms; key coset FFTs %s)" % pr.get("key_coset_ffts", "inside round 3"): pr.get("ms")},
                                        **{names[k_]: v_.get("ms") for k_, v_ in (pr.get("variants") or {}).items()})
    if class_row and class_row.get("ms") is not None:
        out["proof_ms_class_prover_all_ranks"] = class_row["ms"]



def run(args, json_fd):
    from .run import Bench
    b = Bench(args)
    guard = ResultLine(json_fd, b.rank, None)
    rank0 = b.rank == 0
    multi, sim, nbig = b.multi, b.sim, b.nbig
    single = b.world == 1 and not multi and not sim
    with_class = bool((args.class_prover or multi or sim) and not args.no_class_prover and nbig)
    # The headline is the PROOF: K verified five-round proofs between barriers — the single-GPU prover at N == 1, the coset-class
    # prover on all ranks at N > 1.  The op-mix step of rounds 1-5 is measured first, as a short run of its own (`op_mix`); it is also what the line
    # falls back to — labelled — should the proof loop fail or, on N > 1, hang in a collective (the watchdog then prints what exists).
    proof_headline = args.headline == "proof" and nbig > 0 and (single or (with_class and (multi or sim)))
    om_steps, om_warm = (min(args.steps, OPMIX_STEPS), min(args.warmup, OPMIX_WARMUP)) if proof_headline else (args.steps, args.warmup)
    dt, phases_ms = headline.timed_steps(b, guard, om_steps, om_warm)
    # ---- the result line exists from here on: everything below ADDS fields to it (or, the proof loop, takes the headline fields over), each leg in
    # its own try/except; on N > 1 (never run on more than one real GPU: gpurun grants one) also under a watchdog: should a leg hang in a
    # collective, rank 0 still prints what it has (with `aborted_optional_leg` naming the leg) and every rank exits 0.
    out = headline.result_line(b, dt, phases_ms, om_steps, om_warm)
    guard.arm(None, 0)
    guard.out = out
    if rank0:
        out["headline"] = ("op-mix step (SURVEY §8d)" if not proof_headline else
                           "op-mix step (SURVEY §8d) — FALLBACK: the proof loop has not completed (see proof_headline_error / aborted_optional_leg)")
    LEG = float(os.environ.get("PLONK_BENCH_LEG_BUDGET_S", "300"))

    if single and rank0 and not args.no_verify:
        # the op-mix step just timed, checked against the oracle — BEFORE the proof's set-up replaces the contexts' seeded SRS by its trapdoor key
        from . import legs_single as L0
        ver = run_leg(None, "verify", None, lambda: L0.verify_single(b))       # a failed check must be visible, never fatal to the measurement
        out["verification"] = ver
        out["verified"] = "error" not in ver and all(ver.values())

    # ---- N == 1: the proof loop (set-up, W untimed + K timed proofs) right after the op-mix step
    P, proof_ms, rounds_ms = None, None, None
    if single and nbig and (proof_headline or args.next_rows in ("all", "proof")):
        from . import legs_single as L

        def run_proofs():
            nonlocal P
            P = L.SingleProof(b)
            k_, w_ = (args.steps, args.warmup) if proof_headline else (1, 0)
            return headline.timed_proofs(b, guard, P, k_, w_) + (k_, w_)
        res = run_leg(None, "proof_headline", None, run_proofs, error=lambda ex: {"error": repr(ex)})
        if isinstance(res, dict):
            out["proof_headline_error"] = res["error"]
            if P is not None:
                run_leg(None, "proof_close", None, P.close)
                P = None
        else:
            dt_p, rounds_ms, kern_p, k_, w_ = res
            proof_ms = dt_p / k_ * 1e3
            if proof_headline:
                headline.promote_proof(b, out, dt_p, rounds_ms, kern_p, k_, w_,
                                       "K verified five-round proofs of a satisfied synthetic circuit on one GPU (prover.py; dispatcher2.rs:192-713)", P.overlapped)
        if b.overlap or (P is not None and P.overlapped):
            un = run_leg(None, "roofline_unoverlapped", None, lambda: headline.unoverlapped_roofline(b))
            out["roofline_unoverlapped"] = ({"ran": "error" not in un, **({"error": un["error"]} if "error" in un else {})} if b.emulated else un)

    other = None
    if multi and not sim and nbig:
        other = run_leg(guard, "other_scheme", LEG, lambda: M.other_scheme(b))
        if rank0:
            out["other_scheme"] = other
    if multi and nbig and not args.no_poly_parallel:
        pp = run_leg(guard, "polynomial_parallel", LEG, lambda: M.polynomial_parallel(b), error=lambda ex: {"scheme": "polynomial_parallel", "error": repr(ex)})
        if rank0:
            out["polynomial_parallel"] = pp
            if not sim:
                # the three ways to spread the op-mix step over the ranks, side by side (the first is the reference's scheme)
                out["schemes_ms_per_step"] = {b.scheme: out.get("ms_per_step"), **({other["scheme"]: other.get("ms_per_step")} if other and "scheme" in other else {}),
                                              "polynomial_parallel": pp.get("ms_per_step")}
    if multi and not sim and not args.no_verify:
        mv = run_leg(guard, "verify_multi", LEG, lambda: M.verify_multi(b))
        if rank0:
            out["verification"] = mv
            out["verified"] = bool(mv) and "error" not in mv and all(mv.values())

    if (multi or sim) and nbig and b.scheme == "reference2d":
        # per-launch durations of an N > 1 / simulated run are wall times under two transform lanes + commitments (+ the class prover's third context):
        # two op-mix steps on ONE lane, phases apart, give the kernel's own fraction at the per-rank launch size
        un = run_leg(guard, "roofline_unoverlapped", LEG, lambda: headline.unoverlapped_roofline(b))
        if rank0:
            out["roofline_unoverlapped"] = ({"ran": "error" not in (un or {}), **({"error": un["error"]} if "error" in (un or {}) else {})} if b.emulated else un)

    next_rows = _single_gpu_legs(b, out, args, P, proof_ms, rounds_ms) if (single and rank0) else None
    class_row = None
    if with_class:
        def run_class():
            CP = M.ClassProof(b)                               # every rank raises or none does (same sizes everywhere)
            k_, w_ = (args.steps, args.warmup) if (proof_headline and not single) else (1, 0)
            dt_c, rounds_c, kern_c = headline.timed_proofs(b, guard, CP, k_, w_)
            row = CP.finish(dt_c / k_ * 1e3, rounds_c)
            return row, dt_c, rounds_c, kern_c, k_, w_, CP.overlapped
        res = run_leg(guard, "class_prover", 2 * LEG + 2.0 * args.steps, run_class)
        if isinstance(res, tuple):
            class_row, dt_c, rounds_c, kern_c, k_, w_, ovl = res
            accepted = class_row.get("accepted_by_verifier")
            if rank0 and proof_headline and not single and (sim or args.no_verify or accepted is True):
                headline.promote_proof(b, out, dt_c, rounds_c, kern_c, k_, w_,
                                       (f"rank 0's share of K five-round proofs by the coset-class prover on {sim} SIMULATED ranks (diagnostic)" if sim else
                                        f"K verified five-round proofs by the coset-class prover on all {b.world} ranks (class_prover.py; dispatcher2.rs:192-713)"), ovl)
                if not sim and not args.no_verify:
                    out["prover_verified"] = True
                    out["verified"] = bool(out.get("verified")) and True
            elif rank0 and proof_headline and not single:
                out["proof_headline_error"] = f"the class prover's proof was not accepted: {accepted!r}"
        else:
            class_row = res
            if rank0 and proof_headline and not single:
                out["proof_headline_error"] = (res or {}).get("error")
    host_bases = None
    if single and rank0 and not args.no_cpu_baseline:
        from .cpu_baseline import fetch_bases
        host_bases = run_leg(None, "cpu_baseline_bases", None, lambda: fetch_bases(b), error=lambda ex: None)
    if rank0:
        out["next_rows"] = dict(next_rows or {}, class_prover=class_row) if class_row else next_rows
        _proof_fields(out, next_rows, class_row)
        if single and proof_headline and "op_mix" in out and not args.no_verify:
            # the headline is the proof: its verdict gates `verified` — and a proof whose check did not run to a verdict is not a verified headline
            out["verified"] = bool(out.get("verified")) and out.get("prover_verified") is True
    if b.world > 1:
        guard.emit()                             # N > 1: nothing is added after this point; tear-down (communicator destruction) must not cost the line
        guard.arm("teardown", 120.0)
    cfg = dict(curve=args.curve, log_n=args.log_n, q64=b.q64, np=b.np)
    b.close()
    if single and not args.no_other_configs and args.log_n == 24 and args.curve == "bn254" and not args.dense_coset:
        from .other_configs import other_configs
        out["other_configs"] = run_leg(None, "other_configs", None, lambda: other_configs(args), error=lambda ex: [{"error": repr(ex)}])
    if single and rank0 and not args.no_cpu_baseline and host_bases is not None:
        # LAST and under the watchdog: the full-size host pass takes minutes on a slow box — it can cost its own field, never the line
        from .cpu_baseline import cpu_baseline
        guard.start_watchdog()
        out["cpu_baseline"] = run_leg(guard, "cpu_baseline", float(os.environ.get("PLONK_BENCH_CPU_BUDGET_S", "900")), lambda: cpu_baseline(args, cfg, host_bases))
    guard.emit()


