"""BASELINE.json's other single-GPU configurations, each as its own short run of bench.py AFTER the headline measurement has released
the GPU (never part of `value`): configs[1] (2^20-gate BN254) and configs[3] (2^22-gate BLS12-381), each with its op-mix step, its
verification and its verified proofs (the sub-run's own headline)."""
import json
import os
import subprocess
import sys

from .common import ROOT

CONFIGS = (("configs[1]: 2^20-gate BN254, 1 GPU", ["--log-n", "20", "--curve", "bn254"]),
           ("configs[3]: 2^22-gate BLS12-381, 1 GPU", ["--log-n", "22", "--curve", "bls12_381"]))


def entry_of(label, d_, fallback=None):
    """one sub-run's line -> its entry.  Since round 6 the sub-run's headline is its verified PROOF (ms_per_step = proof_ms) with the op-mix step as
    `op_mix`.  `frac`: these sizes overlap contexts inside the timed region, which stretches every launch, so the fraction
    quoted here is the one from the sub-run's UN-OVERLAPPED op-mix steps (`roofline_unoverlapped`) and says so; the overlapped figure is kept under
    its own name and must not be compared with the 2^24 line's."""
    rf = d_.get("roofline") or {}
    un_leg = d_.get("roofline_unoverlapped") or {}
    un_all = [e_ for e_ in [un_leg.get("roofline")] + list(un_leg.get("roofline_other") or []) if e_]
    by_kernel = {e_["kernel"]: {"frac": e_.get("frac"), "avg_launch_ms": e_.get("avg_launch_ms")} for e_ in un_all if e_.get("kernel")}
    op = d_.get("op_mix") or {}
    overlapped = bool((d_.get("config") or {}).get("phase_overlap")) or bool(rf.get("overlap_note"))
    # `frac` is about ONE kernel in every entry — ntt_pass_kernel, the kernel the 2^24 line's `roofline` is about — whichever kernel dominates the sub-run
    # (on BLS12-381 the un-overlapped accumulation outweighs the transforms); the other kernels' un-overlapped fractions sit beside it
    if by_kernel:
        kern = "ntt_pass_kernel" if "ntt_pass_kernel" in by_kernel else next(iter(by_kernel))
        use = dict(by_kernel[kern], kernel=kern)
    else:
        use = {} if overlapped else rf
    un = bool(by_kernel)
    return {"config": label, "headline": d_.get("headline"), "ms_per_step": d_["ms_per_step"], "constraints_per_s": d_["value"], "steps": d_["steps"],
            "op_mix_ms_per_step": op.get("ms_per_step"), "op_mix_constraints_per_s": op.get("constraints_per_s"),
            "op_mix_phases_ms": {k_: v_ for k_, v_ in (op.get("phases_ms") or {}).items() if k_ != "note"},
            "rounds_ms": {k_: v_ for k_, v_ in (d_.get("phases_ms") or {}).items() if k_ != "note"},
            "phase_overlap": (d_.get("config") or {}).get("phase_overlap"),
            "dominant_kernel": use.get("kernel") or rf.get("kernel"), "frac": use.get("frac"), "avg_launch_ms": use.get("avg_launch_ms"),
            "frac_by_kernel_unoverlapped": by_kernel or None,
            "frac_source": ("roofline_unoverlapped: two op-mix steps with the phases one after the other (launch durations without another context's kernels "
                            "beside them)" if un else ("the timed region (no context overlap in this run)" if use else
                                                       "none: the timed region overlaps contexts and no un-overlapped pass ran — a fraction from stretched launches is not quoted")),
            "frac_in_the_overlapped_timed_region": ({"kernel": rf.get("kernel"), "frac": rf.get("frac")} if overlapped else None),
            "verified": d_.get("verified"), "verification": d_.get("verification"),
            "proof_ms": d_.get("proof_ms"), "proof_constraints_per_s": d_.get("proof_constraints_per_s"),
            "prover_verified": d_.get("prover_verified"), "proof_variants_ms": d_.get("proof_variants_ms"), **({"overlap_run_failed": fallback} if fallback else {})}


BUDGET_S = 300.0     # for BOTH sub-runs together (each takes ~25 s): the headline line is only written after this leg returns


def other_configs(args):
    import time
    other = []
    t_end = time.monotonic() + BUDGET_S
    for label, extra in CONFIGS:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--bases", args.bases,
               "--no-cpu-baseline", "--next-rows", "proof", "--no-other-configs"] + extra
        try:
            fallback = None
            try:
                # under the headline watchdog (benchlib/line.py): were the overlapped warm-up / timed steps ever to hang, the sub-run ends itself
                # after a minute with exit code 4 instead of sitting out the timeout below
                env = dict(os.environ, PLONK_BENCH_WATCHDOG="1", PLONK_BENCH_HEADLINE_BUDGET_S="60")
                res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=max(20.0, min(150.0, t_end - time.monotonic())), check=True,
                                     env=env)
            except Exception as ex:         # noqa: BLE001 - these sizes overlap their two phases by default (--overlap-phases auto): if that run fails,
                fallback = repr(ex)         # the phase-after-phase form of rounds 1-3 still gives the line, and the failure is recorded beside it
                if t_end - time.monotonic() < 45.0:
                    raise TimeoutError(f"no time left for the phase-after-phase fall-back after {fallback}")
                res = subprocess.run(cmd + ["--overlap-phases", "off"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                     timeout=min(150.0, t_end - time.monotonic()), check=True, env=dict(os.environ, PLONK_BENCH_PROOF_HELPER="0"))
            d_ = json.loads(res.stdout.decode().strip().splitlines()[-1])
            other.append(entry_of(label, d_, fallback))
        except Exception as ex:             # noqa: BLE001 - the extra lines must never break the headline
            other.append({"config": label, "error": repr(ex)})
    return other
