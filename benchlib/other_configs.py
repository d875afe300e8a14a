"""BASELINE.json's other single-GPU configurations, each as its own short run of bench.py AFTER the headline measurement has released
the GPU (never part of `value`): configs[1] (2^20-gate BN254) and configs[3] (2^22-gate BLS12-381), each with its op-mix step, its
verification and one real proof through the verifier."""
import json
import os
import subprocess
import sys

from .common import ROOT

CONFIGS = (("configs[1]: 2^20-gate BN254, 1 GPU", ["--log-n", "20", "--curve", "bn254"]),
           ("configs[3]: 2^22-gate BLS12-381, 1 GPU", ["--log-n", "22", "--curve", "bls12_381"]))


BUDGET_S = 240.0     # for BOTH sub-runs together (each takes ~25 s): the headline line is only written after this leg returns


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
            rf = d_.get("roofline") or {}
            other.append({"config": label, "ms_per_step": d_["ms_per_step"], "constraints_per_s": d_["value"], "steps": d_["steps"],
                          "phases_ms": {k_: v_ for k_, v_ in (d_.get("phases_ms") or {}).items() if k_ != "note"},
                          "phase_overlap": (d_.get("config") or {}).get("phase_overlap"),      # True: `frac` below is from launches stretched by the overlap
                          "dominant_kernel": rf.get("kernel"), "frac": rf.get("frac"), "avg_launch_ms": rf.get("avg_launch_ms"),
                          "verified": d_.get("verified"), "verification": d_.get("verification"),
                          "proof_ms": d_.get("proof_ms"), "proof_constraints_per_s": d_.get("proof_constraints_per_s"),
                          "prover_verified": d_.get("prover_verified"), "proof_variants_ms": d_.get("proof_variants_ms"), **({"overlap_run_failed": fallback} if fallback else {})})
        except Exception as ex:             # noqa: BLE001 - the extra lines must never break the headline
            other.append({"config": label, "error": repr(ex)})
    return other
