#!/usr/bin/env python3
"""bench.py — the distributed MSM + NTT hot path of a PLONK proof on MI355X.

One "step" = one proof-equivalent pass of the hot path over HBM-resident synthetic inputs: the op mix
the reference prover issues per proof (/root/reference/src/dispatcher2.rs:294-691):

    7  (i)NTT of size n            (5 wire iFFTs, the permutation iFFT, the public-input iFFT)
    25 coset-NTT of size 8n        (13 selectors + 5 sigmas + 5 wires + permutation + public input)
    1  coset-iNTT of size 8n       (quotient)
    13 KZG commitments             (into_repr + n-point MSM each)

Since round 6 the HEADLINE is the proof itself: a step of the timed region = ONE real five-round proof of a satisfied 2^log_n-gate circuit
(dispatcher2.rs:192-713: 13 commitments, 7 NTT(n), 26 NTT(8n), grand product, quotient, evaluations, openings, merlin transcript), K of them
between barriers, the last one accepted by a verifier after the timed region; metric = constraints/sec = n / (time of one proof).  The op mix
above, on its own seeded inputs, is measured first as `op_mix` (the headline of rounds 1-5; `--headline op-mix` times only that).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 24] [--curve bn254]

N == 1: everything on one GPU.  Default n = 2^24, the size BASELINE.json's metric is quoted on (it fits one
        MI355X: the 8n = 2^27 transforms ping-pong two 4 GiB buffers); `--log-n 20` is configs[1].
N  > 1: launched by torchrun, one rank per GPU.  Same n (strong scaling): every NTT is the reference's
        2-D transform — row pass, ONE RCCL all-to-all over xGMI, column pass — and every MSM is
        index-sharded with a 96-byte all-gather + host add.
        Reported beside `value`, never part of it: `other_scheme` (rank-local coset classes, 2 collectives per step) and
        `polynomial_parallel` (SURVEY §8e's alternative: whole operations per rank, whole SRS on every rank, no data-path collective).
Only rank 0 prints, one JSON line.  The parts live in benchlib/ (one function per leg; every leg after the headline runs in its own
try/except and under a watchdog, so a failure there costs that leg's fields, never the line).
"""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL / multi-process)
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib import HBM_PEAK_GBS, N_MSM, N_NTT_BIG, N_NTT_SMALL, POLY_OP_COST, ResultLine, load_pmc, poly_parallel_assignment, run_leg  # noqa: E402,F401
from benchlib.cli import parse, plan  # noqa: E402


def main():
    args = parse()
    if args.dry_run:
        p_ = plan(args, max(args.gpus, args.simulate_ranks, 1))
        print(__import__("json").dumps(p_))
        raise SystemExit(0 if p_["ok"] else 2)
    # stdout carries exactly ONE JSON line: libraries that print there (RCCL's version banner at the first collective) are
    # sent to stderr by pointing fd 1 at fd 2 for the duration of the run; the result goes out through the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from benchlib.program import run
    run(args, json_fd)


if __name__ == "__main__":
    main()
