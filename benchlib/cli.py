"""Command line of bench.py and everything that can be decided without a GPU (--dry-run, tools/preflight_multi.sh)."""
import argparse
import os

from .common import POLY_OP_COST, poly_parallel_assignment


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=int(os.environ.get("PLONK_BENCH_LOG_N", "24")))
    ap.add_argument("--curve", default="bn254", choices=["bn254", "bls12_381"])
    ap.add_argument("--bases", default="distinct", choices=["distinct", "tiled"])
    ap.add_argument("--dense-coset", action="store_true",
                    help="feed the 25 forward coset transforms dense random 8n-point inputs through plonk_ntt_dev (the round-1 bench line) "
                         "instead of the n+3 coefficients the prover actually has (zero-padded to 8n by the reference, dispatcher2.rs:746)")
    ap.add_argument("--n-domain-only", action="store_true",
                    help="BASELINE.json configs[4] (2^28-gate BN254: 'HBM-resident witness' sizing stress): the 8n quotient domain of such a circuit "
                         "does not exist on BN254 (two-adicity 28), so only the n-domain part of the step runs - 7 iNTT(n) + 13 commitments(n)")
    ap.add_argument("--headline", default="proof", choices=["proof", "op-mix"],
                    help="what the K timed steps are.  'proof' (default): K real five-round proofs of a satisfied synthetic circuit (N == 1: prover.py; N > 1: "
                         "the coset-class prover on all ranks), verified after the timed region; the op-mix step is then measured first, as a short run of its "
                         "own (`op_mix`).  'op-mix': K op-mix steps only (the headline of rounds 1-5; what the PMC collection profiles)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run result checks (`verified` becomes null)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--next-rows", default="all", choices=["all", "proof", "none"],
                    help="the SURVEY §8f rows measured after the headline: 'all' = quotient kernel, O(n) rows, the verified proof and its same-proof "
                         "variants; 'proof' = the verified proof only (what the configs[1] / configs[3] sub-runs use); 'none'")
    ap.add_argument("--no-next-rows", action="store_true", help="same as --next-rows none")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run only: skip the compact re-runs at BASELINE.json's configs[1] (2^20 BN254) and configs[3] (2^22 BLS12-381)")
    ap.add_argument("--cpu-sample-log-n", type=int, default=20, help="cpu_baseline: log2 of the first measured sample (every op of the step once)")
    ap.add_argument("--cpu-sample-log-n2", type=int, default=22,
                    help="cpu_baseline: log2 of the second, larger measured sample (the single-threaded transforms and the commitment once each, ~25 s); "
                         "the two give the fitted exponent that is reported beside the full-size measurement.  0 = one sample only")
    ap.add_argument("--cpu-full-size", default="auto", choices=["auto", "on", "off"],
                    help="cpu_baseline: ALSO run every op of the step once at the bench size itself (2^24: ~90 s of host time, 10 GiB of host memory), so "
                         "that `cpu_baseline.value` is a MEASUREMENT at the GPU line's size (`extrapolated: false`) with the smaller samples kept beside it.  "
                         "'auto' = on when the host has the memory for it; 'off' = rounds 4-5's fitted extrapolation")
    ap.add_argument("--no-poly-parallel", action="store_true",
                    help="N > 1: skip the polynomial-level-parallel leg (whole operations per rank, no data-path collective; SURVEY §8e's alternative)")
    ap.add_argument("--simulate-ranks", type=int, default=0,
                    help="diagnostic: run rank 0's share of an S-rank job on ONE GPU with a stand-in for the exchange (--sim-exchange; results are "
                         "garbage, timings are one rank's compute plus device-to-device copies of the bytes it would exchange)")
    ap.add_argument("--sim-exchange", default="standin", choices=["standin", "none"],
                    help="--simulate-ranks: what stands in for the collectives.  'standin' (default): the bytes a collective would move through this "
                         "rank's send / receive buffers are copied device to device on the issuing stream ((S-1)/S of an all-to-all's block set, "
                         "the S-1 foreign blocks of an all-gather) — HBM speed, not xGMI speed, but the lanes' overlap meets a non-zero exchange and "
                         "`sim_exchange` prices the same bytes at the xGMI link rate; 'none': collectives return at once (compute only)")
    ap.add_argument("--class-prover", action="store_true",
                    help="also time the five prover rounds with the multi-rank coset-class prover (class_prover.py) on all ranks; "
                         "on by default for N > 1, reported under next_rows, never part of `value`")
    ap.add_argument("--no-class-prover", action="store_true", help="N > 1: skip the coset-class prover leg")
    ap.add_argument("--scheme", default="reference2d", choices=["classes", "reference2d"],
                    help="N > 1: how the step's transforms are distributed.  'classes': rank s evaluates every polynomial on ITS coset "
                         "class (the points j = s mod N of the 8n-point coset) with a local zero-padding-aware (8n/N)-point transform - no "
                         "exchange for the 25 forward coset FFTs; the quotient's coset iFFT is one class-local inverse transform + ONE all-to-all "
                         "(sum of the classes' contributions) + one all-gather; the 7 size-n iNTTs by residue class (an n/N-point class transform, one all-gather, an "
                         "interleave).  'reference2d' (default): every one "
                         "of the 33 transforms as the reference's 2-D distributed transform (row pass, RCCL all-to-all, column pass), the 25 forward "
                         "coset FFTs from zero-padded rows (plonk_fft1_dev_compact), two lanes so that exchanges overlap the next transform's passes.  "
                         "The other scheme is timed after the headline and reported as `other_scheme`")
    ap.add_argument("--overlap-phases", default="auto", choices=["auto", "on", "off"],
                    help="N = 1: issue the step's transforms on their own context WHILE the 13 commitments run on the two commitment contexts, instead "
                         "of one phase after the other.  Measured as whole bench steps, same lease (profiles/r05_opening_measurements.txt): "
                         "-13.3 %% at 2^20 BN254, -3.7 %% at 2^22 BLS12-381; -1.9 %% at 2^24 BN254 (profiles/r04_overlap_probe.txt).  'auto' = on up to "
                         "2^22 gates (launch gaps and wave tails are a tenth of such a step), off above (the 2^24 line keeps per-launch NTT timings that "
                         "a concurrent accumulation would stretch: they are what `roofline` is computed from).  N > 1 (and --simulate-ranks / "
                         "--multi-path): 'auto' = on — the two transform lanes get contexts and communicators of their own and the commitment threads "
                         "run beside the distributed transforms, so the MSMs also cover the 33 all-to-alls; every collective still comes from the "
                         "main thread (rank 0 of 8 simulated: 101.6 -> 96.2 ms per step)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: validate the arguments for this --gpus (divisibility of r and n, class count, buffer sizes per rank) and print the plan")
    ap.add_argument("--multi-path", action="store_true",
                    help="diagnostic: run the N > 1 code path (communicators, collectives, class scheme) on a world of ONE rank")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch"],
                    help="N > 1 data path: 'rccl' = the communicator inside libplonk_hip.so (plonk_comm_init; grouped ncclSend/ncclRecv on the "
                         "library's stream, no Python in the exchange), 'torch' = torch.distributed.all_to_all_single through the callback")
    args = ap.parse_args()
    if args.no_next_rows:
        args.next_rows = "none"
    return args


def plan(args, S):
    """Everything that can be decided without a GPU: used by --dry-run (tools/preflight_multi.sh) and checked again at start-up."""
    n, m = 1 << args.log_n, 8 << args.log_n
    two_adicity = 28 if args.curve == "bn254" else 32
    problems = []
    if args.n_domain_only:
        if args.log_n > two_adicity:
            problems.append(f"the domain 2^{args.log_n} exceeds the field's two-adicity {two_adicity} (DomainCreationError)")
    elif args.log_n + 3 > two_adicity:
        problems.append(f"the quotient domain 2^{args.log_n + 3} exceeds the field's two-adicity {two_adicity} (DomainCreationError); "
                        f"--n-domain-only runs the n-domain part of the step")
    if S & (S - 1):
        problems.append(f"{S} ranks: the row / column / class partitions need a power of two")
    sizes = {}
    for name, N_ in ((("n", n),) if args.n_domain_only else (("n", n), ("8n", m))):
        log = N_.bit_length() - 1
        r_, c_ = 1 << (log >> 1), 1 << (log - (log >> 1))
        if r_ % S or c_ % S:
            problems.append(f"{S} ranks do not divide r = {r_} / c = {c_} of the 2^{log}-point 2-D transform")
        sizes[name] = {"r": r_, "c": c_, "rows_per_rank": r_ // max(S, 1), "cols_per_rank": c_ // max(S, 1),
                       "bytes_per_pair_per_exchange": (r_ // S) * (c_ // S) * 32 if S > 1 else 0}
    if S > 8:
        problems.append(f"{S} ranks: the coset-class scheme needs N <= 8n/n = 8 classes")
    GiB = float(1 << 30)
    q_bytes = 64 if args.curve == "bn254" else 96
    limb_bytes = 64 if args.curve == "bn254" else 96                      # resident base record (msm_engine.hip: BaseRec)
    me_ = 0 if args.n_domain_only else m
    msm_ws = 3 * 4 * 15 * min(n // S, 1 << 26)                          # digit / sorted-index arrays of one MSM slice
    if S == 1:
        hbm = 2 * n * 32 + 2 * me_ * 32 + (n + 3) * 32 + me_ * 32 + 2 * (n * limb_bytes) + n * q_bytes + 2 * me_ * 32 + 2 * msm_ws + n * 32   # buffers + scratch + SRS (two contexts) + planes
    else:
        hbm = (2 * 2 * (n // S) * 32 + 2 * 2 * (me_ // S) * 32           # reference2d lanes
               + (2 * n * 32 + (n + 3) * 32 + 2 * (me_ // S) * 32 + 3 * me_ * 32 if me_ else 0)     # classes: bn, poly, out/mine, contrib/recv/quot
               + 2 * (n // S) * limb_bytes + (n // S) * q_bytes + (me_ // S) * 32 * 2 + 2 * msm_ws + 3 * (n // S) * 32)
    pp = None
    if S > 1 and not args.n_domain_only:
        # the polynomial-level-parallel leg: whole operations per rank, whole SRS (raw + limb form on two contexts) on every rank
        mine_, load_ = poly_parallel_assignment(S)
        worst = max(sum(1 for o in ops_ if o[0] == "commit") * n * 32 + sum(1 for o in ops_ if o[0] == "coset_fft_8n") * (n + 3) * 32
                    + sum(1 for o in ops_ if o[0] == "intt_n") * 2 * n * 32 + (2 * m * 32 if ("coset_ifft_8n", 0) in ops_ else 0) for ops_ in mine_)
        pp = {"operations_per_rank": [{k_: sum(1 for o in ops_ if o[0] == k_) for k_ in POLY_OP_COST} for ops_ in mine_],
              "modelled_load_ms_per_rank_at_2p24": [round(x, 1) for x in load_],
              "approx_hbm_GiB_per_rank": round((worst + n * q_bytes + 2 * n * limb_bytes + 3 * m * 32 + 2 * 3 * 4 * 15 * min(n, 1 << 26)) / GiB, 1)}
    return {"n": n, "m": m, "ranks": S, "scheme": args.scheme if S > 1 else "single", "transforms": sizes,
            "msm_points_per_rank": n // S, "class_points_per_rank": m // S, "approx_hbm_GiB_per_rank_headline": round(hbm / GiB, 1),
            "polynomial_parallel": pp, "problems": problems, "ok": not problems}
