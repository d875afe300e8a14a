"""Constants of the proof-equivalent step and the polynomial-level-parallel assignment (no GPU, no torch)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_NTT_SMALL, N_NTT_BIG, N_MSM = 7, 26, 13


# ---- polynomial-level parallelism (NTT row: "Alternative for N that fits one GPU: polynomial-level parallelism (25 independent
# coset-FFTs -> GPUs), zero communication - report both").  The 46 operations of a step are independent objects; a rank takes WHOLE operations.
# Relative costs measured on one MI355X at n = 2^24 (profiles/r03_bench_2p24_final.json: 8n zero-padded coset FFT 15.2 ms, dense 8n coset iFFT
# ~19 ms, size-n iNTT 1.9 ms, commitment 21.5 ms); the longest-processing-time rule needs ratios, not absolute times.
POLY_OP_COST = {"commit": 21.5, "coset_ifft_8n": 19.0, "coset_fft_8n": 15.2, "intt_n": 1.9}


def poly_parallel_assignment(n_ranks, nbig=N_NTT_BIG, n_small=N_NTT_SMALL, n_msm=N_MSM, cost=None):
    """-> (ops_of_rank, load_of_rank): every operation of one step on exactly one rank.  An operation is (kind, index): index = the
    commitment / polynomial / vector number of the single-GPU step, so the union over ranks is the single-GPU step on the same inputs.
    Longest-processing-time-first: operations by descending cost, each to the least loaded rank (ties: the lowest rank)."""
    cost = cost or POLY_OP_COST
    ops = [("commit", i) for i in range(n_msm)]
    if nbig:
        ops += [("coset_ifft_8n", 0)] + [("coset_fft_8n", i) for i in range(nbig - 1)]
    ops += [("intt_n", i) for i in range(n_small)]
    ops.sort(key=lambda o: -cost[o[0]])                      # stable: equal-cost operations keep their index order
    mine = [[] for _ in range(n_ranks)]
    load = [0.0] * n_ranks
    for o in ops:
        g = min(range(n_ranks), key=lambda r: (load[r], r))
        mine[g].append(o)
        load[g] += cost[o[0]]
    return mine, load
