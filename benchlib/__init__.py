"""bench.py's parts (one function per leg, every optional leg in its own try/except):

    program.py        the run, stage by stage: op-mix step, the proof loop that takes the headline over, the legs, the CPU baseline last
    cli.py            argument parsing, the no-GPU plan (--dry-run)
    run.py            Bench: process group, contexts, resident synthetic inputs, the proof-equivalent step
    headline.py       the timed regions (op-mix steps, proofs), per-kernel HIP-event times, the roofline, the result line
    legs_single.py    N == 1: SingleProof (what the headline times) and the legs after it: verification against the oracle, next rows, variants
    legs_multi.py     N > 1: ClassProof (what the headline times) and the legs: other scheme, polynomial-level parallelism, verification
    cpu_baseline.py   the oracle timed on the host cores (reported baseline)
    other_configs.py  BASELINE.json configs[1] / configs[3] as short sub-runs
    line.py           ResultLine (one JSON line + watchdog), run_leg
    pmc.py            which PMC numbers a line may quote
"""
from .common import HBM_PEAK_GBS, N_MSM, N_NTT_BIG, N_NTT_SMALL, POLY_OP_COST, poly_parallel_assignment  # noqa: F401
from .line import ResultLine, run_leg  # noqa: F401
from .pmc import load_pmc  # noqa: F401
