#!/bin/bash
# Round 5, third GPU call: (a) the round's new parity tests on the device (summary to a file this time); (b) rank 0 of 8 simulated with VALID data in
# every receive buffer (the second call's "no exchange" class-prover figures committed all-zero vectors: void); (c) the rewritten O(n) rows at 2^24.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_third.txt
: > $T
timeout 1200 python -m pytest tests/test_gpu_class_prover.py tests/test_gpu_golden.py tests/test_gpu_polyops.py tests/test_gpu_prover.py tests/test_gpu_coset_classes.py tests/test_gpu_msm.py \
    -m gpu -x -q -p no:cacheprovider > $O/r05_third_tests.log 2>&1
grep -E "passed|failed|error" $O/r05_third_tests.log | tail -3 | tee -a $T
S="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-next-rows --no-poly-parallel --simulate-ranks 8"
show() {
python - "$1" $O/$2.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    cp = (d.get("next_rows") or {}).get("class_prover") or {}
    print(f"{sys.argv[1]:52s} step {d.get('ms_per_step')} ms  phases {(d.get('phases_ms') or {}).get('transforms')} / {(d.get('phases_ms') or {}).get('commitments')}"
          f"  overlap {d['config'].get('phase_overlap')}  class prover {cp.get('ms')} ms {cp.get('rounds_ms_rank0')} {cp.get('sim_exchange')}")
    nr = d.get("next_rows") or {}
    rows = {k: (v.get("ms"), v.get("frac")) for k, v in nr.items() if isinstance(v, dict) and "frac" in v}
    if rows:
        print(f"{'':52s} rows {rows}  proof {d.get('proof_ms')} {d.get('prover_verified')} rounds {(nr.get('prover_rounds') or {}).get('rounds_ms')}")
except Exception as ex:
    print(f"{sys.argv[1]:52s} FAILED: {ex!r}")
PY
}
timeout 300 python bench.py $S > $O/r05_bench_sim8.json 2> $O/r05_sim8.err;                                            show "sim8, stand-in exchange, rounds 1-2 distributed" r05_bench_sim8
timeout 300 python bench.py $S --sim-exchange none > $O/r05_bench_sim8_noexchange.json 2>> $O/r05_sim8.err;             show "sim8, no exchange (valid data), distributed" r05_bench_sim8_noexchange
PLONK_CLASS_REPLICATED_R12=1 timeout 300 python bench.py $S --sim-exchange none > $O/r05_bench_sim8_noexchange_r12_replicated.json 2>> $O/r05_sim8.err;  show "sim8, no exchange (valid data), rounds 1-2 REPLICATED" r05_bench_sim8_noexchange_r12_replicated
PLONK_CLASS_REPLICATED_R12=1 timeout 300 python bench.py $S > $O/r05_bench_sim8_r12_replicated.json 2>> $O/r05_sim8.err; show "sim8, stand-in exchange, rounds 1-2 REPLICATED" r05_bench_sim8_r12_replicated
timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --next-rows all > $O/r05_rows_2p24.json 2> $O/r05_rows.err;  show "2^24, next rows" r05_rows_2p24
cat $T
