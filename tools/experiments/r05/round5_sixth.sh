#!/bin/bash
# Round 5, sixth GPU call: the driver-style line once more now that profiles/pmc_current.json carries the shipped source hash (so `traffic` / `valu_issue` are
# quoted), rank 0 of 8 simulated with the grid window reduction (small MSMs) against the pyramid, and kernel stats of the simulated rank.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_sixth.txt
: > $T
S="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-next-rows --no-poly-parallel --no-class-prover --simulate-ranks 8 --sim-exchange none"
one() {
  local label=$1 f=$2; shift 2
  timeout 300 "$@" > $O/$f.json 2> $O/$f.err
  python - "$label" $O/$f.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    k = d.get("kernels") or {}
    pick = {n: (k[n]["avg_ms"], k[n]["launches"]) for n in ("msm_sort", "msm_bucket_order", "msm_accumulate_kernel", "msm_reduce") if n in k}
    print(f"{sys.argv[1]:44s} step {d.get('ms_per_step')} ms  phases {d.get('phases_ms', {}).get('transforms')} / {d.get('phases_ms', {}).get('commitments')}  {pick}")
except Exception as ex:
    print(f"{sys.argv[1]:44s} FAILED: {ex!r}")
PY
}
for rep in 1 2; do
  one "sim8 step, pyramid reduction ($rep)" r05_sim8_pyr_$rep python bench.py $S
  PLONK_BENCH_OPTS="msm_reduce_grid=1" one "sim8 step, grid reduction ($rep)" r05_sim8_grid_$rep python bench.py $S
done
PLONK_BENCH_OPTS="msm_fused_order=2" one "sim8 step, fused bucket order forced" r05_sim8_fused python bench.py $S
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sim8 -o s -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-next-rows --no-poly-parallel --simulate-ranks 8 > $O/r05_sim8_under_rocprof.json 2> $O/r05_sim8_prof.err)
find $O/prof_sim8 -name "*kernel_stats.csv" -exec cp {} $O/r05_kernel_stats_sim8.csv \;
find $O/prof_sim8 -name "*.csv" -delete 2>/dev/null
python tools/kstats.py $O/r05_kernel_stats_sim8.csv 2>/dev/null | head -14 | tee -a $T
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_2p24_final.json 2> $O/r05_final_bench.err; echo "bench rc $?" | tee -a $T
python - <<'PY' | tee -a $T
import json
d = json.load(open("gpurun_out/r05_bench_2p24_final.json"))
print("step", d["ms_per_step"], d["phases_ms"]["transforms"], d["phases_ms"]["commitments"], "verified", d["verified"], "proof", d.get("proof_ms"), d.get("prover_verified"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "traffic", "valu_issue")})
print("other", [(r["kernel"], r["frac"], r["avg_launch_ms"], r["traffic"]) for r in d["roofline_other"]])
PY
cat $T
