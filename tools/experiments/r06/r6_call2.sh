#!/bin/bash
# Round 6, second GPU call: (a) the first driver-style line of the round-6 program — K verified PROOFS are the timed region, the op-mix step is
# `op_mix`, the CPU baseline measured at 2^24 in full; (b) VERDICT r5 item 5: the fixed-base table (G = 1, 2, 4 bucket sets) and the grid window
# reduction at the per-rank MSM size, INSIDE rank 0 of 8's proof (2^21-point commitments) and inside the 2^20 proof (configs[1]), alternating in
# ONE lease against the default (plain Pippenger, pyramid reduction).
#   gpurun --timeout 2400 -- 'bash tools/experiments/r06/r6_call2.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r06_call2.txt
: > $T
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_2p24_first.json 2> $O/r06_call2_a.err
echo "driver-style line rc=$?" >> $T
python - $O/r06_bench_2p24_first.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("headline:", d.get("headline"), "| value", d["value"], "ms_per_step", d["ms_per_step"], "steps", d["steps"], "verified", d["verified"], "prover_verified", d.get("prover_verified"))
    print("rounds:", {k: v for k, v in d["phases_ms"].items() if k != "note"})
    print("roofline:", {k: d["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "share_of_step", "traffic")})
    print("op_mix:", d["op_mix"]["ms_per_step"], d["op_mix"]["phases_ms"], (d["op_mix"]["roofline"] or {}).get("frac"))
    c = d.get("cpu_baseline") or {}
    print("cpu_baseline:", {k: c.get(k) for k in ("value", "extrapolated", "fitted_exponent", "cores")}, [(s["log_n"], s["constraints_per_s"], s["s_per_step"]) for s in c.get("samples", [])])
    for o in d.get("other_configs") or []:
        print("other:", {k: o.get(k) for k in ("config", "ms_per_step", "op_mix_ms_per_step", "frac", "frac_in_the_overlapped_timed_region", "verified", "prover_verified", "error")})
    print("variants:", d.get("proof_variants_ms"))
    print("errors:", d.get("proof_headline_error"), d.get("aborted_optional_leg"))
except Exception as ex:
    print("FAILED to read the line:", repr(ex))
PY
S="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-poly-parallel --sim-exchange none --simulate-ranks 8"
show() {
python - "$1" $O/$2.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    k = d["kernels"]
    g = lambda n: round((k.get(n) or {}).get("total_ms", 0.0) / d["steps"], 2)
    ph = {a: b for a, b in d["phases_ms"].items() if a != "note"}
    print(f"{sys.argv[1]:44s} proof {d['ms_per_step']:8.3f} ms  op-mix {d.get('op_mix_ms_per_step')}  r1 {ph.get('round1')} r3c {ph.get('round3_commit')} r5 {ph.get('round5')}  "
          f"per proof: accumulate {g('msm_accumulate_kernel')} reduce {g('msm_reduce')} sort {g('msm_sort')} order {g('msm_bucket_order')} digits {g('msm_digits_kernel')}  err {d.get('proof_headline_error')}")
except Exception as ex:
    print(f"{sys.argv[1]:44s} FAILED: {ex!r}")
PY
}
for rep in 1 2; do
  timeout 400 python bench.py $S > $O/r06_sim8_msm_default_$rep.json 2> $O/r06_call2_b.err; show "sim8 default ($rep)" r06_sim8_msm_default_$rep
  PLONK_BENCH_OPTS=msm_reduce_grid=1 timeout 400 python bench.py $S > $O/r06_sim8_msm_grid_$rep.json 2>> $O/r06_call2_b.err; show "sim8 grid reduction ($rep)" r06_sim8_msm_grid_$rep
  for G in 1 2 4; do
    PLONK_BENCH_OPTS=msm_precompute=2,msm_table_sets=$G timeout 400 python bench.py $S > $O/r06_sim8_msm_tabG${G}_$rep.json 2>> $O/r06_call2_b.err; show "sim8 table, $G bucket set(s) ($rep)" r06_sim8_msm_tabG${G}_$rep
  done
  PLONK_BENCH_OPTS=msm_precompute=2 timeout 400 python bench.py $S > $O/r06_sim8_msm_tabauto_$rep.json 2>> $O/r06_call2_b.err; show "sim8 table, cost model's shape ($rep)" r06_sim8_msm_tabauto_$rep
  PLONK_BENCH_OPTS=msm_precompute=2,msm_table_sets=1,msm_reduce_grid=1 timeout 400 python bench.py $S > $O/r06_sim8_msm_tabG1grid_$rep.json 2>> $O/r06_call2_b.err; show "sim8 table G=1 + grid ($rep)" r06_sim8_msm_tabG1grid_$rep
done
C="--log-n 20 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --next-rows proof"
for rep in 1 2; do
  timeout 300 python bench.py $C > $O/r06_2p20_msm_default_$rep.json 2> $O/r06_call2_c.err; show "2^20 default ($rep)" r06_2p20_msm_default_$rep
  for G in 1 2; do
    PLONK_BENCH_OPTS=msm_precompute=2,msm_table_sets=$G timeout 300 python bench.py $C > $O/r06_2p20_msm_tabG${G}_$rep.json 2>> $O/r06_call2_c.err; show "2^20 table, $G bucket set(s) ($rep)" r06_2p20_msm_tabG${G}_$rep
  done
  PLONK_BENCH_OPTS=msm_reduce_grid=1 timeout 300 python bench.py $C > $O/r06_2p20_msm_grid_$rep.json 2>> $O/r06_call2_c.err; show "2^20 grid reduction ($rep)" r06_2p20_msm_grid_$rep
done
cat $T
