#!/bin/bash
# Round 4, GPU call 2: A/Bs for the commitments phase (VERDICT r3 next #1b) and BLS12-381 (next #2), one box, one call.
#   gpurun --timeout 1800 -- 'bash tools/experiments/r04/r4_call2.sh'
#   A. tools/mb: v_mad_u64_u32 as a DEPENDENT chain (1 / 2 / 4 accumulators) at 1-4 waves per SIMD
#   B. 2^24 BN254 step, commitments phase: shipped / grid reduction / side kernels at wave priority 3 / fewer persistent accumulate waves
#   C. 2^22 BLS12-381 step: the same knobs + forced windows
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 120 tools/mb 2>&1 | grep -i "mad\|instruction" | tee $O/r4c2_mb.txt
B="python bench.py --no-cpu-baseline --no-next-rows --no-other-configs --no-verify --steps 3 --warmup 1"
V3=$R/distributed_plonk_amd/lib/variants/sideprio3/libplonk_hip.so
V1=$R/distributed_plonk_amd/lib/variants/sideprio1/libplonk_hip.so
run() {   # label, lib ('' = shipped), opts, persist, window, extra args
  local lbl=$1 lib=$2 opts=$3 per=$4 win=$5; shift 5
  ( [ -n "$lib" ] && export PLONK_HIP_LIB=$lib; [ -n "$opts" ] && export PLONK_BENCH_OPTS=$opts; [ -n "$per" ] && export PLONK_BENCH_ACC_PERSIST=$per
    [ -n "$win" ] && export PLONK_BENCH_MSM_WINDOW=$win
    timeout 300 $B "$@" > $O/r4c2_$lbl.json 2>> $O/r4c2.err )
}
run bn_base     ""  ""                  "" ""
run bn_grid     ""  msm_reduce_grid=1   "" ""
run bn_p3_grid  $V3 msm_reduce_grid=1   "" ""
run bn_p3_grid_w3 $V3 msm_reduce_grid=1 3  ""
run bn_p3_grid_w2 $V3 msm_reduce_grid=1 2  ""
run bn_grid_w3  ""  msm_reduce_grid=1   3  ""
run bn_p1_grid  $V1 msm_reduce_grid=1   "" ""
run bn_base2    ""  ""                  "" ""
for c in "" 15 17; do
  run bls_base_c$c   ""  ""                "" "$c" --log-n 22 --curve bls12_381
  run bls_grid_c$c   ""  msm_reduce_grid=1 "" "$c" --log-n 22 --curve bls12_381
done
run bls_p3_grid    $V3 msm_reduce_grid=1 "" "" --log-n 22 --curve bls12_381
run bls_p3_grid_w2 $V3 msm_reduce_grid=1 2  "" --log-n 22 --curve bls12_381
run bn20_base   ""  ""                  "" "" --log-n 20 --steps 20 --warmup 3
run bn20_p3_grid $V3 msm_reduce_grid=1  "" "" --log-n 20 --steps 20 --warmup 3
python - <<'PY' | tee $O/r4c2_summary.txt
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4c2_*.json"), key=os.path.getmtime):
    try:
        d = json.load(open(f))
        k = d["kernels"]
        g = lambda n: round(k[n]["avg_ms"], 2) if n in k else None
        print("%-28s step %8.2f  transforms %7.2f  commitments %7.2f | acc %s sort %s order %s reduce %s digits %s" % (
            os.path.basename(f)[5:-5], d["ms_per_step"], d["phases_ms"]["transforms"], d["phases_ms"]["commitments"],
            g("msm_accumulate_kernel"), g("msm_sort"), g("msm_bucket_order"), g("msm_reduce"), g("msm_digits_kernel")))
    except Exception as e:
        print(f, "unreadable:", e)
PY
for G in 0 1; do
  echo "== BLS msm_only 2^22 msm_reduce_grid=$G"
  CURVE=bls12_381 MSM_REDUCE_GRID=$G timeout 120 python tools/msm_only.py 22 2>&1 | grep -v amdgpu.ids
done | tee $O/r4c2_bls_msm_only.txt
tail -5 $O/r4c2.err
