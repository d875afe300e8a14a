#!/bin/bash
# Round 4, GPU call 5: same-box A/B of the MSM unit built without pins (product) against the round-3 form (variant msm_rw); what bounds the quotient
# kernel (26 streams vs 1); the whole -m gpu suite on the product build.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
RW=$R/distributed_plonk_amd/lib/variants/msm_rw/libplonk_hip.so
B="python bench.py --no-cpu-baseline --no-next-rows --no-other-configs --no-verify --steps 3 --warmup 1"
for rep in 1 2; do
  for V in rw product; do
    ( [ $V = rw ] && export PLONK_HIP_LIB=$RW
      echo "=================== $V (rep $rep)"
      for G in 0 1; do
        echo "-- msm_reduce_grid=$G"
        MSM_REDUCE_GRID=$G timeout 120 python tools/msm_only.py 24 2>&1 | grep "commit ms\|accumulate_kernel \|msm_reduce\|msm_sort"
        MSM_REDUCE_GRID=$G timeout 120 python tools/msm_only.py 20 21 2>&1 | grep "commit ms\|accumulate_kernel \|msm_reduce"
        CURVE=bls12_381 MSM_REDUCE_GRID=$G timeout 120 python tools/msm_only.py 22 2>&1 | grep "commit ms\|accumulate_kernel \|msm_reduce"
      done
      timeout 300 $B > $O/r4c5_bn24_${V}_$rep.json 2>> $O/r4c5.err
      PLONK_BENCH_OPTS=msm_reduce_grid=1 timeout 300 $B > $O/r4c5_bn24_${V}_grid_$rep.json 2>> $O/r4c5.err
      timeout 300 $B --log-n 22 --curve bls12_381 > $O/r4c5_bls22_${V}_$rep.json 2>> $O/r4c5.err
      PLONK_BENCH_OPTS=msm_reduce_grid=1 timeout 300 $B --log-n 22 --curve bls12_381 > $O/r4c5_bls22_${V}_grid_$rep.json 2>> $O/r4c5.err
      timeout 300 $B --log-n 20 --steps 20 --warmup 3 > $O/r4c5_bn20_${V}_$rep.json 2>> $O/r4c5.err
      PLONK_BENCH_OPTS=msm_reduce_grid=1 timeout 300 $B --log-n 20 --steps 20 --warmup 3 > $O/r4c5_bn20_${V}_grid_$rep.json 2>> $O/r4c5.err )
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/r4c5_msm_only.txt
python - <<'PY' | tee $O/r4c5_summary.txt
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4c5_b*.json")):
    try:
        d = json.load(open(f))
        k = d["kernels"]
        g = lambda n: round(k[n]["avg_ms"], 2) if n in k else None
        print("%-28s step %8.2f  transforms %7.2f  commitments %7.2f | acc %s sort %s reduce %s" % (
            os.path.basename(f)[5:-5], d["ms_per_step"], d["phases_ms"]["transforms"], d["phases_ms"]["commitments"], g("msm_accumulate_kernel"), g("msm_sort"), g("msm_reduce")))
    except Exception as e:
        print(f, "unreadable:", e)
PY
echo "== quotient kernel: 26 input streams vs ONE buffer aliased 25 times"
timeout 300 python tools/quotient_bench.py 24 2>&1 | grep "quotient_fuse=[0367]" | tee $O/r4c5_quot.txt
QUOT_ALIAS=1 timeout 300 python tools/quotient_bench.py 24 2>&1 | grep "quotient_fuse=[0367]" | sed 's/^/ALIASED /' | tee -a $O/r4c5_quot.txt
echo "== the whole -m gpu suite"
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $O/r4c5_gpu_suite.txt
