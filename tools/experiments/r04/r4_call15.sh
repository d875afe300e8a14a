#!/bin/bash
# Round 4, GPU call 15: resident base records as one aligned 64-byte sector (x || y in 32-bit words; BLS12-381: 96 B) against the round 1-3 limb form
# (72 / 112 B, 8-byte aligned): parity of the MSM suites on the packed build, then both builds alternating on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_table.py tests/test_gpu_commit_many.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/r4c15_parity.txt
OLD=$R/distributed_plonk_amd/lib/variants/bases72/libplonk_hip.so
B="python bench.py --no-cpu-baseline --no-next-rows --no-other-configs --no-verify --steps 3 --warmup 1"
for rep in 1 2; do
  for V in limb72 packed64; do
    ( [ $V = limb72 ] && export PLONK_HIP_LIB=$OLD
      echo "== $V (rep $rep)"
      timeout 120 python tools/msm_only.py 24 2>&1 | grep "commit ms\|accumulate_kernel "
      timeout 120 python tools/msm_only.py 20 22 2>&1 | grep "commit ms\|accumulate_kernel "
      CURVE=bls12_381 timeout 120 python tools/msm_only.py 22 2>&1 | grep "commit ms\|accumulate_kernel "
      timeout 300 $B > $O/r4c15_bn24_${V}_$rep.json 2>> $O/r4c15.err
      timeout 300 $B --log-n 22 --curve bls12_381 > $O/r4c15_bls22_${V}_$rep.json 2>> $O/r4c15.err
      timeout 300 $B --log-n 20 --steps 20 --warmup 3 > $O/r4c15_bn20_${V}_$rep.json 2>> $O/r4c15.err
      python -c "
import json
for t in ('bn24','bls22','bn20'):
    d=json.load(open('$O/r4c15_%s_${V}_$rep.json' % t)); print('   ', t, 'step', d['ms_per_step'], 'transforms', d['phases_ms']['transforms'], 'commitments', d['phases_ms']['commitments'], 'acc', round(d['kernels']['msm_accumulate_kernel']['avg_ms'],2))" )
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/r4c15_ab.txt
