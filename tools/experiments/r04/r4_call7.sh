#!/bin/bash
# Round 4, GPU call 7: precomputed-quotient (Shoup) butterflies for BLS12-381's Fr: parity, then A/B in one library (PLONK_NTT_NO_SHOUP=1
# keeps the Montgomery butterflies), alternating on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_coset_classes.py tests/test_gpu_distributed.py tests/test_gpu_fullsize.py tests/test_gpu_prover.py tests/test_gpu_quotient.py \
    -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/r4c7_parity.txt
B="python bench.py --no-cpu-baseline --no-next-rows --no-other-configs --steps 3 --warmup 1 --log-n 22 --curve bls12_381"
for rep in 1 2; do
  for M in shoup mont; do
    ( [ $M = mont ] && export PLONK_NTT_NO_SHOUP=1
      echo "== $M (rep $rep)"
      CURVE=bls12_381 timeout 120 python tools/coset_eval_only.py 22 2>&1 | grep coset_eval
      CURVE=bls12_381 timeout 120 python tools/coset_eval_only.py 24 2>&1 | grep coset_eval
      timeout 300 $B > $O/r4c7_bls22_${M}_$rep.json 2>> $O/r4c7.err
      python -c "
import json; d=json.load(open('$O/r4c7_bls22_${M}_$rep.json')); print('   step', d['ms_per_step'], 'transforms', d['phases_ms']['transforms'], 'commitments', d['phases_ms']['commitments'], 'verified', d['verified'])" )
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/r4c7_ab.txt
