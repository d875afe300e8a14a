#!/bin/bash
# Round 4, GPU call 8: pass plans of the sizes BELOW 2^24 (configs[1]: 2^20 BN254, configs[3]: 2^22 BLS12-381), whose balanced 7 + 7 + 6 / 7 + 7 + 8 split
# runs the generic pass kernel (16-column tiles: no bank swizzle, no precomputed-quotient butterflies).  Knobs: PLONK_NTT_LOGT7=3 (8-column tiles for 2^7-row
# passes), PLONK_NTT_PREFER8=1 (8 + 8 + remainder).  One box, alternating.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-next-rows --no-other-configs --no-verify"
for rep in 1 2; do
  for K in base logt7 prefer8 both; do
    ( [ $K = logt7 -o $K = both ] && export PLONK_NTT_LOGT7=3
      [ $K = prefer8 -o $K = both ] && export PLONK_NTT_PREFER8=1
      echo "== $K (rep $rep)"
      timeout 120 python tools/coset_eval_only.py 20 21 22 23 2>&1 | grep coset_eval | cut -c1-150
      CURVE=bls12_381 timeout 120 python tools/coset_eval_only.py 22 2>&1 | grep coset_eval | cut -c1-150
      timeout 120 python tools/ntt_only.py 20 22 23 2>&1 | grep NTT
      timeout 300 $B --log-n 20 --steps 20 --warmup 3 > $O/r4c8_bn20_${K}_$rep.json 2>> $O/r4c8.err
      timeout 300 $B --log-n 22 --curve bls12_381 --steps 3 --warmup 1 > $O/r4c8_bls22_${K}_$rep.json 2>> $O/r4c8.err
      python -c "
import json
for t in ('bn20','bls22'):
    d=json.load(open('$O/r4c8_%s_${K}_$rep.json' % t)); print('   ', t, 'step', d['ms_per_step'], 'transforms', d['phases_ms']['transforms'], 'commitments', d['phases_ms']['commitments'])" )
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/r4c8_ab.txt
