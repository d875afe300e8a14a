#!/bin/bash
# Round 4, GPU call 9: 8-column tiles (swizzled + Shoup instantiation) also for 2^6-row passes?  PLONK_NTT_LOGT6=3 against the default (32 columns, generic kernel).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-next-rows --no-other-configs --no-verify"
for rep in 1 2; do
  for K in base logt6; do
    ( [ $K = logt6 ] && export PLONK_NTT_LOGT6=3
      echo "== $K (rep $rep)"
      timeout 120 python tools/coset_eval_only.py 18 19 20 2>&1 | grep coset_eval | cut -c1-170
      timeout 120 python tools/ntt_only.py 12 13 19 20 2>&1 | grep NTT
      timeout 300 $B --log-n 20 --steps 20 --warmup 3 > $O/r4c9_bn20_${K}_$rep.json 2>> $O/r4c9.err
      python -c "
import json
d=json.load(open('$O/r4c9_bn20_${K}_$rep.json')); print('    bn20 step', d['ms_per_step'], 'transforms', d['phases_ms']['transforms'], 'commitments', d['phases_ms']['commitments'])" )
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/r4c9_ab.txt
