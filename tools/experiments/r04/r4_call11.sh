#!/bin/bash
# Round 4, GPU call 11: window width at 2^20 points and persistent-wave count on BLS12-381 (3 waves per SIMD there), in the step; one box, alternating.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-next-rows --no-other-configs --no-verify"
one() { python -c "
import json,sys
d=json.load(open(sys.argv[1])); k=d['kernels']; print('   ', sys.argv[2], 'step', d['ms_per_step'], 'transforms', d['phases_ms']['transforms'], 'commitments', d['phases_ms']['commitments'], 'acc', round(k['msm_accumulate_kernel']['avg_ms'],2), 'reduce', round(k['msm_reduce']['avg_ms'],2), 'sort', round(k['msm_sort']['avg_ms'],2))" $1 "$2"; }
for rep in 1 2; do
  for W in 0 14 15 17; do
    PLONK_BENCH_MSM_WINDOW=$W timeout 300 $B --log-n 20 --steps 20 --warmup 3 > $O/r4c11_bn20_w${W}_$rep.json 2>> $O/r4c11.err; one $O/r4c11_bn20_w${W}_$rep.json "2^20 BN254 window $W"
  done
  for P in 4 3 6; do
    PLONK_BENCH_ACC_PERSIST=$P timeout 300 $B --log-n 22 --curve bls12_381 --steps 3 --warmup 1 > $O/r4c11_bls_p${P}_$rep.json 2>> $O/r4c11.err; one $O/r4c11_bls_p${P}_$rep.json "2^22 BLS12-381 persist $P"
  done
  for P in 4 5; do
    PLONK_BENCH_ACC_PERSIST=$P timeout 300 $B --steps 3 --warmup 1 > $O/r4c11_bn24_p${P}_$rep.json 2>> $O/r4c11.err; one $O/r4c11_bn24_p${P}_$rep.json "2^24 BN254 persist $P"
  done
done 2>&1 | tee $O/r4c11_ab.txt
