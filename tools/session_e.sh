#!/bin/bash
# sustained A/B of the whole step: constexpr-tile NTT build against the previous build, same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PREV=$GRAFT_REPO_ROOT/distributed_plonk_amd/lib/libplonk_hip_prev.so
Q="--steps 6 --warmup 2 --no-next-rows --no-cpu-baseline --no-other-configs --no-verify"
for i in 1 2; do
  for which in new prev; do
    if [ $which = prev ]; then export PLONK_HIP_LIB=$PREV; else unset PLONK_HIP_LIB; fi
    timeout 200 python bench.py $Q 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernels']
print('$which', d['ms_per_step'], d['phases_ms']['transforms'], d['phases_ms']['commitments'], k['ntt_pass_kernel<8>']['avg_ms'], k['ntt_pass_kernel<9>']['avg_ms'])"
  done
done 2>&1 | tee gpurun_out/e_sustained_ab.log
