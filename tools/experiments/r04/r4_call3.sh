#!/bin/bash
# Round 4, GPU call 3: the accumulator pins of fp29.hpp / flimb.hpp without their s_nop (profiles/r04_pin_nop_experiment.txt).
#   gpurun --timeout 1800 -- 'bash tools/experiments/r04/r4_call3.sh'
# Four builds of the same sources (distributed_plonk_amd/lib/variants/*): pin_rw = the round 1-3 form `asm("" : "+v"(acc))` (one s_nop 0 per mad),
# pin_use = `asm volatile("" :: "v"(acc))` (no VGPR definition, no s_nop), pin_use_nohor = the same + -slp-vectorize-hor=false,
# nopin_nohor = no pins at all, SLP's horizontal-reduction splitting switched off instead.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 120 tools/mb 2>&1 | grep -i "mad64 chain x1\|v_mad_u64_u32 " | tee $O/r4c3_mb.txt
for V in pin_rw pin_use pin_use_nohor nopin_nohor pin_rw; do
  export PLONK_HIP_LIB=$R/distributed_plonk_amd/lib/variants/$V/libplonk_hip.so
  echo "=================== $V"
  timeout 120 python tools/coset_eval_only.py 24 2>&1 | grep coset_eval
  CURVE=bls12_381 timeout 120 python tools/coset_eval_only.py 22 2>&1 | grep coset_eval
  timeout 120 python tools/ntt_only.py 24 27 2>&1 | grep NTT
  timeout 120 python tools/msm_only.py 24 2>&1 | grep "commit ms\|accumulate_kernel \|msm_reduce\|msm_sort"
  timeout 120 python tools/msm_only.py 20 2>&1 | grep "commit ms\|accumulate_kernel \|msm_reduce"
  CURVE=bls12_381 timeout 120 python tools/msm_only.py 22 2>&1 | grep "commit ms\|accumulate_kernel \|msm_reduce"
  MSM_REDUCE_GRID=1 timeout 120 python tools/msm_only.py 24 2>&1 | grep "msm_reduce"
  CURVE=bls12_381 MSM_REDUCE_GRID=1 timeout 120 python tools/msm_only.py 22 2>&1 | grep "msm_reduce"
  timeout 200 python tools/quotient_bench.py 24 2>&1 | grep quotient_fuse
done 2>&1 | tee $O/r4c3_variants.txt
for V in pin_use nopin_nohor; do
  export PLONK_HIP_LIB=$R/distributed_plonk_amd/lib/variants/$V/libplonk_hip.so
  echo "=================== parity, $V"
  timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_ntt.py tests/test_gpu_msm.py tests/test_gpu_quotient.py tests/test_gpu_polyops.py tests/test_gpu_golden.py \
      tests/test_gpu_coset_classes.py tests/test_gpu_commit_many.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
done 2>&1 | tee $O/r4c3_parity.txt
