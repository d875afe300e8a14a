#!/bin/bash
# round-2 session D: compile-time tile shape in the swizzled NTT instantiation (per-lane invariants hoisted): parity, A/B against the previous build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_coset_classes.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py -x -q -m gpu > $O/d_tests.log 2>&1; tail -2 $O/d_tests.log
PREV=$GRAFT_REPO_ROOT/distributed_plonk_amd/lib/libplonk_hip_prev.so
{
for i in 1 2; do
  echo "-- new";  timeout 120 python tools/coset_eval_only.py 24 2>/dev/null | tail -1
  echo "-- prev"; PLONK_HIP_LIB=$PREV timeout 120 python tools/coset_eval_only.py 24 2>/dev/null | tail -1
done
echo "-- new dense"; timeout 120 python tools/ntt_only.py 24 27 2>/dev/null | tail -2
echo "-- prev dense"; PLONK_HIP_LIB=$PREV timeout 120 python tools/ntt_only.py 24 27 2>/dev/null | tail -2
} 2>&1 | tee $O/d_ntt_ab.log
