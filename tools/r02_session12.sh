#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
L=$R/distributed_plonk_amd/lib
for v in "" _pins _unroll "" _pins; do
  if [ -n "$v" ]; then export PLONK_HIP_LIB=$L/libplonk_hip$v.so; else unset PLONK_HIP_LIB; fi
  timeout 300 python tools/coset_eval_only.py 24 2>&1 | grep coset_eval | sed "s/^/variant[$v] /"
  timeout 300 python tools/ntt_only.py 24 27 2>&1 | grep NTT | sed "s/^/variant[$v] /"
done
export PLONK_HIP_LIB=$L/libplonk_hip_pins.so
(timeout 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_coset_classes.py -m gpu -x -q 2>&1 | tail -3)
