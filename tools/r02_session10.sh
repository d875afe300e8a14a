#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_table.py -m gpu -x -q 2>&1 | tail -3)
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k msm 2>&1 | tail -3)
for f in 1 0 1 0; do MSM_FUSED=$f timeout 300 python tools/msm_only.py 24 2>&1 | grep -E "commit|accumulate_kernel" | tr '\n' ' '; echo " fused=$f"; done
for f in 1 0; do CURVE=bls12_381 MSM_FUSED=$f timeout 300 python tools/msm_only.py 22 2>&1 | grep -E "commit|accumulate_kernel" | tr '\n' ' '; echo " bls fused=$f"; done
