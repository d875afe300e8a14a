#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for wv in 4 5 6 4 5; do MSM_WAVES=$wv timeout 300 python tools/msm_only.py 24 2>&1 | grep -E "commit|accumulate_kernel" | tr '\n' ' '; echo " waves=$wv"; done
for wv in 4 5 6; do MSM_WAVES=$wv timeout 300 python tools/msm_only.py 21 2>&1 | grep -E "commit|accumulate_kernel" | tr '\n' ' '; echo " 2^21 waves=$wv"; done
for wv in 4 5 6; do CURVE=bls12_381 MSM_WAVES=$wv timeout 300 python tools/msm_only.py 22 2>&1 | grep -E "commit|accumulate_kernel" | tr '\n' ' '; echo " bls waves=$wv"; done
