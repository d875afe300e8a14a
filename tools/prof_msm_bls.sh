#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
CURVE=bls12_381 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bls -o r -- python $GRAFT_REPO_ROOT/tools/msm_only.py 22 > $O/prof_msm_bls.log 2>&1
grep -v "^W2\|^E2\|amdgpu.ids" $O/prof_msm_bls.log
python $GRAFT_REPO_ROOT/tools/kstats.py $(find /tmp/prof_bls -name "*kernel_stats.csv" | head -1) reduce points_sum accumulate sort
