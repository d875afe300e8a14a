#!/bin/bash
# rocprofv3 kernel stats of one 2^24 MSM per forced window width: tools/prof_msm.sh 20 17
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
for c in "$@"; do
  MSM_WINDOW=$c rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$c -o r -- python $GRAFT_REPO_ROOT/tools/msm_only.py 24 > $GRAFT_REPO_ROOT/gpurun_out/prof_msm_c$c.log 2>&1
  echo "== c=$c"
  f=$(find /tmp/prof_c$c -name "*kernel_stats.csv" | head -1)
  if [ -z "$f" ]; then tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_msm_c$c.log; find /tmp/prof_c$c | head; continue; fi
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/msm_c${c}_kernel_stats.csv
  python $GRAFT_REPO_ROOT/tools/kstats.py "$f" sort scan bucket_size digits reduce points_sum
done
