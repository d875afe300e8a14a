#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command (the file profiles/rNN_kernel_stats_2p24.csv is made from)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b_prof_final.json 2> $O/b_prof_final.err
find $O/prof_final -name "*kernel_trace.csv" -delete
python $R/tools/kstats.py $O/prof_final/r01_kernel_stats.csv ntt_pass msm_accumulate quotient_evals sort_scatter
head -c 200 $O/b_prof_final.json
