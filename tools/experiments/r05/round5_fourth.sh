#!/bin/bash
# Round 5, fourth GPU call: the WHOLE -m gpu suite on the round's sources so far, the split quotient kernel against the compact one (alternating, one
# process), and a kernel-level view of the rewritten division / evaluation.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_fourth.txt
: > $T
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r05_fourth_tests.log 2>&1
grep -E "passed|failed|error" $O/r05_fourth_tests.log | tail -3 | tee -a $T
timeout 300 python tools/quotient_ab.py 24 6 8 6 2>&1 | tail -3 | tee -a $T
timeout 300 python tools/quotient_ab.py 20 6 8 6 2>&1 | tail -3 | tee -a $T
timeout 200 python tools/poly_rows_only.py 24 2>&1 | tail -3 | tee -a $T
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rows -o rows -- python $R/tools/poly_rows_only.py 24 3 > $O/r05_rows_prof.log 2>&1)
find $O/prof_rows -name "*kernel_stats.csv" -exec cp {} $O/r05_kernel_stats_poly_rows_2p24.csv \;
python tools/kstats.py $O/r05_kernel_stats_poly_rows_2p24.csv 2>/dev/null | head -12 | tee -a $T
find $O/prof_rows -name "*.csv" -delete 2>/dev/null
cat $T
