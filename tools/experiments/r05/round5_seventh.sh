#!/bin/bash
# Round 5, seventh GPU call: the ABI fuzzer with its three new operation kinds on the device, the permutation product kernel by kernel.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_seventh.txt
: > $T
timeout 400 python tools/fuzz_abi.py --seconds 150 --seed 77 --max-log 13 2>&1 | tail -2 | tee -a $T
for op in perm_product_ranges class_ifft init_refuses_bad_srs; do timeout 200 python tools/fuzz_abi.py --seconds 25 --seed 78 --max-log 16 --only $op 2>&1 | tail -1 | tee -a $T; done
timeout 300 python tools/poly_rows_only.py 24 3 2>&1 | grep "2^24" | tee -a $T
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_perm -o rows -- python $R/tools/poly_rows_only.py 24 2 > $O/r05_perm_prof.log 2>&1)
find $O/prof_perm -name "*kernel_stats.csv" -exec cp {} $O/r05_kernel_stats_perm_rows_2p24.csv \;
python tools/kstats.py $O/r05_kernel_stats_perm_rows_2p24.csv 2>/dev/null | head -16 | tee -a $T
find $O/prof_perm -name "*.csv" -delete 2>/dev/null
cat $T
