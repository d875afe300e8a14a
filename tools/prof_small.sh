#!/bin/bash
# Where a SMALL step spends its time (BASELINE configs[1]: 2^20 gates; and rank 0's share of an 8-rank 2^24 job): the plain bench line, then
# rocprofv3 kernel-trace stats of the same command.   usage (gpurun): bash tools/prof_small.sh [tag]
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
COMMON="--no-cpu-baseline --no-next-rows --no-other-configs"
python $R/bench.py --log-n 20 --steps 20 --warmup 3 $COMMON > $O/${TAG}_bench_2p20.json 2> $O/b20.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof20 -o s -- python $R/bench.py --log-n 20 --steps 5 --warmup 2 $COMMON --no-verify > $O/b20_prof.json 2> $O/b20_prof.err
find $O/prof20 -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_2p20.csv \;
python $R/bench.py --simulate-ranks 8 --steps 10 --warmup 2 $COMMON --no-verify --no-class-prover > $O/${TAG}_bench_sim8.json 2> $O/sim8.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sim8 -o s -- python $R/bench.py --simulate-ranks 8 --steps 3 --warmup 1 $COMMON --no-verify --no-class-prover > $O/sim8_prof.json 2> $O/sim8_prof.err
find $O/prof_sim8 -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_sim8.csv \;
find $O/prof20 $O/prof_sim8 -name "*.csv" -delete 2>/dev/null
head -c 400 $O/${TAG}_bench_2p20.json; echo; head -c 400 $O/${TAG}_bench_sim8.json; echo
head -25 $O/${TAG}_kernel_stats_2p20.csv | cut -c1-150
head -25 $O/${TAG}_kernel_stats_sim8.csv | cut -c1-150
