#!/bin/bash
# Round profile collection on the GPU box (gpurun): kernel-trace stats of the default bench, then the three PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters) each in its own run, as MI355X_MICROARCH.md prescribes.  Outputs land in gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b_prof_final.json 2> $O/b_prof_final.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
find $O/prof_final $O/pmc_fetch $O/pmc_write $O/pmc_sq -name "*.csv" | head -20
# keep the merged-back payload small: drop the per-dispatch traces, keep stats + counter collections
find $O/prof_final -name "*kernel_trace.csv" -delete
find $O/pmc_fetch $O/pmc_write $O/pmc_sq -name "*kernel_trace.csv" -delete
du -sh $O
