#!/bin/bash
# Round profile collection on the GPU box (gpurun): kernel-trace stats of the default bench command, then the PMC passes — FETCH_SIZE,
# WRITE_SIZE and the SQ counters each in its OWN rocprofv3 run (MI355X_MICROARCH.md; never combined with other trace domains) —
# merged by tools/pmc_collect.py into one small JSON stamped with the kernel-source hash.  Everything lands in gpurun_out/; copy
# the summaries into profiles/ afterwards (tools/collect_profiles.sh prints the cp lines).   usage: bash tools/collect_profiles.sh r03
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
# the PMC passes profile the op-mix step alone (--headline op-mix): the same kernels and launch shapes as inside a proof (99 pass launches and 7 batched
# accumulations per step / proof), without the proof's set-up under the counters' serialisation; the stats pass runs the round-6 default (proofs timed)
BENCH="python $R/bench.py --headline op-mix --steps 1 --warmup 1 --no-cpu-baseline --no-next-rows --no-other-configs --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o $TAG -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-next-rows --no-other-configs --no-verify > $O/${TAG}_bench_under_rocprof.json 2> $O/prof_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $BENCH > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $BENCH > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -o p -- $BENCH > /dev/null 2> $O/pmc_sq.err
cd $R
python tools/pmc_collect.py $O/pmc_fetch $O/pmc_write $O/pmc_sq 24 bn254 1 $O/pmc_current.json
find $O/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_2p24.csv \;
# the same trace grouped by grid size: the n-point and 8n-point passes of ntt_pass_kernel separately (the stats file mixes them in the whole command's
# ratio, one timed proof in 21 : 78 — tools/ktrace_by_grid.py)
python tools/ktrace_by_grid.py $O/prof_stats ntt_pass msm_accumulate_kernel quotient_evals > $O/${TAG}_kernel_trace_by_grid_2p24.txt 2>&1
# FETCH_SIZE calibration on known byte counts (wide stream vs the MSM's 72-byte gathers)
if [ -x $R/tools/fetch_calib_bin ]; then
  (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_calib -o p -- $R/tools/fetch_calib_bin > $O/fetch_calib.out 2> $O/fetch_calib.err)
  python tools/fetch_calib_report.py $O/pmc_calib > $O/${TAG}_fetch_calibration.txt; cat $O/fetch_calib.out >> $O/${TAG}_fetch_calibration.txt
  cat $O/${TAG}_fetch_calibration.txt
fi
# one 2^24 MSM ALONE (single context, nothing interleaved): the sort / reduce / accumulate costs without the second stream's contention
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msm_alone -o m -- python $R/tools/msm_only.py 24 > $O/${TAG}_msm_alone.log 2>&1
find $O/prof_msm_alone -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_msm_alone_2p24.csv \;
find $O/prof_msm_alone -name "*.csv" -delete 2>/dev/null
# keep the merged-back payload small: drop the per-dispatch traces and raw counter dumps
find $O/prof_stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_calib -name "*.csv" -delete 2>/dev/null
ls -la $O/pmc_current.json $O/${TAG}_kernel_stats_2p24.csv
head -5 $O/${TAG}_kernel_stats_2p24.csv | cut -c1-160
echo "cp gpurun_out/pmc_current.json profiles/pmc_current.json; cp gpurun_out/pmc_current.json profiles/${TAG}_pmc_2p24.json; cp gpurun_out/${TAG}_kernel_stats_2p24.csv gpurun_out/${TAG}_kernel_trace_by_grid_2p24.txt profiles/; cp gpurun_out/${TAG}_fetch_calibration.txt profiles/; cp gpurun_out/${TAG}_kernel_stats_msm_alone_2p24.csv profiles/"
