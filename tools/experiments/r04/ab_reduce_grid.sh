#!/bin/bash
# First GPU call for the grid reduction ("msm_reduce_grid", msm_engine.hip 5b — built at the end of round 3 with no GPU minutes left: parity on the host
# emulation only).  One gpurun call, ~6 min:   gpurun --timeout 900 -- 'bash tools/experiments/r04/ab_reduce_grid.sh'
#   1. its parity test on the real device (both curves, every window width, batched round);
#   2. one MSM alone at 2^20 / 2^21 / 2^24 points, pyramid vs grid, per-phase HIP-event times (msm_reduce is the line to read);
#   3. the step at 2^20 and 2^24 and rank 0's share of an 8-rank job, pyramid vs grid, same box, same call, verified.
# Expected from the structure: the reduction of a 2^20 / 2^21-point MSM several times shorter (8 + 2 dependent launches of 5- and 16-deep
# addition chains -> 2 launches of tree sums), neutral at 2^24 (same two additions per bucket).  Adopt as the default only if (3) agrees.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_msm.py -m gpu -q -x -k "grid_reduction" -p no:cacheprovider 2>&1 | tail -3 | tee $O/ab_grid_parity.txt
for LOGN in 20 21 24; do
  for G in 0 1; do
    echo "== msm_only 2^$LOGN msm_reduce_grid=$G" | tee -a $O/ab_grid_msm_only.txt
    MSM_REDUCE_GRID=$G python tools/msm_only.py $LOGN 2>&1 | tee -a $O/ab_grid_msm_only.txt
  done
done
B="python bench.py --no-cpu-baseline --no-next-rows --no-other-configs"
for G in 0 1 0 1; do
  PLONK_BENCH_OPTS=msm_reduce_grid=$G $B --log-n 20 --steps 20 --warmup 3 > $O/ab_grid_step_2p20_g${G}_$RANDOM.json 2>> $O/ab_grid.err
done
for G in 0 1; do
  PLONK_BENCH_OPTS=msm_reduce_grid=$G $B --steps 3 --warmup 1 > $O/ab_grid_step_2p24_g$G.json 2>> $O/ab_grid.err
  PLONK_BENCH_OPTS=msm_reduce_grid=$G $B --simulate-ranks 8 --steps 5 --warmup 1 > $O/ab_grid_step_sim8_g$G.json 2>> $O/ab_grid.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/ab_grid_step_*.json")):
    try:
        d = json.load(open(f))
        print(f, "ms_per_step", d["ms_per_step"], "commitments", d["phases_ms"]["commitments"], "verified", d["verified"], d["config"].get("experiment_opts"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
