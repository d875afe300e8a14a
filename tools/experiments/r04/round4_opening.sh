#!/bin/bash
# Opening GPU call of the next round (everything below was built at the end of round 3 with no GPU minutes left and is parity-checked on the
# host emulation only).  One gpurun call, ~12 min:     gpurun --timeout 1500 -- 'bash tools/experiments/r04/round4_opening.sh'
#   1. the tests that have never met a GPU: the grid window reduction, the circuit-shaped commitments, bench.py's N > 1 program on a world of one;
#   2. the grid-reduction A/B (tools/experiments/r04/ab_reduce_grid.sh: MSM alone at 2^20 / 2^21 / 2^24, the step at 2^20 / 2^24 / simulated 8 ranks);
#   3. the polynomial-level-parallel scheme: the busiest rank's share of 2, 4 and 8 ranks at 2^24 on this GPU (compute only - the scheme has no
#      data-path collective), beside rank 0's share of the reference's 2-D scheme from the same runs (`ms_per_step` of the line).
# Results: gpurun_out/r4open_*.  Adopt msm_reduce_grid as the default only if (2) agrees at every size; copy what is kept into profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_quotient.py tests/test_gpu_polyops.py tests/test_gpu_zz_bench_program.py -m gpu -q -x -p no:cacheprovider \
    -k "grid_reduction or circuit_shaped or tiny_domain or lincomb_and_blind or world_of_one or busiest_rank" 2>&1 | tail -5 | tee $O/r4open_new_tests.txt
#   1b. the differential fuzzer on the real library for the first time (15 operations, odd shapes, ~3 min); promote a slice of it to a -m gpu test once green
timeout 400 python tools/fuzz_abi.py --seconds 180 --seed 2026 --max-log 13 2>&1 | tail -3 | tee $O/r4open_fuzz.txt
timeout 700 bash tools/experiments/r04/ab_reduce_grid.sh 2>&1 | tail -40 | tee $O/r4open_ab_grid.txt
B="python bench.py --no-cpu-baseline --no-next-rows --no-other-configs --no-class-prover --steps 3 --warmup 1"
for S in 2 4 8; do
  timeout 400 $B --simulate-ranks $S > $O/r4open_sim$S.json 2>> $O/r4open.err
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r4open_sim*.json")):
    try:
        d = json.load(open(f))
        pp = d.get("polynomial_parallel") or {}
        print(f, "| reference2d rank 0 share, no exchange:", d["ms_per_step"], "ms | polynomial_parallel busiest rank:", pp.get("ms_per_step"), "ms (modelled",
              max(pp.get("modelled_load_ms_per_rank") or [0]), ") verified", pp.get("verified"), pp.get("error"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
