#!/bin/bash
# round-2 session B: wave-local ordering points in the class-route NTT kernel (parity, A/B), tile width check, commit lanes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_coset_classes.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/b_tests.log 2>&1; tail -2 $O/b_tests.log
{
for i in 1 2; do
  echo "-- wave-local (default)"; timeout 120 python tools/coset_eval_only.py 24 2>/dev/null | tail -1
  echo "-- all barriers";         PLONK_NTT_ALL_BARRIERS=1 timeout 120 python tools/coset_eval_only.py 24 2>/dev/null | tail -1
done
echo "-- LOGT8=2"; PLONK_NTT_LOGT8=2 timeout 120 python tools/coset_eval_only.py 24 2>/dev/null | tail -1
echo "-- LOGT8=4"; PLONK_NTT_LOGT8=4 timeout 120 python tools/coset_eval_only.py 24 2>/dev/null | tail -1
} 2>&1 | tee $O/b_ntt_ab.log
Q="--no-next-rows --no-cpu-baseline --no-other-configs --no-verify --steps 3 --warmup 1"
for l in 2 3; do
  PLONK_BENCH_COMMIT_LANES=$l timeout 300 python bench.py $Q 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('commit lanes $l:', d['ms_per_step'], d['phases_ms']['transforms'], d['phases_ms']['commitments'])"
done 2>&1 | tee $O/b_lanes.log
