#!/bin/bash
# commit_many experiment: tests + bench with/without batching, 1 or 2 lanes, at 2^24 / 2^20 / simulated 8-rank shard
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_commit_many.py tests/test_gpu_msm.py -x -q -m gpu > gpurun_out/cm_tests.log 2>&1
tail -5 gpurun_out/cm_tests.log
Q="--no-verify --no-next-rows --no-other-configs --steps 3 --warmup 1"
for cfg in "1 2" "1 1" "0 2"; do
  set -- $cfg
  for ln in 24 20; do
    echo "== batch=$1 lanes=$2 log_n=$ln"
    PLONK_BENCH_COMMIT_BATCH=$1 PLONK_BENCH_COMMIT_USE_LANES=$2 timeout 600 python bench.py $Q --log-n $ln 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], {k: v['total_ms'] for k, v in d.get('kernels', {}).items() if k.startswith('msm')})
"
  done
done 2>&1 | tee gpurun_out/cm_bench.log
