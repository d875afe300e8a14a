#!/bin/bash
# Round 5, fifth GPU call (a second lease for the split quotient kernel; the fused bucket ordering inside the step; the division after D2's diet)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_fifth.txt
: > $T
timeout 300 python tools/quotient_ab.py 24 6 8 6 2>&1 | grep quotient_fuse | tee -a $T
timeout 300 python tools/quotient_ab.py 22 6 8 6 2>&1 | grep quotient_fuse | tee -a $T
timeout 300 python tools/quotient_ab.py 21 6 8 6 2>&1 | grep quotient_fuse | tee -a $T
QUOT_CURVE=bls12_381 timeout 300 python tools/quotient_ab.py 22 6 8 6 2>&1 | grep quotient_fuse | sed 's/^/bls12_381 /' | tee -a $T
timeout 200 python tools/poly_rows_only.py 24 2>&1 | grep "2^24" | tee -a $T
C="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-next-rows"
one() {
  local label=$1 f=$2; shift 2
  timeout 300 "$@" > $O/$f.json 2> $O/$f.err
  python - "$label" $O/$f.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    k = d.get("kernels") or {}
    pick = {n: (k[n]["avg_ms"], k[n]["launches"]) for n in ("msm_sort", "msm_bucket_order", "msm_accumulate_kernel", "msm_reduce") if n in k}
    print(f"{sys.argv[1]:44s} step {d.get('ms_per_step')} ms  phases {d.get('phases_ms', {}).get('transforms')} / {d.get('phases_ms', {}).get('commitments')}  verified {d.get('verified')}  {pick}")
except Exception as ex:
    print(f"{sys.argv[1]:44s} FAILED: {ex!r}")
PY
}
for rep in 1 2; do
  PLONK_BENCH_OPTS="msm_fused_order=0" one "2^24 step, bucket order in 3 launches ($rep)" r05_order_off_$rep python tools/bench_notorch.py $C
  one "2^24 step, bucket order fused ($rep)" r05_order_on_$rep python tools/bench_notorch.py $C
done
for rep in 1 2; do
  PLONK_BENCH_OPTS="msm_fused_order=0" one "2^20 step, bucket order in 3 launches ($rep)" r05_order20_off_$rep python tools/bench_notorch.py --log-n 20 $C
  one "2^20 step, bucket order fused ($rep)" r05_order20_on_$rep python tools/bench_notorch.py --log-n 20 $C
done
cat $T
