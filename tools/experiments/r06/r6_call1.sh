#!/bin/bash
# Round 6, first GPU call: (a) the sympy-pinned golden tests and the NTT parity tests on the device with the tuned kernel on 2^5-row passes;
# (b) rank 0 of 8 simulated, op-mix step, the 2^5-row pass on the generic kernel (PLONK_NTT_LOGT5=6: round 5's behaviour) against the tuned
# one (default), alternating in ONE lease; (c) kernel stats of the simulated rank with the tuned kernel.
#   gpurun --timeout 1500 -- 'bash tools/experiments/r06/r6_call1.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r06_call1.txt
: > $T
timeout 900 python -m pytest -m gpu -q -x -p no:cacheprovider tests/test_gpu_golden.py tests/test_gpu_ntt.py tests/test_gpu_coset_classes.py tests/test_gpu_distributed.py \
    tests/test_gpu_class_prover.py > $O/r06_call1_tests.txt 2>&1
echo "tests: $(tail -1 $O/r06_call1_tests.txt)" >> $T
S="--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-next-rows --no-poly-parallel --no-class-prover --simulate-ranks 8"
show() {
python - "$1" $O/$2.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    k = d["kernels"]
    g = lambda n: (k.get(n) or {}).get("avg_ms")
    print(f"{sys.argv[1]:44s} step {d['ms_per_step']:8.3f} ms  phases {d['phases_ms']['transforms']:.1f}/{d['phases_ms']['commitments']:.1f}  frac {d['roofline']['frac']}  "
          f"ntt avg {g('ntt_pass_kernel')}  <5> {g('ntt_pass_kernel<5>')} x{(k.get('ntt_pass_kernel<5>') or {}).get('launches')}  <6> {g('ntt_pass_kernel<6>')}  <7> {g('ntt_pass_kernel<7>')}  "
          f"ntt total/step {k['ntt_pass_kernel']['total_ms'] / d['steps']:.1f}")
except Exception as ex:
    print(f"{sys.argv[1]:44s} FAILED: {ex!r}")
PY
}
for rep in 1 2; do
  for ex in none standin; do
    PLONK_NTT_LOGT5=6 timeout 300 python bench.py $S --sim-exchange $ex > $O/r06_sim8_generic5_${ex}_$rep.json 2> $O/r06_call1.err; show "generic <5> (64 columns), exchange $ex ($rep)" r06_sim8_generic5_${ex}_$rep
    timeout 300 python bench.py $S --sim-exchange $ex > $O/r06_sim8_tuned5_${ex}_$rep.json 2>> $O/r06_call1.err; show "tuned <5> (8 columns), exchange $ex ($rep)" r06_sim8_tuned5_${ex}_$rep
  done
done
for lt in 4 5; do
  PLONK_NTT_LOGT5=$lt timeout 300 python bench.py $S --sim-exchange none > $O/r06_sim8_logt5_$lt.json 2>> $O/r06_call1.err; show "generic <5>, $((1<<lt)) columns, exchange none" r06_sim8_logt5_$lt
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/r06_prof_sim8 -o sim8 --output-format csv -- python $R/bench.py $S --sim-exchange standin > $O/r06_sim8_under_rocprof.json 2>> $O/r06_call1.err
cp $(find $O/r06_prof_sim8 -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_sim8.csv 2>/dev/null
rm -rf $O/r06_prof_sim8
cat $T
