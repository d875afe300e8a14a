#!/bin/bash
# Round 5, second GPU call: what the round built on the CPU, on the device.  (a) the new parity tests; (b) rank 0 of 8 simulated — the op-mix step
# and the class prover's proof with rounds 1-2 distributed vs replicated, with and without the exchange stand-in; (c) Prover(fft_helper) at 2^24.
#   gpurun --timeout 1500 -- 'bash tools/round5_second.sh'      -> gpurun_out/r05_second.txt + gpurun_out/r05_*.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_second.txt
: > $T
timeout 900 python -m pytest tests/test_gpu_class_prover.py tests/test_gpu_golden.py tests/test_gpu_polyops.py tests/test_gpu_prover.py tests/test_gpu_coset_classes.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $T
S="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-next-rows --no-poly-parallel --simulate-ranks 8"
show() {
python - "$1" $O/$2.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    cp = (d.get("next_rows") or {}).get("class_prover") or {}
    pr = (d.get("next_rows") or {}).get("prover_rounds") or {}
    print(f"{sys.argv[1]:52s} step {d.get('ms_per_step')} ms  phases {(d.get('phases_ms') or {}).get('transforms')} / {(d.get('phases_ms') or {}).get('commitments')}"
          f"  overlap {d['config'].get('phase_overlap')}  class prover {cp.get('ms')} ms {cp.get('rounds_ms_rank0')} {cp.get('sim_exchange')}"
          f"  proof {d.get('proof_ms')} {d.get('prover_verified')} {pr.get('key_coset_ffts')} variants { {k: (v.get('ms'), v.get('same_proof_as_the_verified_one')) for k, v in (pr.get('variants') or {}).items()} }")
except Exception as ex:
    print(f"{sys.argv[1]:52s} FAILED: {ex!r}")
PY
}
timeout 300 python bench.py $S > $O/r05_bench_sim8.json 2> $O/r05_sim8.err;                                         show "sim8, stand-in exchange, rounds 1-2 distributed" r05_bench_sim8
PLONK_CLASS_REPLICATED_R12=1 timeout 300 python bench.py $S > $O/r05_bench_sim8_r12_replicated.json 2>> $O/r05_sim8.err;  show "sim8, stand-in exchange, rounds 1-2 REPLICATED" r05_bench_sim8_r12_replicated
timeout 300 python bench.py $S --sim-exchange none > $O/r05_bench_sim8_noexchange.json 2>> $O/r05_sim8.err;          show "sim8, no exchange, rounds 1-2 distributed" r05_bench_sim8_noexchange
timeout 300 python bench.py $S --sim-exchange none --overlap-phases off > $O/r05_bench_sim8_noexchange_apart.json 2>> $O/r05_sim8.err;   show "sim8, no exchange, phases apart" r05_bench_sim8_noexchange_apart
timeout 300 python bench.py $S --scheme classes > $O/r05_bench_sim8_classes.json 2>> $O/r05_sim8.err;                show "sim8, scheme classes, stand-in" r05_bench_sim8_classes
# Prover(fft_helper) at 2^24: the proof with the third context, and the same rounds without it as the variant (same process, same lease)
C="--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --next-rows proof"
PLONK_BENCH_PROOF_HELPER=1 PLONK_BENCH_HELPER_AB=1 timeout 400 python bench.py --log-n 24 $C > $O/r05_helper_2p24_on.json 2> $O/r05_helper.err;   show "2^24 proof, key coset FFTs beside rounds 1-2" r05_helper_2p24_on
PLONK_BENCH_PROOF_HELPER=0 PLONK_BENCH_HELPER_AB=1 timeout 400 python bench.py --log-n 24 $C > $O/r05_helper_2p24_off.json 2>> $O/r05_helper.err;  show "2^24 proof, key coset FFTs inside round 3" r05_helper_2p24_off
cat $T
