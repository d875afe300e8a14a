#!/bin/bash
# Round 5: ClassProver(fft_helper) — rank 0 of 8 simulated, the class prover's proof with the key's 18 class evaluations inside round 3 / on a third context
# beside rounds 1-2, alternating, both exchange stand-ins.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_tenth.txt
: > $T
S="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-next-rows --no-poly-parallel --simulate-ranks 8"
show() {
python - "$1" $O/$2.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    cp = (d.get("next_rows") or {}).get("class_prover") or {}
    print(f"{sys.argv[1]:52s} class prover {cp.get('ms')} ms  {cp.get('key_class_evaluations')}  {cp.get('rounds_ms_rank0')}  resident-key variant {(cp.get('variant_resident_key_class_cosets') or {}).get('ms')}")
except Exception as ex:
    print(f"{sys.argv[1]:52s} FAILED: {ex!r}")
PY
}
for rep in 1 2; do
  PLONK_CLASS_FFT_HELPER=0 timeout 300 python bench.py $S --sim-exchange none > $O/r05_cph_off_$rep.json 2> $O/r05_cph.err; show "no exchange, key evaluations inside round 3 ($rep)" r05_cph_off_$rep
  PLONK_CLASS_FFT_HELPER=1 timeout 300 python bench.py $S --sim-exchange none > $O/r05_cph_on_$rep.json 2>> $O/r05_cph.err; show "no exchange, key evaluations beside rounds 1-2 ($rep)" r05_cph_on_$rep
done
PLONK_CLASS_FFT_HELPER=0 timeout 300 python bench.py $S > $O/r05_cph_off_s.json 2>> $O/r05_cph.err; show "stand-in, key evaluations inside round 3" r05_cph_off_s
PLONK_CLASS_FFT_HELPER=1 timeout 300 python bench.py $S > $O/r05_cph_on_s.json 2>> $O/r05_cph.err; show "stand-in, key evaluations beside rounds 1-2" r05_cph_on_s
cat $T
