#!/bin/bash
# Round 5: the wires' class evaluations on the helper too (after round 1's polynomials exist), against the key's only.  Rank 0 of 8, alternating.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_eleventh.txt
: > $T
S="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-next-rows --no-poly-parallel --simulate-ranks 8"
show() {
python - "$1" $O/$2.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    cp = (d.get("next_rows") or {}).get("class_prover") or {}
    print(f"{sys.argv[1]:52s} class prover {cp.get('ms')} ms  {cp.get('rounds_ms_rank0')}")
except Exception as ex:
    print(f"{sys.argv[1]:52s} FAILED: {ex!r}")
PY
}
for rep in 1 2; do
  PLONK_CLASS_WIRE_HELPER=0 timeout 300 python bench.py $S --sim-exchange none > $O/r05_cpw_off_$rep.json 2> $O/r05_cpw.err; show "no exchange, helper: key only ($rep)" r05_cpw_off_$rep
  timeout 300 python bench.py $S --sim-exchange none > $O/r05_cpw_on_$rep.json 2>> $O/r05_cpw.err; show "no exchange, helper: key + wires ($rep)" r05_cpw_on_$rep
done
PLONK_CLASS_WIRE_HELPER=0 timeout 300 python bench.py $S > $O/r05_cpw_off_s.json 2>> $O/r05_cpw.err; show "stand-in, helper: key only" r05_cpw_off_s
timeout 300 python bench.py $S > $O/r05_cpw_on_s.json 2>> $O/r05_cpw.err; show "stand-in, helper: key + wires" r05_cpw_on_s
cat $T
