#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) 2>&1 | tee $O/t5.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/b5.json 2> $O/b5.err; echo bench rc $?
python - <<'PY'
import json
d=json.load(open("gpurun_out/b5.json"))
print(d["ms_per_step"], d["verified"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_note"], d["roofline"]["valu_issue"])
print([ (o["config"], o.get("ms_per_step"), o.get("verified")) for o in d.get("other_configs",[])])
print(d["next_rows"]["quotient_evals_kernel"], d["next_rows"]["prover_rounds"]["ms"], {k:v["ms"] for k,v in d["next_rows"]["prover_rounds"]["variants"].items()})
print({k:(v["avg_ms"],v["launches"]) for k,v in d["kernels"].items()})
PY
