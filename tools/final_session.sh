#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/t_final.log 2>&1; grep -E "passed|failed" $O/t_final.log | tail -2
bash tools/collect_profiles.sh r02 > $O/collect.log 2>&1; tail -1 $O/collect.log
cp $O/pmc_current.json $R/profiles/pmc_current.json
timeout 900 python bench.py --steps 5 --warmup 2 > $O/b_final.json 2> $O/b_final.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/b_final.json"))
print(d["ms_per_step"], d["verified"], d["roofline"]["frac"], d["roofline"]["traffic"], (d["roofline"]["valu_issue"] or {}).get("frac_of_launch"))
print([(o["config"][:12], o.get("ms_per_step"), o.get("verified")) for o in d["other_configs"]])
print(d["next_rows"]["prover_rounds"]["ms"], d["next_rows"]["quotient_evals_kernel"]["ms"], d["next_rows"]["perm_product"]["ms"])
PY
