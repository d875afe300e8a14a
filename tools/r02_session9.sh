#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_coset_classes.py tests/test_gpu_ntt.py tests/test_gpu_class_prover.py -m gpu -x -q 2>&1 | tail -12)
timeout 600 python bench.py --simulate-ranks 8 --scheme reference2d --steps 3 --warmup 1 --no-cpu-baseline > $O/b9_sim8r.json 2> $O/b9.err; echo rc $?; tail -c 300 $O/b9.err
timeout 600 python bench.py --simulate-ranks 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/b9_sim8c.json 2>> $O/b9.err; echo rc $?
timeout 600 python bench.py --multi-path --steps 2 --warmup 1 > $O/b9_mp.json 2>> $O/b9.err; echo rc $?
python - <<'PY'
import json
for f in ["b9_sim8r","b9_sim8c","b9_mp"]:
    d=json.load(open(f"gpurun_out/{f}.json")); print(f, d["ms_per_step"], d["config"]["scheme"], d.get("other_scheme"), (d.get("next_rows") or {}).get("class_prover",{}).get("ms"))
PY
