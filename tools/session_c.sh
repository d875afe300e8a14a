#!/bin/bash
# round-2 session C: Prover / ClassProver commit their rounds through plonk_commit_many_dev
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_prover.py tests/test_gpu_class_prover.py -x -q -m gpu > $O/c_tests.log 2>&1; tail -2 $O/c_tests.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-verify > $O/c_bench.json 2> $O/c_bench.err; echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c_bench.json").read().strip().splitlines()[-1])
pr = d["next_rows"]["prover_rounds"]
print(d["ms_per_step"], pr["ms"], pr["rounds_ms"], {k: v.get("ms") for k, v in pr["variants"].items()})
PY
