#!/bin/bash
# GPU session: correctness of the new transport / lazy canon, tile-width sweep, default bench (verified + other configs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_class_prover.py tests/test_gpu_ntt.py tests/test_gpu_coset_classes.py tests/test_gpu_prover.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -25) > $O/t2.log 2>&1
tail -8 $O/t2.log
for lt in 2 3 4; do PLONK_NTT_LOGT8=$lt timeout 300 python tools/coset_eval_only.py 24 2>&1 | grep coset_eval; done > $O/sweep_logt8.txt 2>&1
PLONK_NTT_LOGT9=2 PLONK_NTT_LOGT8=2 timeout 300 python tools/ntt_only.py 24 27 >> $O/sweep_logt8.txt 2>&1
timeout 300 python tools/ntt_only.py 24 27 >> $O/sweep_logt8.txt 2>&1
cat $O/sweep_logt8.txt
timeout 900 python bench.py > $O/b2.json 2> $O/b2.err; echo bench rc $?; tail -c 600 $O/b2.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/b2.json"))
print(d["ms_per_step"], d["verified"], d["verification"], d["roofline"]["frac"], d["cpu_baseline"])
print(json.dumps(d.get("other_configs")))
print(json.dumps(d["kernels"]))
PY
