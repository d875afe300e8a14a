#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_table.py tests/test_gpu_prover.py -m gpu -x -q 2>&1 | tail -3)
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k msm 2>&1 | tail -3)
for l in 24 21; do timeout 300 python tools/msm_only.py $l 2>&1 | grep -v amdgpu.ids | grep -E "commit|sort|accum|reduce"; done
MSM_WINDOW=18 timeout 300 python tools/msm_only.py 24 2>&1 | grep -E "commit|sort|accum|reduce"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-next-rows --no-other-configs > $O/b8.json 2> $O/b8.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/b8.json"))
k=d["kernels"]; ntt=k["ntt_pass_kernel"]["total_ms"]/d["steps"]
print(d["ms_per_step"], d["verified"], "ntt", round(ntt,1), "msm phase", round(d["ms_per_step"]-ntt,1), {x:k[x]["avg_ms"] for x in ("msm_accumulate_kernel","msm_sort","msm_reduce")})
PY
