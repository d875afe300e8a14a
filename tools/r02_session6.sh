#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for L in 2 3 4; do
PLONK_BENCH_COMMIT_LANES=$L timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-next-rows --no-other-configs --no-verify > $O/b6_$L.json 2> $O/b6_$L.err
python - <<PY
import json
d=json.load(open("gpurun_out/b6_$L.json"))
k=d["kernels"]
ntt=k["ntt_pass_kernel"]["total_ms"]/d["steps"]
print("lanes $L", d["ms_per_step"], "ntt", round(ntt,1), "msm phase", round(d["ms_per_step"]-ntt,1), {x:k[x]["avg_ms"] for x in ("msm_accumulate_kernel","msm_sort","msm_reduce")})
PY
done
