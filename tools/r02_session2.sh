#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python bench.py --multi-path --steps 2 --warmup 1 > $O/b_mp.json 2> $O/b_mp.err; echo multi-path rc $?; tail -c 800 $O/b_mp.err
timeout 600 python bench.py --simulate-ranks 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/b_sim8c.json 2> $O/b_sim8c.err; echo sim8 classes rc $?; tail -c 300 $O/b_sim8c.err
timeout 600 python bench.py --simulate-ranks 8 --scheme reference2d --steps 3 --warmup 1 --no-cpu-baseline > $O/b_sim8r.json 2> $O/b_sim8r.err; echo sim8 ref rc $?
timeout 600 python bench.py --simulate-ranks 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/b_sim2c.json 2> $O/b_sim2c.err; echo sim2 rc $?
python - <<'PY'
import json
for f in ["b_mp","b_sim8c","b_sim8r","b_sim2c"]:
    try:
        d=json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["ms_per_step"], d["config"]["parallelism"][:90], d.get("other_scheme"), (d.get("next_rows") or {}).get("class_prover",{}).get("ms"), d["config"].get("rccl"))
        print("   ", {k:(v["avg_ms"],v["launches"]) for k,v in d["kernels"].items() if k in ("ntt_pass_kernel","msm_accumulate_kernel","msm_sort")})
    except Exception as e:
        print(f, "ERR", e)
PY
(timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_table.py tests/test_gpu_quotient.py -m gpu -x -q 2>&1 | tail -5)
