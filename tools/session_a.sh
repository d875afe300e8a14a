#!/bin/bash
# round-2 session A: bench flow after the restructure (early result line, watchdog, phases), commit batching on the simulated 8-rank shard
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
Q="--no-next-rows --no-cpu-baseline --no-other-configs"
PLONK_BENCH_WATCHDOG=1 timeout 300 python bench.py --steps 2 --warmup 1 $Q > $O/a_n1.json 2> $O/a_n1.err; echo "n1 rc=$?"
for b in 1 0; do
  PLONK_BENCH_COMMIT_BATCH=$b timeout 300 python bench.py --steps 3 --warmup 1 --simulate-ranks 8 --no-verify $Q > $O/a_sim8_ref2d_b$b.json 2> $O/a_sim8_ref2d_b$b.err; echo "sim8 ref2d batch=$b rc=$?"
done
PLONK_BENCH_COMMIT_BATCH=1 timeout 300 python bench.py --steps 3 --warmup 1 --simulate-ranks 8 --scheme classes --no-verify $Q > $O/a_sim8_cls_b1.json 2> $O/a_sim8_cls_b1.err; echo "sim8 classes rc=$?"
PLONK_BENCH_WATCHDOG=1 timeout 600 python bench.py --steps 2 --warmup 1 --multi-path $Q > $O/a_multipath.json 2> $O/a_multipath.err; echo "multipath rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/a_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d.get("phases_ms", {}).get("transforms"), d.get("phases_ms", {}).get("commitments"), d.get("verified"),
              (d.get("other_scheme") or {}).get("ms_per_step"), ((d.get("next_rows") or {}).get("class_prover") or {}).get("ms"), d.get("aborted_optional_leg"))
    except Exception as ex:
        print(f, "ERR", ex)
PY
tail -3 $O/a_n1.err $O/a_multipath.err
