#!/bin/bash
# The closing collection of a round, on the sources that ship (nothing that changes a kernel lands after it): the whole -m gpu suite,
# smoke(), the driver-style bench line, kernel stats + PMC (tools/collect_profiles.sh: pmc_current.json gets the shipped source hash).  ~6 GPU-minutes.
#   gpurun --timeout 2400 -- 'bash tools/closing_collection.sh r05'      then copy what tools/collect_profiles.sh prints into profiles/
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/${TAG}_closing_gpu_suite.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/${TAG}_closing_smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_2p24_final.json 2> $O/${TAG}_closing_bench.err; echo "bench rc $?"
timeout 900 bash tools/collect_profiles.sh $TAG 2>&1 | tail -4
python - $TAG <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[1]}_bench_2p24_final.json"))
print("headline", d.get("headline"), "| proof", d["ms_per_step"], "ms x", d["steps"], "value", d["value"], "verified", d["verified"], "prover_verified", d.get("prover_verified"), d.get("proof_headline_error"))
print("rounds", {k: v for k, v in d["phases_ms"].items() if k != "note"})
op = d.get("op_mix") or {}
print("op_mix", op.get("ms_per_step"), {k: v for k, v in (op.get("phases_ms") or {}).items() if k != "note"}, (op.get("roofline") or {}).get("frac"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "traffic")}, [(r["kernel"], r["achieved"], r["frac"], r["avg_launch_ms"]) for r in d["roofline_other"]])
for oc in d.get("other_configs") or []:
    print({k: oc.get(k) for k in ("config", "ms_per_step", "op_mix_ms_per_step", "frac", "frac_in_the_overlapped_timed_region", "verified", "prover_verified", "error")})
print("next rows", {k: (v.get("ms"), v.get("frac")) for k, v in (d.get("next_rows") or {}).items() if isinstance(v, dict) and "ms" in v})
print("variants", d.get("proof_variants_ms"))
c = d.get("cpu_baseline") or {}
print("cpu", c.get("value"), "extrapolated", c.get("extrapolated"), c.get("fitted_exponent"), c.get("samples"), c.get("error"))
PY
