#!/bin/bash
# Round 6, seventh GPU call: rank 0 of 8 with the un-overlapped roofline leg (one transform lane, phases apart: the kernel's own fraction at the per-rank
# launch size, VERDICT r5 item 2's "frac >= 0.04"), and two more driver-style lines of the final sources on fresh leases' boxes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
cd $R
T=$O/r06_call7.txt
: > $T
for ex in standin none; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-poly-parallel --simulate-ranks 8 --sim-exchange $ex > $O/r06_bench_sim8_$ex.json 2> $O/r06_call7.err
  python - $ex $O/r06_bench_sim8_$ex.json >> $T <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
un = d.get("roofline_unoverlapped") or {}
r = un.get("roofline") or {}
print(f"sim8 {sys.argv[1]:8s} proof {d['ms_per_step']} ms  op-mix {d.get('op_mix_ms_per_step')} ms  | overlapped frac (proof loop) {d['roofline']['frac']} avg {d['roofline']['avg_launch_ms']} ms"
      f"  | UN-overlapped: op-mix {un.get('op_mix_ms_per_step_phases_apart')} ms, {r.get('kernel')} frac {r.get('frac')} avg {r.get('avg_launch_ms')} ms; other {[(x['kernel'], x['frac']) for x in un.get('roofline_other') or []]} {un.get('error')}")
PY
done
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_2p24_final_b.json 2>> $O/r06_call7.err
python - >> $T <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_bench_2p24_final_b.json").read().splitlines() if l.startswith("{")][-1])
print("driver-style:", d["headline"][:40], "| proof", d["ms_per_step"], "value", d["value"], "verified", d["verified"], d.get("prover_verified"), "| op-mix", d["op_mix_ms_per_step"],
      "| frac", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "traffic", d["roofline"]["traffic"], "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("extrapolated"))
for o in d.get("other_configs") or []:
    print("   ", {k: o.get(k) for k in ("config", "ms_per_step", "op_mix_ms_per_step", "frac", "verified", "prover_verified", "error")})
PY
cat $T
