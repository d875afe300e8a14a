#!/bin/bash
# Round 4, last GPU call: evidence on the SHIPPED library that the closing collection does not cover — the N > 1 code path on one GPU (--multi-path through
# real communicators; rank 0's share of 8 ranks), configs[4]'s n-domain as a 1-GPU stress run, the fuzzer for two more minutes, clocks beside the step.
#   gpurun --timeout 1500 -- 'bash tools/experiments/r04/r4_evidence.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export MASTER_ADDR=127.0.0.1
timeout 400 python bench.py --multi-path --steps 3 --warmup 1 > $O/r04_bench_multipath_world1_verified.json 2> $O/r4ev_mp.err; echo "multi-path rc $?"
timeout 400 python bench.py --simulate-ranks 8 --steps 5 --warmup 1 > $O/r04_bench_sim8.json 2> $O/r4ev_sim8.err; echo "sim8 rc $?"
timeout 400 python bench.py --log-n 28 --n-domain-only --steps 2 --warmup 1 > $O/r04_bench_bn254_2p28_ndomain_1gpu.json 2> $O/r4ev_2p28.err; echo "2^28 rc $?"
timeout 300 python tools/fuzz_abi.py --seconds 120 --seed 4004 --max-log 13 2>&1 | tail -1 | cut -c1-400 | tee $O/r04_fuzz_device.txt
timeout 200 bash tools/clock_sample.sh 2>&1 | tail -16 | tee $O/r04_clock_samples.txt
python - <<'PY'
import json
for f in ("r04_bench_multipath_world1_verified", "r04_bench_sim8", "r04_bench_bn254_2p28_ndomain_1gpu"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        pp = d.get("polynomial_parallel") or {}
        cp = (d.get("next_rows") or {}).get("class_prover") or {}
        print(f, "| step", d["ms_per_step"], d["phases_ms"]["transforms"], d["phases_ms"]["commitments"], "| verified", d.get("verified"), "| other", (d.get("other_scheme") or {}).get("ms_per_step"),
              "| poly_parallel", pp.get("ms_per_step"), pp.get("verified"), "| class prover", cp.get("ms"), cp.get("accepted_by_verifier"), "| aborted", d.get("aborted_optional_leg"))
    except Exception as e:
        print(f, "unreadable", e)
PY
