#!/bin/bash
# Round 6, third GPU call: the closing collection on the round's sources (whole -m gpu suite, smoke, the driver-style line — K verified proofs timed, the
# CPU baseline measured at 2^24 —, kernel stats of the same command, PMC passes) + the N > 1 program's one-GPU diagnostics with the proof as their
# headline (rank 0 of 8 simulated with both exchange stand-ins; the whole N > 1 code path through real RCCL communicators at world 1, order check on).
#   gpurun --timeout 3000 -- 'bash tools/experiments/r06/r6_call3.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
cd $R
bash tools/closing_collection.sh r06 2>&1 | tee $O/r06_closing.txt
S="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
timeout 600 python bench.py $S --simulate-ranks 8 > $O/r06_bench_sim8.json 2> $O/r06_call3.err
timeout 600 python bench.py $S --simulate-ranks 8 --sim-exchange none > $O/r06_bench_sim8_noexchange.json 2>> $O/r06_call3.err
PLONK_COMM_CHECK_ORDER=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --multi-path > $O/r06_bench_multipath_world1_verified.json 2>> $O/r06_call3.err
python - <<'PY' | tee -a $O/r06_closing.txt
import json
for f in ("r06_bench_sim8", "r06_bench_sim8_noexchange", "r06_bench_multipath_world1_verified"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json").read().splitlines() if l.startswith("{")][-1])
        cp = (d.get("next_rows") or {}).get("class_prover") or {}
        print(f, "|", d.get("headline"), "| proof", d["ms_per_step"], "op-mix", d.get("op_mix_ms_per_step"), "verified", d.get("verified"), d.get("prover_verified"),
              "accepted", cp.get("accepted_by_verifier"), "err", d.get("proof_headline_error"), d.get("aborted_optional_leg"))
        print("   rounds", {k: v for k, v in (d.get("phases_ms") or {}).items() if k != "note"}, "exchange", json.dumps(d.get("exchange"))[:600])
    except Exception as ex:
        print(f, "FAILED", repr(ex))
PY
