#!/bin/bash
# Round 6, fourth GPU call: plonk_trim on the device (parity test), the quotient no-alias rule, and the whole N > 1 code path through real RCCL
# communicators at world 1 with the collective order check on — its class prover ran out of memory in the third call (the op-mix legs' pooled
# exchange buffers and factor planes, ~100 GiB on the one GPU, were still cached when the proof's buffers came): ClassProof trims the contexts first.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest -m gpu -q -x -p no:cacheprovider tests/test_gpu_distributed.py tests/test_gpu_quotient.py tests/test_gpu_zz_bench_program.py 2>&1 | tail -3 | tee $O/r06_call4.txt
PLONK_COMM_CHECK_ORDER=1 timeout 1200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --multi-path > $O/r06_bench_multipath_world1_verified.json 2> $O/r06_call4.err
python - <<'PY' | tee -a $O/r06_call4.txt
import json
d = json.loads([l for l in open("gpurun_out/r06_bench_multipath_world1_verified.json").read().splitlines() if l.startswith("{")][-1])
cp = (d.get("next_rows") or {}).get("class_prover") or {}
print(d.get("headline"), "| proof", d["ms_per_step"], "op-mix", d.get("op_mix_ms_per_step"), "verified", d.get("verified"), d.get("prover_verified"), "accepted", cp.get("accepted_by_verifier"),
      "err", d.get("proof_headline_error"), d.get("aborted_optional_leg"))
print("rounds", {k: v for k, v in (d.get("phases_ms") or {}).items() if k != "note"})
print("exchange", json.dumps(d.get("exchange"))[:900])
print("op_mix exchange", json.dumps((d.get("op_mix") or {}).get("exchange"))[:900])
print("rccl", d["config"].get("rccl"))
PY
