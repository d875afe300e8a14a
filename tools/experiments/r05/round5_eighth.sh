#!/bin/bash
# Round 5, eighth GPU call: (a) the WHOLE N > 1 code path through real RCCL communicators on a world of one rank, with the collective order check on
# (verification leg, class prover with rounds 1-2 distributed through ncclAllGather, proof handed to the verifier); (b) rank 0 of 2 / 4 / 8 simulated:
# the scaling picture one GPU can give; (c) configs[4]'s per-rank share (2^28 gates, n-domain only, rank 0 of 8).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_eighth.txt
: > $T
show() {
python - "$1" $O/$2.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    cp = (d.get("next_rows") or {}).get("class_prover") or {}
    print(f"{sys.argv[1]:40s} step {d.get('ms_per_step')} ms ({(d.get('phases_ms') or {}).get('transforms')} + {(d.get('phases_ms') or {}).get('commitments')})  verified {d.get('verified')}"
          f"  class prover {cp.get('ms')} ms accepted {cp.get('accepted_by_verifier')} resident-key variant {(cp.get('variant_resident_key_class_cosets') or {}).get('ms')}"
          f"  rccl {(d.get('config') or {}).get('rccl')}  other_scheme {(d.get('other_scheme') or {}).get('ms_per_step')} poly_parallel {(d.get('polynomial_parallel') or {}).get('ms_per_step')}")
    if d.get("verification"):
        print(f"{'':40s} verification {d['verification']}")
except Exception as ex:
    print(f"{sys.argv[1]:40s} FAILED: {ex!r}")
PY
}
C="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-next-rows"
PLONK_COMM_CHECK_ORDER=1 timeout 600 python bench.py $C --multi-path > $O/r05_bench_multipath_world1_verified.json 2> $O/r05_multipath.err; show "multi-path, world 1, RCCL, order check on" r05_bench_multipath_world1_verified
for S in 2 4 8; do
  timeout 400 python bench.py $C --simulate-ranks $S > $O/r05_bench_sim$S.json 2> $O/r05_sim$S.err; show "rank 0 of $S simulated (stand-in exchange)" r05_bench_sim$S
done
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-next-rows --log-n 28 --n-domain-only --simulate-ranks 8 > $O/r05_bench_sim8_2p28_ndomain.json 2> $O/r05_sim8_28.err; show "2^28 n-domain, rank 0 of 8 simulated" r05_bench_sim8_2p28_ndomain
tail -3 $O/r05_multipath.err >> $T
cat $T
