#!/bin/bash
# Round 6, sixth GPU call: the scaling picture one GPU can give, with the PROOF as headline — rank 0's share of the class prover's proof on 2 / 4 / 8
# simulated ranks (stand-in exchange) beside the single-GPU proof of the same lease; configs[4] (2^28 gates, n-domain part only: two-adicity 28) on one GPU
# and as rank 0 of 8; configs[3] (BLS12-381 2^22) and configs[1] (2^20) as stand-alone driver-style lines.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
cd $R
T=$O/r06_call6.txt
: > $T
C="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-poly-parallel"
show() {
python - "$1" $O/$2.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    ph = {a: b for a, b in (d.get("phases_ms") or {}).items() if a != "note"}
    print(f"{sys.argv[1]:34s} {d['ms_per_step']:10.3f} ms per step  value {d['value']:.4g}  op-mix {d.get('op_mix_ms_per_step')}  verified {d.get('verified')} {d.get('prover_verified')}  | {str(d.get('headline'))[:70]} | {ph} err {d.get('proof_headline_error')}")
except Exception as ex:
    print(f"{sys.argv[1]:34s} FAILED: {ex!r}")
PY
}
timeout 600 python bench.py $C --next-rows proof > $O/r06_scal_n1.json 2> $O/r06_call6.err; show "one GPU" r06_scal_n1
for S in 2 4 8; do
  timeout 600 python bench.py $C --simulate-ranks $S > $O/r06_bench_sim$S.json 2>> $O/r06_call6.err; show "rank 0 of $S, stand-in exchange" r06_bench_sim$S
done
timeout 900 python bench.py --log-n 28 --n-domain-only --steps 3 --warmup 1 > $O/r06_bench_bn254_2p28_ndomain_1gpu.json 2>> $O/r06_call6.err; show "2^28 n-domain, one GPU" r06_bench_bn254_2p28_ndomain_1gpu
timeout 900 python bench.py --log-n 28 --n-domain-only --steps 3 --warmup 1 --simulate-ranks 8 > $O/r06_bench_sim8_2p28_ndomain.json 2>> $O/r06_call6.err; show "2^28 n-domain, rank 0 of 8" r06_bench_sim8_2p28_ndomain
timeout 600 python bench.py --log-n 22 --curve bls12_381 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/r06_bench_bls12_381_2p22.json 2>> $O/r06_call6.err; show "BLS12-381 2^22 (configs[3])" r06_bench_bls12_381_2p22
timeout 600 python bench.py --log-n 20 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/r06_bench_bn254_2p20.json 2>> $O/r06_call6.err; show "BN254 2^20 (configs[1])" r06_bench_bn254_2p20
cat $T
