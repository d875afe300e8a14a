#!/bin/bash
# First GPU call of round 5 (≈ 3 GPU-minutes): times what round 4 built after its closing collection and could not time
# (profiles/README.md): the phases of a step overlapped at 2^20 / BLS12-381 2^22 / simulated 8 ranks, and
# Prover(fft_helper=...) at the same sizes.  Every pair runs in ONE lease, alternating; only same-call pairs are comparable.
#   gpurun --timeout 600 -- 'bash tools/round5_opening.sh'      -> gpurun_out/r05_opening.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=$O/r05_opening.txt
: > $T
C="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
one() {        # label, file, command...
  local label=$1 f=$2; shift 2
  timeout 200 "$@" > $O/$f.json 2> $O/$f.err
  python - "$label" $O/$f.json >> $T <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
    pr = (d.get("next_rows") or {}).get("prover_rounds") or {}
    var = {k: (v.get("ms"), v.get("same_proof_as_the_verified_one")) for k, v in (pr.get("variants") or {}).items()}
    print(f"{sys.argv[1]:44s} step {d.get('ms_per_step')} ms  phases {d.get('phases_ms', {}).get('transforms')} / {d.get('phases_ms', {}).get('commitments')}"
          f"  overlap {d['config'].get('phase_overlap')}  verified {d.get('verified')}  proof {d.get('proof_ms')} ms {d.get('prover_verified')}  variants {var}")
except Exception as ex:
    print(f"{sys.argv[1]:44s} FAILED: {ex!r}")
PY
}
# N = 1 without torch (tools/bench_notorch.py: the same program, a minute less start-up per process on a fresh box)
for rep in 1 2; do
  one "2^20 BN254, phases apart ($rep)"      r05_bn20_off_$rep  python tools/bench_notorch.py --log-n 20 $C --next-rows proof --overlap-phases off
  one "2^20 BN254, phases overlapped ($rep)" r05_bn20_on_$rep   python tools/bench_notorch.py --log-n 20 $C --next-rows proof --overlap-phases on
done
one "2^22 BLS12-381, phases apart"      r05_bls22_off python tools/bench_notorch.py --log-n 22 --curve bls12_381 $C --next-rows proof --overlap-phases off
one "2^22 BLS12-381, phases overlapped" r05_bls22_on  python tools/bench_notorch.py --log-n 22 --curve bls12_381 $C --next-rows proof --overlap-phases on
# rank 0 of 8 simulated on one GPU (bench.py proper: this path creates a world-1 process group)
S="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-next-rows --no-class-prover --no-poly-parallel --simulate-ranks 8"
one "2^24 BN254, rank 0 of 8 simulated, apart"      r05_sim8_off python bench.py $S --overlap-phases off
one "2^24 BN254, rank 0 of 8 simulated, overlapped" r05_sim8_on  python bench.py $S --overlap-phases on
cat $T
