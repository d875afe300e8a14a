#!/bin/bash
# Round 6, ninth GPU call: 4-column tiles for the 2^8- and 2^9-row passes (ntt_pass_kernel<.., SWZ = 2>: half the tile, twice the workgroups per CU — four / two
# independent barrier groups instead of two / one; PLONK_NTT_LOGT8 / LOGT9 = 2) against the shipped 8-column tiles, alternating processes in ONE lease:
# the 8n coset FFT from n + 3 coefficients (three <8> passes), the dense 2^27-point transform (three <9> passes), then the proof.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
cd $R
T=$O/r06_call9.txt
: > $T
timeout 600 env PLONK_NTT_LOGT8=2 PLONK_NTT_LOGT9=2 python -m pytest -m gpu -q -x -p no:cacheprovider tests/test_gpu_ntt.py tests/test_gpu_coset_classes.py 2>&1 | grep -E "passed|failed" | tail -1 | tee -a $T
for rep in 1 2 3; do
  timeout 200 python tools/coset_eval_only.py 24 2>/dev/null | tail -1 | tee -a $T
  PLONK_NTT_LOGT8=2 timeout 200 python tools/coset_eval_only.py 24 2>/dev/null | tail -1 | tee -a $T
done
for rep in 1 2; do
  timeout 200 python tools/ntt_only.py 27 2>/dev/null | tail -1 | sed 's/^/LOGT9=- /' | tee -a $T
  PLONK_NTT_LOGT9=2 timeout 200 python tools/ntt_only.py 27 2>/dev/null | tail -1 | sed 's/^/LOGT9=2 /' | tee -a $T
done
C="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --next-rows proof"
for rep in 1 2; do
  for v in - 2; do
    if [ $v = - ]; then E=""; else E="PLONK_NTT_LOGT8=2 PLONK_NTT_LOGT9=2"; fi
    env $E timeout 400 python bench.py $C > $O/r06_t4_$v_$rep.json 2> $O/r06_call9.err
    python - "$v" $O/r06_t4_$v_$rep.json <<'PY' | tee -a $T
import json, sys
d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
print(f"proof, tile columns {'8' if sys.argv[1] == '-' else '4'}: {d['ms_per_step']} ms  op-mix {d['op_mix_ms_per_step']}  ntt avg {d['roofline']['avg_launch_ms']} frac {d['roofline']['frac']}  verified {d['verified']} {d.get('prover_verified')}")
PY
  done
done
