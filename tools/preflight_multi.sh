#!/bin/bash
# Pre-flight for the driver's multi-GPU runs (no GPU needed): for every rank count the scaling run uses, check that bench.py accepts
# the arguments, that r / c / n divide, and print the per-rank sizes; then run the world-size-2 CPU tests of the N > 1 path (gloo).
#   bash tools/preflight_multi.sh
set -e
cd "$(dirname "$0")/.."
for cfg in "--log-n 24 --curve bn254" "--log-n 20 --curve bn254" "--log-n 22 --curve bls12_381"; do
  for n in 1 2 4 8; do
    echo "== bench.py --gpus $n $cfg --dry-run"
    python bench.py --gpus $n $cfg --dry-run
    python bench.py --gpus $n $cfg --scheme reference2d --dry-run > /dev/null
  done
done
echo "== configs[4]: 2^28-gate BN254, n-domain part only"
python bench.py --gpus 8 --log-n 28 --n-domain-only --dry-run
echo "== 2^28-gate BN254 with the quotient domain must be refused: the 8n domain does not exist"
if python bench.py --gpus 8 --log-n 28 --dry-run; then echo "unexpected: accepted"; exit 1; fi
echo "== world_size 2 / 4 CPU tests of the multi-rank path (gloo)"
python -m pytest tests/test_gloo_multirank.py -q -x
echo "preflight ok"
