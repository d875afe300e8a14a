#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_quotient.py tests/test_gpu_ntt.py tests/test_gpu_coset_classes.py tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -6)
for wv in 4 3 2; do PLONK_QUOT_WAVES=$wv timeout 300 python tools/quotient_bench.py 24 2>&1 | grep quotient | sed "s/^/waves=$wv /"; done
PLONK_QUOT_WAVES=2 timeout 600 python -m pytest tests/test_gpu_quotient.py -m gpu -x -q 2>&1 | tail -2
PLONK_QUOT_WAVES=4 timeout 600 python -m pytest tests/test_gpu_quotient.py -m gpu -x -q 2>&1 | tail -2
echo "--- swizzle on"; timeout 300 python tools/coset_eval_only.py 24 2>&1 | grep coset_eval; timeout 300 python tools/ntt_only.py 24 27 2>&1 | grep NTT
echo "--- swizzle off"; PLONK_NTT_NO_SWIZZLE=1 timeout 300 python tools/coset_eval_only.py 24 2>&1 | grep coset_eval; PLONK_NTT_NO_SWIZZLE=1 timeout 300 python tools/ntt_only.py 24 27 2>&1 | grep NTT
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "ntt" 2>&1 | tail -3)
