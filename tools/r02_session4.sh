#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_quotient.py tests/test_gpu_coset_classes.py -m gpu -x -q 2>&1 | tail -6)
timeout 600 python tools/quotient_bench.py 24 2>&1 | grep quotient_fuse | tee $O/quotient_variants.txt
(timeout 900 python -m pytest tests/test_gpu_prover.py tests/test_gpu_class_prover.py -m gpu -x -q 2>&1 | tail -4)
