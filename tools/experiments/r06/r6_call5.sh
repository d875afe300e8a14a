#!/bin/bash
# Round 6, fifth GPU call: the compiled C++ prover (host/plonk_prover.hpp through tests/host_cpp/prover_check.cpp) on the device, both curves; the ABI
# fuzzer with its nineteenth operation kind (plonk_trim between operations) on the real library.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
cd $R
T=$O/r06_call5.txt
: > $T
timeout 900 python -m pytest -m gpu -q -x -p no:cacheprovider tests/test_host_cpp.py > $O/r06_call5_tests.txt 2>&1
grep -E "passed|failed|error" $O/r06_call5_tests.txt | tail -2 | tee -a $T
timeout 500 python tools/fuzz_abi.py --seconds 240 --seed 606 --max-log 13 2>&1 | grep -E "fuzz ok|MISMATCH" | tee -a $T
timeout 200 python tools/fuzz_abi.py --seconds 40 --seed 607 --max-log 16 --only trim 2>&1 | grep -E "fuzz ok|MISMATCH" | tee -a $T
timeout 300 python tools/fuzz_abi.py --seconds 120 --seed 608 --max-log 14 --curve bls12_381 2>&1 | grep -E "fuzz ok|MISMATCH" | tee -a $T
cat $T
