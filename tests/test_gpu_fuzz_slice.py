"""A fixed-seed slice of the differential ABI fuzzer (tools/fuzz_abi.py) against the REAL library on an MI355X: 300 random operations of
all nineteen kinds (single kernels, plonk_trim between operations, the gate-range grand product, the residue-class iFFT, refused SRSs, distributed transforms, batched commitments, fixed-base tables, whole proofs handed to the verifier) with
random shapes, flags and options, every result compared with the CPU oracle bit for bit.  Round 3 could only run the fuzzer against the host
emulation (tests/test_hostemu.py); its first run on the device (round 4, gpurun: 1651 operations in 100 s, no mismatch — profiles/
r04_opening_measurements.txt) is pinned here as a test.  The oracle is the checker; the product path is the C ABI."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_differential_fuzz_slice_on_the_device():
    r = subprocess.run([sys.executable, "tools/fuzz_abi.py", "--seconds", "240", "--max-ops", "300", "--seed", "2026", "--max-log", "13"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz ok: 300 operations" in r.stdout, (r.stdout + r.stderr)[-3000:]
