"""The host side in compiled code: host/plonk_host.hpp is a C++ mirror of the reference worker / dispatcher orchestration on the bare
C ABI (the reference's host is Rust; no Rust toolchain here — ffi/plonk_hip.rs is that binding as source).  tests/host_cpp/host_check.cpp
drives it — distributed FFT in all four modes on S = 1, 2, 4 in-process workers, sharded MSM, commit_polynomial, a prover round through
plonk_commit_many_dev on every worker's key range — and compares with
the oracle, with no Python between the host program and libplonk_hip.so."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "distributed_plonk_amd", "lib")


def _build(tmp_path):
    from distributed_plonk_amd import _ffi
    _ffi.lib()                                          # the library must exist (fails loudly otherwise)
    exe = str(tmp_path / "host_check")
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "host"),
           os.path.join(ROOT, "tests", "host_cpp", "host_check.cpp"), "-o", exe, "-L" + LIBDIR, "-lplonk_hip", "-ldl", "-Wl,-rpath," + LIBDIR]
    subprocess.check_call(cmd)
    return exe


def test_cpp_host_builds_against_the_c_abi_only(tmp_path):
    """No GPU needed: plain g++ (not hipcc) compiles the host mirror against include/plonk_hip.h and links libplonk_hip.so; without a
    device the program reports the library's error instead of falling back to anything."""
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    res = subprocess.run([exe, os.path.join(ROOT, "oracle", "libplonk_oracle.so"), "0"], capture_output=True, text=True)
    assert res.returncode == 1 and "plonk error -3" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [0, 1])
def test_cpp_host_matches_oracle(tmp_path, curve):
    exe = _build(tmp_path)
    from oracle import oracle as O
    O.lib()                                             # make sure the oracle library is built
    res = subprocess.run([exe, os.path.join(ROOT, "oracle", "libplonk_oracle.so"), str(curve)], capture_output=True, text=True, timeout=600)
    sys.stdout.write(res.stdout)
    sys.stderr.write(res.stderr)
    assert res.returncode == 0 and "host_check ok" in res.stdout
