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


# ---- the five prover rounds in compiled code: host/plonk_prover.hpp (Prover::prove, dispatcher2.rs:192-713, with its merlin transcript) ---------
def _build_prover(tmp_path):
    from distributed_plonk_amd import _ffi
    _ffi.lib()
    exe = str(tmp_path / "prover_check")
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "host"),
           os.path.join(ROOT, "tests", "host_cpp", "prover_check.cpp"), "-o", exe, "-L" + LIBDIR, "-lplonk_hip", "-ldl", "-Wl,-rpath," + LIBDIR]
    subprocess.check_call(cmd)
    return exe


def check_cpp_prover(tmp_path, exe, curve, cid, log_n, seed, env=None):
    """Writes a satisfied circuit + SRS for the compiled prover, runs it, and checks EVERYTHING it returned: the verifying key against oracle
    commitments, the five challenges against the Python transcript replayed over the proof's own commitments and evaluations (so both
    transcripts absorbed identical bytes), the proof against the oracle's restatement of the rounds fed with those challenges, and the
    serialized proof against transcript.serialize_proof."""
    import numpy as np
    from distributed_plonk_amd.prover import FiatShamir
    from distributed_plonk_amd.transcript import PlonkTranscript, serialize_proof
    from oracle import oracle as O, prover_ref as P
    n = 1 << log_n
    Q = O.FQ_LIMBS[cid]
    circ = P.make_circuit(cid, log_n, seed=seed)
    ck, inf = P.make_ck(cid, n, seed=seed + 1, unique=min(64, n))
    bl = dict(wires=O.rand_fr(cid, seed + 2, 10).reshape(5, 2, 4), perm=O.rand_fr(cid, seed + 3, 3))
    num_inputs = 2
    parts = [np.array([cid, log_n, ck.shape[0], num_inputs], dtype=np.uint64), ck, circ["selectors"], circ["sigmas"], circ["k"], circ["wires"],
             circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl["wires"], bl["perm"]]
    fin, fout = str(tmp_path / f"in_{curve}.bin"), str(tmp_path / f"out_{curve}.bin")
    with open(fin, "wb") as fh:
        for a in parts:
            fh.write(np.ascontiguousarray(a, dtype="<u8").tobytes())
    res = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and "prover_check ok" in res.stdout, (res.stdout + res.stderr)[-2000:]
    out = np.fromfile(fout, dtype="<u8")
    pos = [0]

    def pt():
        xy, flag = out[pos[0]:pos[0] + 2 * Q].copy(), bool(out[pos[0] + 2 * Q])
        pos[0] += 2 * Q + 1
        return xy, flag

    def fr():
        v = out[pos[0]:pos[0] + 4].copy()
        pos[0] += 4
        return v

    sel_c, sig_c = [pt() for _ in range(13)], [pt() for _ in range(5)]
    proof = {"wires_poly_comms": [pt() for _ in range(5)], "prod_perm_poly_comm": pt()}
    n_split = int(out[pos[0]]); pos[0] += 1
    proof["split_quot_poly_comms"] = [pt() for _ in range(n_split)]
    proof["opening_proof"], proof["shifted_opening_proof"] = pt(), pt()
    ch = {k: fr() for k in ("beta", "gamma", "alpha", "zeta", "v")}
    proof["wires_evals"], proof["wire_sigma_evals"], proof["perm_next_eval"] = [fr() for _ in range(5)], [fr() for _ in range(4)], fr()
    n_ser = int(out[pos[0]]); pos[0] += 1
    ser = out[pos[0]:].tobytes()[:n_ser]
    same = lambda a, b: a[1] == b[1] and np.array_equal(a[0], b[0])
    # (1) the verifying key: 18 commitments against the oracle
    for j, poly in enumerate(list(circ["selectors"]) + list(circ["sigmas"])):
        want = O.jac_to_affine(cid, O.commit_polynomial(cid, ck, poly, inf=inf, threads=8))
        assert same((sel_c + sig_c)[j], want), ("vk", j)
    # (2) the challenges: the Python transcript over the C++ proof's own commitments / evaluations draws the same five field elements
    t = PlonkTranscript(curve)
    t.append_vk_and_pub_input(n, num_inputs, list(circ["k"]), sel_c, sig_c, list(circ["pub_input"][:num_inputs]))
    fs = FiatShamir(t)
    for label in ("beta", "gamma", "alpha", "zeta", "v"):
        assert np.array_equal(fs(label, proof), ch[label]), label
    # (3) the proof: the oracle's rounds with those challenges
    want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, ch, threads=8)
    assert n_split == 5
    for key in ("wires_poly_comms", "split_quot_poly_comms"):
        for g, x in zip(proof[key], want[key]):
            assert same(g, x), key
    for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
        assert same(proof[key], want[key]), key
    for key in ("wires_evals", "wire_sigma_evals"):
        assert np.array_equal(np.stack(proof[key]), np.stack(want[key])), key
    assert np.array_equal(proof["perm_next_eval"], want["perm_next_eval"])
    # (4) the serialization: byte for byte what transcript.serialize_proof gives for the same proof
    assert ser == serialize_proof(curve, proof)


def test_cpp_prover_builds_against_the_c_abi_only(tmp_path):
    """host/plonk_prover.hpp compiles with plain g++ -Wall -Werror against include/plonk_hip.h alone (no HIP headers, no Python)."""
    _build_prover(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("curve,cid,log_n", [("bn254", 0, 9), ("bls12_381", 1, 6)])
def test_cpp_prover_matches_oracle_and_python_transcript(tmp_path, curve, cid, log_n):
    check_cpp_prover(tmp_path, _build_prover(tmp_path), curve, cid, log_n, seed=1200 + log_n)
