"""The host side of distributed_plonk_amd.prover (round sequencing, challenge points, linearisation coefficients, the 6-coset
Vandermonde reconstruction, two-lane commitments, Fiat-Shamir) on CPU: the device calls are served by the oracle through the
test-only stand-in tests/cpu_worker.py, so a mistake in the orchestration shows up without a GPU.  The same flows run on the
real library in tests/test_gpu_prover.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cpu_worker import CpuWorker  # noqa: E402

from distributed_plonk_amd.prover import Prover, WrongQuotientPolyDegree  # noqa: E402
from distributed_plonk_amd.transcript import serialize_proof  # noqa: E402


def _instance(oracle, cid, log_n, seed):
    from oracle import prover_ref as P
    n = 1 << log_n
    circ = P.make_circuit(cid, log_n, seed=seed)
    ck, inf = P.make_ck(cid, n, seed=seed + 1, unique=8)
    bl = dict(wires=oracle.rand_fr(cid, seed + 2, 10).reshape(5, 2, 4), perm=oracle.rand_fr(cid, seed + 3, 3))
    return P, circ, ck, inf, bl


def _same(a, b):
    return a[1] == b[1] and np.array_equal(a[0], b[0])


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("mode,cache,two_lanes", [("coset8n", False, False), ("coset8n", True, True), ("classes6", False, True), ("classes6", True, False)])
def test_prover_rounds_on_cpu_stand_in(oracle, curve, cid, mode, cache, two_lanes):
    log_n = 4
    n = 1 << log_n
    P, circ, ck, inf, bl = _instance(oracle, cid, log_n, 1234)
    w = CpuWorker(curve)
    w.init(ck, n, 8 * n)
    helper = None
    if two_lanes:                       # shares the arena: both "contexts" see the same device memory, as on a GPU
        helper = CpuWorker(curve, share=w)
        helper.init(ck, n, 8 * n)
    pv = Prover(w, log_n, cache_key_cosets=cache, quotient_mode=mode, commit_helper=helper)
    pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
    fs = pv.fiat_shamir(circ["pub_input"][:2])
    got = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, fs, keep=True)
    want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, fs.drawn)
    for key in ("wires_poly_comms", "split_quot_poly_comms"):
        assert all(_same(g, x) for g, x in zip(got[key], want[key])), key
    for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
        assert _same(got[key], want[key]), key
    for key in ("wires_evals", "wire_sigma_evals"):
        assert np.array_equal(np.stack(got[key]), np.stack(want[key])), key
    assert np.array_equal(got["perm_next_eval"], want["perm_next_eval"])
    for key in ("perm_product", "perm_poly", "quot_poly", "lin_poly", "batch_poly"):
        assert np.array_equal(got["_debug"][key], want[key]), key
    # proof bytes: 13 compressed points, 10 field elements, 4 length prefixes
    q_bytes = 32 if curve == "bn254" else 48
    assert len(serialize_proof(curve, got)) == 13 * q_bytes + 10 * 32 + 4 * 8
    # a second proof with a broken witness is rejected by the degree check
    bad = circ["wires"].copy()
    bad[4, 2] = oracle.rand_fr(cid, 99, 1)[0]
    with pytest.raises(WrongQuotientPolyDegree):
        pv.prove(bad, circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, pv.fiat_shamir(circ["pub_input"][:2]))
    pv.close()


def test_six_coset_vandermonde_is_inverted(oracle):
    """V[s][u] = c_s^u with c_s = (g w_m^s)^n; the host-side inverse used by quotient_mode="classes6"."""
    w = CpuWorker("bn254")
    pv = Prover(w, 6, quotient_mode="classes6")
    cls = pv._class_setup()
    f, n, m, p = pv.f, pv.n, pv.m, pv.f.p
    w_m = f.root_of_unity(m)
    c = [pow(f.generator * pow(w_m, s, p) % p, n, p) for s in range(6)]
    vinv = [[f.from_limbs(x) for x in row] for row in cls["vinv"]]
    for u in range(6):
        for t in range(6):
            assert sum(vinv[u][s] * pow(c[s], t, p) for s in range(6)) % p == (1 if u == t else 0)
    assert len(set(c)) == 6
    with pytest.raises(ValueError):
        Prover(w, 3, quotient_mode="classes6")


@pytest.mark.parametrize("S", [1, 2, 4])
def test_dispatcher_call_sequence_on_cpu_stand_in(oracle, S):
    """distributed_plonk_amd.dispatcher.Dispatcher — the reference's `Prover::fft` sequence (dispatcher2.rs:732-787: fft_init on
    every worker, one fft1 per decimated row, fft2_prepare with the block exchange, fft2, host undecimate), its sharded MSM
    (dispatcher.rs:218-238) and commit_polynomial (dispatcher2.rs:835-893) — with S in-process stand-in workers."""
    from distributed_plonk_amd.dispatcher import Dispatcher
    cid, n = 0, 1 << 6
    first = CpuWorker("bn254")
    workers = [first] + [CpuWorker("bn254", share=first) for _ in range(S - 1)]
    d = Dispatcher(workers)
    bases = oracle.gen_bases(cid, 3, 16, n)
    d.init(bases, n, 8 * n)
    v = oracle.rand_fr(cid, 5, n)
    for is_quot, is_inv, is_coset in [(False, True, False), (True, False, True), (True, True, True), (False, False, False)]:
        N = 8 * n if is_quot else n
        x = np.zeros((N, 4), dtype=np.uint64)
        x[:n] = v
        assert np.array_equal(d.fft(v, is_quot, is_inv, is_coset), oracle.ntt(cid, x, is_inv, is_coset)), (is_quot, is_inv, is_coset)
    sc = oracle.from_mont(cid, oracle.rand_fr(cid, 6, n))
    got = oracle.jac_to_affine(cid, d.msm(sc))
    want = oracle.jac_to_affine(cid, oracle.msm(cid, bases, sc))
    assert got[1] == want[1] and np.array_equal(got[0], want[0])
    poly = oracle.rand_fr(cid, 7, n - 5)
    xy, isinf = d.commit_polynomial(poly)
    want = oracle.jac_to_affine(cid, oracle.commit_polynomial(cid, bases, poly))
    assert isinf == bool(want[1]) and np.array_equal(xy, want[0])


def test_bench_dry_run_plans_every_rank_count():
    """bench.py --dry-run (tools/preflight_multi.sh): no GPU, validates divisibility / sizes for the rank counts the scaling run uses,
    and refuses what the library would refuse (the 8n domain of a 2^28-gate BN254 circuit does not exist)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for gpus in (1, 2, 4, 8):
        for scheme in ("classes", "reference2d"):
            res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--scheme", scheme, "--dry-run"],
                                 capture_output=True, text=True, timeout=120)
            assert res.returncode == 0, res.stderr
            plan = json.loads(res.stdout.strip().splitlines()[-1])
            assert plan["ok"] and plan["ranks"] == gpus and plan["msm_points_per_rank"] == (1 << 24) // gpus
            assert plan["transforms"]["8n"]["r"] % gpus == 0 and plan["class_points_per_rank"] == (8 << 24) // gpus
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--log-n", "28", "--dry-run"], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2 and "two-adicity" in bad.stdout
    odd = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--dry-run"], capture_output=True, text=True, timeout=120)
    assert odd.returncode == 2
