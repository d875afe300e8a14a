"""Whole-vector NTT (plonk_ntt / plonk_ntt_dev) vs the oracle: bit-exact Montgomery limbs.

Mirrors the shape of the reference's tests: 4 modes x several domain sizes incl. odd log N
(dispatcher.rs:246-350 uses 2^11 and 2^13; playground.rs:82-103 uses 512 plus the zero-padding and
round-trip identities)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = [(False, False), (True, False), (False, True), (True, True)]


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n", [1, 2, 3, 4, 5, 9, 10, 11, 13, 16])
def test_ntt_matches_oracle(gpu_workers, oracle, curve, cid, log_n):
    w = gpu_workers(curve)
    v = oracle.rand_fr(cid, 1000 + log_n, 1 << log_n)
    for inv, coset in MODES:
        got = w.ntt(v, inv, coset)
        want = oracle.ntt(cid, v, inv, coset, threads=4)
        assert np.array_equal(got, want), f"{curve} 2^{log_n} inv={inv} coset={coset}"


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n", [9, 10, 18])
def test_ntt_extreme_inputs_and_both_butterfly_forms(gpu_workers, oracle, curve, cid, log_n):
    """The lazy bounds of the pass kernel at their top (ntt_kernels.hpp: values grow 4p per stage to < 36p with the precomputed-
    quotient butterflies, which BLS12-381's Fr uses since round 4): inputs of all p - 1, alternating 0 / p - 1 and a single p - 1 among zeros,
    through a full 2^9-row pass (log_n = 9), a two-pass plan (10) and two full passes (18), every mode, with the Shoup butterflies (the
    default) and with the Montgomery ones (option ntt_shoup = 0) — bit-exact against the oracle both ways."""
    w = gpu_workers(curve)
    n = 1 << log_n
    pm1 = oracle.field_const(cid, 0, 0) - np.array([1, 0, 0, 0], dtype=np.uint64)       # p - 1: self-inverse under the Montgomery map up to sign, any fixed residue will do
    rnd = oracle.rand_fr(cid, 4242 + log_n, n)
    cases = []
    a = np.tile(pm1, (n, 1)); cases.append(a)
    b = np.zeros((n, 4), dtype=np.uint64); b[::2] = pm1; cases.append(b)
    c = np.zeros((n, 4), dtype=np.uint64); c[n - 1] = pm1; cases.append(c)
    d = rnd.copy(); d[: n // 2] = pm1; cases.append(d)
    try:
        for shoup in (1, 0):
            w.set_option("ntt_shoup", shoup)
            for v in cases:
                for inv, coset in MODES:
                    assert np.array_equal(w.ntt(v, inv, coset), oracle.ntt(cid, v, inv, coset, threads=8)), (curve, log_n, shoup, inv, coset)
    finally:
        w.set_option("ntt_shoup", 1)


@pytest.mark.parametrize("log_n", [20, 21])
def test_ntt_large_three_pass(gpu_workers, oracle, log_n):
    w = gpu_workers("bn254")
    v = oracle.rand_fr(0, 77, 1 << log_n)
    for inv, coset in [(False, True), (True, True)]:
        got = w.ntt(v, inv, coset)
        want = oracle.ntt(0, v, inv, coset, threads=8)
        assert np.array_equal(got, want)


def test_playground_identities(gpu_workers, oracle):
    """playground.rs:100-102."""
    w = gpu_workers("bls12_381")
    l = 512
    exps = oracle.rand_fr(1, 5, l)
    t = np.zeros((2 * l, 4), dtype=np.uint64)
    t[:l] = exps
    assert np.array_equal(w.ntt(t, False, True), oracle.ntt(1, t, False, True))      # zero padding
    assert np.array_equal(w.ntt(w.ntt(exps, False, True), True, True), exps)         # coset round trip
    assert np.array_equal(w.ntt(w.ntt(exps, False, False), True, False), exps)


def test_domain_errors(gpu_workers):
    from distributed_plonk_amd._ffi import PlonkError
    w = gpu_workers("bn254")
    with pytest.raises(PlonkError) as e:
        w.ntt(np.zeros((3, 4), dtype=np.uint64))
    assert e.value.code == -2
    with pytest.raises(PlonkError):
        w.init(None, 1 << 29, 0)          # BN254 two-adicity is 28


def test_transpose(gpu_workers, oracle):
    w = gpu_workers("bn254")
    for rows, cols in [(32, 64), (64, 32), (48, 80), (1, 7)]:
        v = oracle.rand_fr(0, rows * cols, rows * cols)
        got = w.transpose(v, rows, cols)
        want = v.reshape(rows, cols, 4).transpose(1, 0, 2).reshape(-1, 4)
        assert np.array_equal(got, want)
