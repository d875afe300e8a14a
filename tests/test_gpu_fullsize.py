"""BASELINE.json's full sizes, checked through size-independent properties (the oracle cannot run 2^24..2^27
in test time): transform round trips, agreement with the oracle on a strided sample computed through the
reference's own 2-D decomposition identity, and an exact MSM check that exploits tiled bases
(sum_i s_i P_{i mod u} = sum_j (sum_{i = j mod u} s_i) P_j, the distribution of dispatcher.rs:190-196)."""
import numpy as np
import pytest

from distributed_plonk_amd._ffi import MsmWorkload

pytestmark = pytest.mark.gpu


def _ints_mod_sum(sc, u, p):
    """Per residue class j (mod u): sum of the 256-bit scalars sc[i], i = j mod u, reduced mod p -> (u,4) limbs."""
    n = sc.shape[0]
    halves = sc.view(np.uint32).reshape(n // u, u, 8).astype(np.uint64).sum(axis=0)     # (u, 8) sums of 32-bit halves
    out = np.zeros((u, 4), dtype=np.uint64)
    for j in range(u):
        v = sum(int(halves[j, k]) << (32 * k) for k in range(8)) % p
        out[j] = [(v >> (64 * k)) & (2**64 - 1) for k in range(4)]
    return out


@pytest.mark.parametrize("curve,cid,log_n", [("bn254", 0, 24), ("bls12_381", 1, 22)])
def test_msm_full_size_tiled_bases_exact(gpu_workers, oracle, curve, cid, log_n):
    """configs[2] per-GPU scale (BN254 2^24) and configs[3] (BLS12-381 2^22)."""
    w = gpu_workers(curve)
    n, u = 1 << log_n, 1 << 11
    q = w.q64
    d_b = w.alloc(n * 16 * q)
    d_s = w.alloc(n * 32)
    d_c = w.alloc(n * 32)
    w.synth_bases(0xB0, u, n, d_b.ptr)
    w.synth_fr(0x5C, d_s.ptr, n)
    w.init_dev(d_b.ptr, n, 0, 0)
    # canonical scalars on the device (into_repr), as commit_polynomial does
    from distributed_plonk_amd._ffi import check
    sc_mont = d_s.download((n, 4))
    got = w.commit_dev(d_s.ptr, n)                                      # into_repr + MSM over all n bases
    p = int.from_bytes(oracle.field_const(cid, 0, 0).tobytes(), "little")
    sc = oracle.from_mont(cid, sc_mont)
    agg = _ints_mod_sum(sc, u, p)
    bases_u = oracle.gen_bases(cid, 0xB0, u, u)
    want = oracle.msm(cid, bases_u, agg, threads=8)
    g, gi = w.g1_to_affine(got)
    e, ei = oracle.jac_to_affine(cid, want)
    assert gi == ei and np.array_equal(g, e)
    for b in (d_b, d_s, d_c):
        b.free()


def distinct_bases_expected(oracle, cid, seed, sc, threads=8):
    """Exact expected point for the SRS-like distribution bench.py times by default — see oracle/checks.py."""
    from oracle import checks
    return checks.msm_expected_distinct(cid, seed, sc, threads=threads)


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("n", [300, (1 << 16) + 77])
def test_msm_distinct_bases_matches_oracle_directly(gpu_workers, oracle, curve, cid, n):
    """Pairwise-distinct bases (no P+P redo, the pure mixed-addition path) against the oracle's MSM on the SAME downloaded bases,
    and the A/B decomposition helper against both (pins the helper used at full size)."""
    w = gpu_workers(curve)
    q = w.q64
    d_b, d_s = w.alloc(n * 16 * q), w.alloc(n * 32)
    w.synth_bases(0x5EED, 0, n, d_b.ptr)
    w.synth_fr(0x77, d_s.ptr, n)
    w.init_dev(d_b.ptr, n, 0, 0)
    bases = d_b.download((n, 2 * q))
    assert len({bytes(r) for r in bases[: min(n, 5000)]}) == min(n, 5000)          # distinct indeed
    sc = oracle.from_mont(cid, d_s.download((n, 4)))
    got = w.commit_dev(d_s.ptr, n)
    g, gi = w.g1_to_affine(got)
    e, ei = oracle.jac_to_affine(cid, oracle.msm(cid, bases, sc, threads=8))
    assert gi == ei and np.array_equal(g, e)
    e2, ei2 = oracle.jac_to_affine(cid, distinct_bases_expected(oracle, cid, 0x5EED, sc))
    assert ei2 == ei and np.array_equal(e2, e)
    d_b.free(); d_s.free()


@pytest.mark.parametrize("curve,cid,log_n", [("bn254", 0, 24), ("bls12_381", 1, 22)])
def test_msm_full_size_distinct_bases_exact(gpu_workers, oracle, curve, cid, log_n):
    """the configuration bench.py times (`--bases distinct`, full bucket load through the fast mixed-addition
    kernel) compared with the oracle EXACTLY at BASELINE's sizes."""
    w = gpu_workers(curve)
    n, q = 1 << log_n, w.q64
    d_b, d_s = w.alloc(n * 16 * q), w.alloc(n * 32)
    w.synth_bases(0x5EED, 0, n, d_b.ptr)
    w.synth_fr(0xD15EA5E, d_s.ptr, n)
    w.init_dev(d_b.ptr, n, 0, 0)
    got = w.commit_dev(d_s.ptr, n)
    sc = oracle.from_mont(cid, d_s.download((n, 4)))
    g, gi = w.g1_to_affine(got)
    e, ei = oracle.jac_to_affine(cid, distinct_bases_expected(oracle, cid, 0x5EED, sc))
    assert gi == ei and np.array_equal(g, e)
    d_b.free(); d_s.free()


def test_msm_sliced_small_matches_oracle(gpu_workers, oracle):
    """The slicing path of msm_run (MSMs above 2^26 points are computed slice by slice) forced at 2^10-point slices:
    17 slices incl. a ragged last one, against the oracle directly."""
    w = gpu_workers("bn254")
    n = (1 << 14) + 5
    bases = oracle.gen_bases(0, 31, 500, n)
    sc = oracle.from_mont(0, oracle.rand_fr(0, 32, n))
    w.init(bases, 0, 0)
    w.set_option("msm_slice_log", 10)
    try:
        got = w.var_msm(MsmWorkload(0, n), sc)
    finally:
        w.set_option("msm_slice_log", 26)
    g, gi = w.g1_to_affine(got)
    e, ei = oracle.jac_to_affine(0, oracle.msm(0, bases, sc, threads=8))
    assert gi == ei and np.array_equal(g, e)


def test_msm_above_2p26_points_sliced_exact(gpu_workers, oracle):
    """configs[4] territory (2^28 gates over 8 GPUs = 2^25 per GPU; a single GPU can be handed more): n = 2^26 + 2^20 points
    takes the real two-slice path; exact through tiled bases."""
    w = gpu_workers("bn254")
    n, u, q = (1 << 26) + (1 << 20), 1 << 11, w.q64
    d_b, d_s = w.alloc(n * 16 * q), w.alloc(n * 32)
    w.synth_bases(0xB1, u, n, d_b.ptr)
    w.synth_fr(0x5D, d_s.ptr, n)
    w.init_dev(d_b.ptr, n, 0, 0)
    got = w.commit_dev(d_s.ptr, n)
    p = int.from_bytes(oracle.field_const(0, 0, 0).tobytes(), "little")
    sc = oracle.from_mont(0, d_s.download((n, 4)))
    want = oracle.msm(0, oracle.gen_bases(0, 0xB1, u, u), _ints_mod_sum(sc, u, p), threads=8)
    g, gi = w.g1_to_affine(got)
    e, ei = oracle.jac_to_affine(0, want)
    assert gi == ei and np.array_equal(g, e)
    d_b.free(); d_s.free()


def test_ntt_2p25_matches_oracle_everywhere(gpu_workers, oracle):
    """Full comparison against the oracle at 2^25 (passes 9,8,8): 33M outputs x 3 modes."""
    w = gpu_workers("bn254")
    v = oracle.rand_fr(0, 5, 1 << 25)
    for inv, coset in [(False, False), (False, True), (True, True)]:
        assert np.array_equal(w.ntt(v, inv, coset), oracle.ntt(0, v, inv, coset, threads=64)), (inv, coset)


def test_ntt_many_round_trips_2p20(gpu_workers):
    """64 different random vectors through coset NTT + inverse at 2^20 (67M elements in total)."""
    w = gpu_workers("bls12_381")
    N = 1 << 20
    a, b, c = w.alloc(N * 32), w.alloc(N * 32), w.alloc(N * 32)
    for seed in range(64):
        w.synth_fr(1000 + seed, a.ptr, N)
        w.memcpy_d2d(c.ptr, a.ptr, N * 32)
        w.ntt_dev(c.ptr, b.ptr, N, False, bool(seed & 1))
        w.ntt_dev(b.ptr, c.ptr, N, True, bool(seed & 1))
        assert np.array_equal(c.download((N, 4)), a.download((N, 4))), seed
    a.free(); b.free(); c.free()


@pytest.mark.parametrize("log_n", [24, 27])
def test_ntt_full_size_round_trip_and_sample(gpu_workers, oracle, log_n):
    """n = 2^24 and the quotient domain 8n = 2^27 (three 2^9 passes): coset round trip is the identity, and
    a decimated sub-transform matches the oracle:  for x supported on multiples of 2^k the size-N transform
    restricted to the first N/2^k outputs equals the size-N/2^k transform of the decimated input
    (playground.rs:100 uses the same zero-padding identity)."""
    w = gpu_workers("bn254")
    N = 1 << log_n
    a, b, c = w.alloc(N * 32), w.alloc(N * 32), w.alloc(N * 32)
    w.synth_fr(7, a.ptr, N)
    x = a.download((1 << 12, 4))                       # keep a prefix of the input
    w.memcpy_d2d(c.ptr, a.ptr, N * 32)
    w.ntt_dev(c.ptr, b.ptr, N, False, True)            # coset NTT (c destroyed)
    w.ntt_dev(b.ptr, c.ptr, N, True, True)             # coset iNTT
    # EVERY element must come back (a lazy-reduction slip shows up as a handful of wrong runs, not everywhere)
    CH = 1 << 22
    for off in range(0, N, CH):
        assert np.array_equal(c.download((CH, 4), byte_offset=off * 32), a.download((CH, 4), byte_offset=off * 32)), off
    w.memcpy_d2d(c.ptr, a.ptr, N * 32)
    w.ntt_dev(c.ptr, b.ptr, N, True, False)            # plain iNTT / NTT pair as well
    w.ntt_dev(b.ptr, c.ptr, N, False, False)
    for off in range(0, N, CH):
        assert np.array_equal(c.download((CH, 4), byte_offset=off * 32), a.download((CH, 4), byte_offset=off * 32)), off
    # zero-padded input: first M coefficients random, rest zero  =>  X[k * N/M] = NTT_M(x)[k]
    M = 1 << 12
    import ctypes as C
    from distributed_plonk_amd._ffi import check
    check(w.lib.plonk_dev_free(w.ctx, c.ptr)); c.ptr = None
    z = np.zeros((N, 4), dtype=np.uint64) if log_n <= 24 else None
    if z is not None:
        z[:M] = x
        a.upload(z)
        w.ntt_dev(a.ptr, b.ptr, N, False, False)
        full = b.download((N, 4))
        assert np.array_equal(full[:: N // M], oracle.ntt(0, x, False, False))
    a.free(); b.free()


def test_ntt_maximum_domain_2p28(gpu_workers):
    """BN254's largest radix-2 domain (two-adicity 28; configs[4] is an n = 2^28 transform): four 2^7 passes over
    8 GiB, every element must survive NTT -> iNTT; 2^29 is a DomainCreationError."""
    from distributed_plonk_amd._ffi import PlonkError
    w = gpu_workers("bn254")
    N = 1 << 28
    a, b, c = w.alloc(N * 32), w.alloc(N * 32), w.alloc(N * 32)
    w.synth_fr(28, a.ptr, N)
    w.memcpy_d2d(c.ptr, a.ptr, N * 32)
    w.ntt_dev(c.ptr, b.ptr, N, False, True)
    w.ntt_dev(b.ptr, c.ptr, N, True, True)
    CH = 1 << 23
    for off in range(0, N, CH):
        assert np.array_equal(c.download((CH, 4), byte_offset=off * 32), a.download((CH, 4), byte_offset=off * 32)), off
    with pytest.raises(PlonkError) as e:
        w.ntt_dev(a.ptr, b.ptr, 1 << 29, False, False)
    assert e.value.code == -2
    a.free(); b.free(); c.free()
