"""plonk_commit_many_dev: the independent commitments of a prover round (dispatcher2.rs:313-321 five wires, :519-531 five quotient
parts, :690-697 two openings) as ONE Pippenger problem.  Every point must equal the one-at-a-time commit_polynomial
(dispatcher2.rs:835-893) bit for bit, and the oracle's MSM on the same downloaded operands."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(w, n, u=0, seed=0xC0DE):
    q = w.q64
    d_b = w.alloc(n * 16 * q)
    w.synth_bases(seed, u, n, d_b.ptr)
    w.init_dev(d_b.ptr, n, 0, 0)
    return d_b


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("n,lens", [(700, [700, 1, 0, 699, 333]), ((1 << 14) + 5, [(1 << 14) + 5, 1 << 14, 9000]), (64, [64])])
def test_commit_many_matches_oracle_and_single_commits(gpu_workers, oracle, curve, cid, n, lens):
    """Ragged lengths (a zero-length polynomial included) against the oracle's MSM on the same bases and scalars."""
    w = gpu_workers(curve)
    q = w.q64
    d_b = _setup(w, n)
    bases = d_b.download((n, 2 * q))
    polys = []
    for i, ln in enumerate(lens):
        d = w.alloc(max(ln, 1) * 32)
        w.synth_fr(0x900 + i, d.ptr, max(ln, 1))
        polys.append(d)
    got = w.commit_many_dev([(d.ptr, ln) for d, ln in zip(polys, lens)])
    assert got.shape == (len(lens), 3 * q)
    for i, (d, ln) in enumerate(zip(polys, lens)):
        g, gi = w.g1_to_affine(got[i])
        one = w.commit_dev(d.ptr, ln)
        o, oi = w.g1_to_affine(one)
        assert gi == oi and np.array_equal(g, o), f"polynomial {i}: batched != single"
        if ln:
            sc = oracle.from_mont(cid, d.download((max(ln, 1), 4))[:ln])
            e, ei = oracle.jac_to_affine(cid, oracle.msm(cid, bases[:ln], sc, threads=4))
        else:
            e, ei = None, True
        assert gi == ei and (ei or np.array_equal(g, e)), f"polynomial {i}: != oracle"
    for d in polys:
        d.free()
    d_b.free()


def test_commit_many_with_start_offset_and_clamping(gpu_workers, oracle):
    """`start` selects the key range (ClassProver's shard); lengths beyond the resident key are clamped like commit_polynomial."""
    w = gpu_workers("bn254")
    n = 5000
    d_b = _setup(w, n)
    d = [w.alloc(n * 32) for _ in range(3)]
    for i, x in enumerate(d):
        w.synth_fr(0x41 + i, x.ptr, n)
    start = 1234
    lens = [n - start, 100, n]            # the last is clamped to n - start
    got = w.commit_many_dev([(x.ptr + start * 32, ln) for x, ln in zip(d, lens)], start=start)
    for i, (x, ln) in enumerate(zip(d, lens)):
        one = w.commit_range_dev(x.ptr + start * 32, start, min(ln, n - start))
        g, gi = w.g1_to_affine(got[i])
        o, oi = w.g1_to_affine(one)
        assert gi == oi and np.array_equal(g, o)
    for x in d:
        x.free()
    d_b.free()


def test_commit_many_repeated_bases_and_identical_vectors(gpu_workers, oracle):
    """Tiled bases (P + P and P - P inside buckets: the redo path) and the same vector five times -> five equal points."""
    w = gpu_workers("bn254")
    n, u = 1 << 15, 7
    d_b = _setup(w, n, u=u)
    d = w.alloc(n * 32)
    w.synth_fr(0xAB, d.ptr, n)
    got = w.commit_many_dev([(d.ptr, n)] * 5)
    one = w.commit_dev(d.ptr, n)
    o, oi = w.g1_to_affine(one)
    for i in range(5):
        g, gi = w.g1_to_affine(got[i])
        assert gi == oi and np.array_equal(g, o)
    d.free()
    d_b.free()


def test_commit_many_group_limit_and_slices(gpu_workers):
    """msm_batch_max = 2 splits five vectors into groups of 2 + 2 + 1; msm_slice_log = 10 additionally slices the points:
    same five points."""
    w = gpu_workers("bn254")
    n = 6000
    d_b = _setup(w, n)
    d = [w.alloc(n * 32) for _ in range(5)]
    for i, x in enumerate(d):
        w.synth_fr(0x70 + i, x.ptr, n)
    items = [(x.ptr, n - 7 * i) for i, x in enumerate(d)]
    ref = w.commit_many_dev(items)
    try:
        w.set_option("msm_batch_max", 2)
        a = w.commit_many_dev(items)
        w.set_option("msm_slice_log", 10)
        b = w.commit_many_dev(items)
    finally:
        w.set_option("msm_batch_max", 32)
        w.set_option("msm_slice_log", 26)
    for i in range(5):
        r = w.g1_to_affine(ref[i])
        for other in (a, b):
            o = w.g1_to_affine(other[i])
            assert r[1] == o[1] and np.array_equal(r[0], o[0])
    for x in d:
        x.free()
    d_b.free()


def test_commit_many_argument_errors(gpu_workers):
    from distributed_plonk_amd._ffi import PlonkError
    w = gpu_workers("bn254")
    d_b = _setup(w, 256)
    d = w.alloc(256 * 32)
    w.synth_fr(1, d.ptr, 256)
    assert w.commit_many_dev([]).shape == (0, 3 * w.q64)
    with pytest.raises(PlonkError):
        w.commit_many_dev([(d.ptr, 10)], start=257)
    with pytest.raises(PlonkError):
        w.commit_many_dev([(0, 10)])
    zero = w.commit_many_dev([(0, 0), (d.ptr, 0)])
    for i in range(2):
        assert w.g1_to_affine(zero[i])[1]
    d.free()
    d_b.free()


@pytest.mark.parametrize("curve,cid,log_n", [("bn254", 0, 24), ("bls12_381", 1, 22)])
def test_commit_many_full_size_exact(gpu_workers, oracle, curve, cid, log_n):
    """BASELINE's sizes, three scalar vectors of ragged length in one launch set (45 / 48 windows: the n * W * K < 2^32 bound of the sort's
    entry indices is within a factor of six), pairwise-distinct bases: every point against the EXACT expected point
    (oracle/checks.py: two 4096-point oracle MSMs of aggregated scalars; the tail beyond a shorter vector's length counts as zero)."""
    from oracle import checks
    w = gpu_workers(curve)
    n, q = 1 << log_n, w.q64
    d_b = _setup(w, n, seed=0x5EED)
    lens = [n, n - 12345, n // 2 + 1]
    polys = []
    for i, ln in enumerate(lens):
        d = w.alloc(n * 32)
        w.synth_fr(0xD15 + i, d.ptr, n)
        polys.append(d)
    got = w.commit_many_dev([(d.ptr, ln) for d, ln in zip(polys, lens)])
    for i, (d, ln) in enumerate(zip(polys, lens)):
        sc = oracle.from_mont(cid, d.download((n, 4)))
        sc[ln:] = 0
        e, ei = oracle.jac_to_affine(cid, checks.msm_expected_distinct(cid, 0x5EED, sc))
        g, gi = w.g1_to_affine(got[i])
        assert gi == ei and np.array_equal(g, e), f"polynomial {i}"
    for d in polys:
        d.free()
    d_b.free()
