"""MSM (varMsm / commit_polynomial / round1) vs the oracle, compared in affine form
(a Jacobian triple is not unique — SURVEY fact 7)."""
import numpy as np
import pytest

from distributed_plonk_amd._ffi import MsmWorkload

pytestmark = pytest.mark.gpu


def _affine_eq(w, oracle, cid, got_jac, want_jac):
    g, gi = w.g1_to_affine(got_jac)
    o, oi = oracle.jac_to_affine(cid, want_jac)
    return gi == oi and np.array_equal(g, o)


def _bases_with_inf(oracle, cid, bases, inf_idx):
    b = bases.copy()
    inf = np.zeros(len(b), dtype=np.uint8)
    for i in inf_idx:
        b[i] = 0
        inf[i] = 1
    return b, inf


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("n", [1, 5, 40, 1000, 1 << 14])
def test_msm_matches_oracle(gpu_workers, oracle, curve, cid, n):
    w = gpu_workers(curve)
    uniq = min(n, 64)                                   # duplicated bases => P+P in buckets (dispatcher.rs:194-196)
    bases = oracle.gen_bases(cid, 9, uniq, n)
    b, inf = _bases_with_inf(oracle, cid, bases, [3] if n > 3 else [])     # infinity base (dispatcher2.rs:1101)
    sc = oracle.from_mont(cid, oracle.rand_fr(cid, 21, n))
    sc[0] = 0
    if n > 2:
        sc[1] = [1, 0, 0, 0]
        sc[2] = oracle.field_const(cid, 0, 0) - np.array([1, 0, 0, 0], dtype=np.uint64)    # p - 1
    w.init(b, 0, 0)
    got = w.var_msm(MsmWorkload(0, n), sc)
    want = oracle.msm(cid, bases, sc, inf, threads=4)
    assert _affine_eq(w, oracle, cid, got, want)
    if n <= 40:
        assert _affine_eq(w, oracle, cid, got, oracle.msm_naive(cid, bases, sc, inf))


def test_msm_sharded_like_test_msm(gpu_workers, oracle):
    """dispatcher.rs:177-244: S contiguous shards, reduce(a+b) == monolithic MSM."""
    w = gpu_workers("bn254")
    n, S = 1 << 16, 4
    bases = oracle.gen_bases(0, 3, 1 << 11, n)
    sc = oracle.from_mont(0, oracle.rand_fr(0, 4, n))
    w.init(bases, 0, 0)
    acc = None
    for i in range(S):
        lo, hi = i * n // S, (i + 1) * n // S
        part = w.var_msm(MsmWorkload(lo, hi), sc[lo:hi])
        acc = part if acc is None else w.g1_add(acc, part)
    assert _affine_eq(w, oracle, 0, acc, oracle.msm(0, bases, sc, threads=8))
    assert _affine_eq(w, oracle, 0, acc, w.var_msm(MsmWorkload(0, n), sc))


def test_msm_cancellation_and_zero(gpu_workers, oracle):
    w = gpu_workers("bn254")
    bases = oracle.gen_bases(0, 1, 1, 2)               # P, P
    p = oracle.field_const(0, 0, 0)
    s = np.array([[5, 0, 0, 0]], dtype=np.uint64)
    neg = (p - np.array([5, 0, 0, 0], dtype=np.uint64)).reshape(1, 4)   # -5 mod r (no borrow: low limb > 5)
    w.init(bases, 0, 0)
    out = w.var_msm(MsmWorkload(0, 2), np.vstack([s, neg]))
    assert w.g1_to_affine(out)[1]                      # 5P + (-5)P = infinity
    assert w.g1_to_affine(w.var_msm(MsmWorkload(0, 0), np.zeros((0, 4), dtype=np.uint64)))[1]
    from distributed_plonk_amd._ffi import PlonkError
    with pytest.raises(PlonkError):
        w.var_msm(MsmWorkload(1, 5), s)                # out of range: the reference would panic (worker.rs:180)


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_commit_and_round1(gpu_workers, oracle, curve, cid):
    w = gpu_workers(curve)
    n = 1 << 10
    bases = oracle.gen_bases(cid, 2, 128, n + 32)
    w.init(bases, n, 8 * n)
    coeffs = oracle.rand_fr(cid, 8, n)
    assert _affine_eq(w, oracle, cid, w.commit(coeffs), oracle.commit_polynomial(cid, bases, coeffs, threads=4))
    evals = oracle.rand_fr(cid, 9, n)
    bl = oracle.rand_fr(cid, 10, 2)
    poly, cm = oracle.round1(cid, bases, evals, bl, threads=4)
    got = w.round1(evals, bl)
    assert np.array_equal(w.get_wire(n + 2), poly)
    assert _affine_eq(w, oracle, cid, got, cm)


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n", [5, 10])
def test_circuit_shaped_commitments(gpu_workers, oracle, curve, cid, log_n):
    """SURVEY §8d's circuit-shaped inputs: the prover's commit key is n + 3 powers padded with G1Affine::zero() — arkworks' (0, 1, true) —
    to a multiple of 32 (dispatcher.rs:541-542, dispatcher2.rs:207-208), shipped as raw `[G1Affine]` memory (utils.rs:27-33), and
    commit_polynomial zero-pads n, n + 2 or n + 3 coefficients to that length (dispatcher.rs:1048-1050).  Whole-key commitments of each
    length, and the sharded form of dispatcher2.rs:870-890 (S contiguous ranges of the PADDED key; the last ranges see only zero scalars
    and infinite bases), against the oracle."""
    from distributed_plonk_amd import _ffi
    w = gpu_workers(curve)
    q = w.q64
    n = 1 << log_n
    real = n + 3
    L = ((real + 31) >> 5) << 5
    bases = oracle.gen_bases(cid, 41 + log_n, real, L)         # L distinct-seeded points; the tail is overwritten by infinity below
    inf = np.zeros(L, dtype=np.uint8)
    inf[real:] = 1
    stride = 16 * q + 8
    raw = np.zeros((L, stride), dtype=np.uint8)
    raw[:, :16 * q] = bases.view(np.uint8).reshape(L, 16 * q)
    one = oracle.field_const(cid, 1, 1).view(np.uint8)
    raw[real:, :8 * q] = 0
    raw[real:, 8 * q:16 * q] = one
    raw[real:, 16 * q] = 1
    w.init(raw, n, 8 * n, layout=_ffi.PLONK_BASES_ARK)
    for k, ln in enumerate((n, n + 2, n + 3)):                  # selector / wire / permutation polynomial lengths
        coeffs = oracle.rand_fr(cid, 50 + k, ln)
        want = oracle.commit_polynomial(cid, bases, coeffs, inf, threads=4)
        assert _affine_eq(w, oracle, cid, w.commit(coeffs), want), ln
        # the reference's sharded form: scalars = into_repr(coeffs) zero-padded to the key length, S ranges [i*L/S, (i+1)*L/S)
        sc = np.zeros((L, 4), dtype=np.uint64)
        sc[:ln] = oracle.from_mont(cid, coeffs)
        for S in (1, 2, 8):
            acc = None
            for i in range(S):
                lo, hi = i * L // S, (i + 1) * L // S
                part = w.var_msm(MsmWorkload(lo, hi), sc[lo:hi])
                acc = part if acc is None else w.g1_add(acc, part)
            assert _affine_eq(w, oracle, cid, acc, want), (ln, S)


def test_synth_inputs_match_oracle(gpu_workers, oracle):
    for curve, cid in [("bn254", 0), ("bls12_381", 1)]:
        w = gpu_workers(curve)
        n = 1000
        buf = w.alloc(n * 32)
        w.synth_fr(123, buf.ptr, n)
        assert np.array_equal(buf.download((n, 4)), oracle.rand_fr(cid, 123, n))
        buf.free()
        q = w.q64
        pb = w.alloc(300 * 16 * q)
        w.synth_bases(77, 100, 300, pb.ptr)
        assert np.array_equal(pb.download((300, 2 * q)), oracle.gen_bases(cid, 77, 100, 300))
        w.synth_bases(5, 0, 300, pb.ptr)               # pairwise-distinct variant
        pts = pb.download((300, 2 * q))
        assert all(oracle.on_curve(cid, p) for p in pts[:50])
        assert len({p.tobytes() for p in pts}) == 300
        pb.free()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_msm_single_repeated_base_hits_every_exceptional_path(gpu_workers, oracle, curve, cid):
    """All bases equal: every bucket is a multiple of one point, so bucket accumulation meets P + P on its
    second addition and the reduction pyramid keeps adding equal / opposite points.  sum_i s_i * P = (sum_i s_i) * P."""
    w = gpu_workers(curve)
    for n, small in [(4096, False), (3000, True)]:
        bases = oracle.gen_bases(cid, 4, 1, n)
        sc = oracle.from_mont(cid, oracle.rand_fr(cid, 6, n))
        if small:                                  # tiny scalars: only window 0 is populated, buckets 1..16 heavily
            sc[:] = 0
            sc[:, 0] = np.arange(n) % 17
        p = int.from_bytes(oracle.field_const(cid, 0, 0).tobytes(), "little")
        tot = sum(int.from_bytes(s.tobytes(), "little") for s in sc) % p
        k = np.array([(tot >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
        w.init(bases, 0, 0)
        got, gi = w.g1_to_affine(w.var_msm(MsmWorkload(0, n), sc))
        exp, ei = oracle.jac_to_affine(cid, oracle.scalar_mul(cid, bases[0], k))
        assert gi == ei and np.array_equal(got, exp)


def test_msm_skewed_scalars_and_forced_windows(gpu_workers, oracle):
    """Distributions that put (almost) every point of a window into one bucket — real witnesses are full of 0/1
    values, and a window width that leaves two bits for the top window does it for uniform scalars too.
    Buckets above 2048 entries are shared by several workgroups (msm_heavy_kernel)."""
    w = gpu_workers("bn254")
    n = 1 << 16
    bases = oracle.gen_bases(0, 12, 257, n)
    bases[100] = 0
    inf = np.zeros(n, dtype=np.uint8); inf[100] = 1
    w.init(bases, 0, 0)
    rnd = oracle.from_mont(0, oracle.rand_fr(0, 13, n))
    cases = {}
    same = np.repeat(rnd[:1], n, axis=0)                       # one scalar everywhere: one bucket per window
    cases["all-equal"] = same
    small = np.zeros((n, 4), dtype=np.uint64); small[:, 0] = np.arange(n) % 3      # 0 / 1 / 2
    cases["tiny"] = small
    ones = np.zeros((n, 4), dtype=np.uint64); ones[:, 0] = 1
    cases["all-ones"] = ones
    try:
        for name, sc in cases.items():
            got = w.var_msm(MsmWorkload(0, n), sc)
            assert _affine_eq(w, oracle, 0, got, oracle.msm(0, bases, sc, inf, threads=8)), name
        for c in (18, 12, 5):                                  # 18: two bits left for the top window of a 254-bit scalar
            w.set_option("msm_window", c)
            got = w.var_msm(MsmWorkload(0, n), rnd)
            assert _affine_eq(w, oracle, 0, got, oracle.msm(0, bases, rnd, inf, threads=8)), c
    finally:
        w.set_option("msm_window", 0)


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_persistent_accumulation_matches_plain_grid_and_oracle(gpu_workers, oracle, curve, cid):
    """`msm_acc_persist`: the bucket accumulation as a fixed grid of persistent waves that take 64 buckets at a time from a global counter
    (the default at full size, where the plain grid would be 26 rounds of workgroups).  Forced here onto tiny grids (1, 3, 40 workgroups:
    every wave loops many times, the last group is ragged) with duplicated bases (redo path), an infinity base, a heavy bucket
    (one scalar everywhere) and a forced window, against the plain grid and the oracle."""
    w = gpu_workers(curve)
    n = 1 << 13
    bases = oracle.gen_bases(cid, 31, 200, n)
    b, inf = _bases_with_inf(oracle, cid, bases, [7])
    w.init(b, 0, 0)
    rnd = oracle.from_mont(cid, oracle.rand_fr(cid, 32, n))
    same = np.repeat(rnd[:1], n, axis=0)
    try:
        for name, sc, window in (("uniform", rnd, 0), ("uniform-c11", rnd, 11), ("all-equal", same, 0)):
            w.set_option("msm_window", window)
            want = oracle.msm(cid, bases, sc, inf, threads=8)
            w.set_option("msm_acc_persist", 0)
            plain = w.var_msm(MsmWorkload(0, n), sc)
            assert _affine_eq(w, oracle, cid, plain, want), (name, "plain")
            for grid in (1, 3, 40):
                w.set_option("msm_acc_persist", -grid)
                got = w.var_msm(MsmWorkload(0, n), sc)
                assert _affine_eq(w, oracle, cid, got, want), (name, grid)
        # a batched round through the persistent path (K scalar vectors = K * W windows of one bucket problem)
        w.set_option("msm_window", 0)
        w.set_option("msm_acc_persist", -5)
        vecs = [oracle.rand_fr(cid, 40 + k, n) for k in range(3)]
        bufs = [w.alloc(n * 32) for _ in vecs]
        for d, v in zip(bufs, vecs):
            d.upload(v)
        pts = w.commit_many_dev([(d.ptr, n) for d in bufs])
        for v, p in zip(vecs, pts):
            assert _affine_eq(w, oracle, cid, p, oracle.msm(cid, bases, oracle.from_mont(cid, v), inf, threads=8))
        for d in bufs:
            d.free()
    finally:
        w.set_option("msm_window", 0)
        w.set_option("msm_acc_persist", 4)


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_bucket_ordering_fused_into_the_sort_matches_the_three_launch_form(gpu_workers, oracle, curve, cid):
    """`msm_fused_order` (round 5): the bucket-size histogram taken inside the level-2 sort — both partition kernels: the direct one (narrow
    windows) and the LDS-staged one (2^8 and more buckets per partition), its chunked path included — and scanned inside the placement.  The
    default switches it on from 2^23 points per launch only, so it is FORCED here (2) against the three-launch form (0) and the oracle, with
    duplicated bases, an infinity base, a heavy bucket and a batched round."""
    w = gpu_workers(curve)
    n = 1 << 13
    bases = oracle.gen_bases(cid, 33, 300, n)
    b, inf = _bases_with_inf(oracle, cid, bases, [11])
    w.init(b, 0, 0)
    rnd = oracle.from_mont(cid, oracle.rand_fr(cid, 34, n))
    same = np.repeat(rnd[:1], n, axis=0)
    try:
        for name, sc, window, cap in (("uniform", rnd, 0, 0), ("c5-direct-sort", rnd, 5, 0), ("c14-staged-sort", rnd, 14, 0), ("c14-chunked", rnd, 14, 1024),
                                      ("all-equal", same, 0, 0)):
            w.set_option("msm_window", window)
            w.set_option("msm_sort_stage_cap", cap)
            want = oracle.msm(cid, bases, sc, inf, threads=8)
            for mode in (0, 2):
                w.set_option("msm_fused_order", mode)
                assert _affine_eq(w, oracle, cid, w.var_msm(MsmWorkload(0, n), sc), want), (name, mode)
        w.set_option("msm_window", 0)
        w.set_option("msm_sort_stage_cap", 0)
        w.set_option("msm_fused_order", 2)
        vecs = [oracle.rand_fr(cid, 50 + k, n) for k in range(3)]
        bufs = [w.alloc(n * 32) for _ in vecs]
        for d, v in zip(bufs, vecs):
            d.upload(v)
        for v, p in zip(vecs, w.commit_many_dev([(d.ptr, n) for d in bufs])):
            assert _affine_eq(w, oracle, cid, p, oracle.msm(cid, bases, oracle.from_mont(cid, v), inf, threads=8))
        for d in bufs:
            d.free()
    finally:
        w.set_option("msm_window", 0)
        w.set_option("msm_sort_stage_cap", 0)
        w.set_option("msm_fused_order", 1)


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_grid_reduction_matches_pyramid_and_oracle(gpu_workers, oracle, curve, cid):
    """`msm_reduce_grid` (experiment, off by default): the window reduction V = sum_j (j+1) B_j as row / column tree sums of the H x L bucket
    grid plus bit sums of the <= 1024 row / column totals (msm_engine.hip, 5b) instead of the running-sum pyramid.  Every window width from
    3 bits (a 2 x 2 grid) to 13 (64 x 64) and the plan's own choice — odd and even bucket-index widths, grids narrower than a workgroup's
    column / row coverage — with duplicated bases (P + P inside the trees), an infinity base, all scalars equal (one non-empty bucket per window:
    every other tree input is the point at infinity), scalars 0 / 1 / p - 1, and a batched round; against the pyramid and the oracle."""
    w = gpu_workers(curve)
    n = 1 << 12
    bases = oracle.gen_bases(cid, 77, 300, n)
    b, inf = _bases_with_inf(oracle, cid, bases, [5, 6])
    w.init(b, 0, 0)
    rnd = oracle.from_mont(cid, oracle.rand_fr(cid, 78, n))
    rnd[0] = 0
    rnd[1] = [1, 0, 0, 0]
    rnd[2] = oracle.field_const(cid, 0, 0) - np.array([1, 0, 0, 0], dtype=np.uint64)
    same = np.repeat(rnd[9:10], n, axis=0)
    try:
        for name, sc, windows in (("uniform", rnd, (0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13)), ("all-equal", same, (0, 4, 7, 12))):
            want = oracle.msm(cid, bases, sc, inf, threads=8)
            for window in windows:
                w.set_option("msm_window", window)
                w.set_option("msm_reduce_grid", 0)
                pyramid = w.var_msm(MsmWorkload(0, n), sc)
                w.set_option("msm_reduce_grid", 1)
                grid = w.var_msm(MsmWorkload(0, n), sc)
                assert _affine_eq(w, oracle, cid, pyramid, want), (name, window, "pyramid")
                assert _affine_eq(w, oracle, cid, grid, want), (name, window, "grid")
        # a sub-range of the resident bases and a batched round (K vectors = K * W bucket sets reduced by one pair of launches), ragged lengths
        w.set_option("msm_window", 0)
        lo, hi = 100, 3001
        assert _affine_eq(w, oracle, cid, w.var_msm(MsmWorkload(lo, hi), rnd[lo:hi]), oracle.msm(cid, bases[lo:hi], rnd[lo:hi], inf[lo:hi], threads=8))
        lens = (n, n - 37, 1, 0)
        vecs = [oracle.rand_fr(cid, 90 + k, n) for k in range(len(lens))]
        bufs = [w.alloc(n * 32) for _ in vecs]
        for d, v in zip(bufs, vecs):
            d.upload(v)
        pts = w.commit_many_dev([(d.ptr, ln) for d, ln in zip(bufs, lens)])
        for v, ln, p_ in zip(vecs, lens, pts):
            if ln:
                assert _affine_eq(w, oracle, cid, p_, oracle.msm(cid, bases[:ln], oracle.from_mont(cid, v)[:ln], inf[:ln], threads=8)), ln
            else:
                assert w.g1_to_affine(p_)[1], "the empty polynomial commits to the point at infinity"
        for d in bufs:
            d.free()
    finally:
        w.set_option("msm_window", 0)
        w.set_option("msm_reduce_grid", 0)


def test_level2_sort_in_chunks_matches_single_chunk_and_oracle(gpu_workers, oracle):
    """The staged level-2 sort orders a partition in LDS; a partition larger than its buffer (every partition above 2^24 points; skewed
    scalars at any size) is ordered in several chunks of consecutive buckets.  Forced here with a 1024-entry buffer at 2^18 points and a
    20-bit window (256-entry partitions fit, the top window's and the skewed vector's do not), against the default buffer and the oracle."""
    w = gpu_workers("bn254")
    n = 1 << 18
    bases = oracle.gen_bases(0, 51, 1 << 10, n)
    w.init(bases, 0, 0)
    rnd = oracle.from_mont(0, oracle.rand_fr(0, 52, n))
    skew = rnd.copy()
    skew[:, 0] &= np.uint64(0xFFF)                              # window 0: 2^18 points in 4096 buckets of 4 partitions
    try:
        w.set_option("msm_window", 20)
        for name, sc in (("uniform", rnd), ("skewed", skew)):
            want = oracle.msm(0, bases, sc, threads=8)
            w.set_option("msm_sort_stage_cap", 0)
            assert _affine_eq(w, oracle, 0, w.var_msm(MsmWorkload(0, n), sc), want), (name, "default buffer")
            for cap in (1024, 3000):
                w.set_option("msm_sort_stage_cap", cap)
                assert _affine_eq(w, oracle, 0, w.var_msm(MsmWorkload(0, n), sc), want), (name, cap)
    finally:
        w.set_option("msm_window", 0)
        w.set_option("msm_sort_stage_cap", 0)
