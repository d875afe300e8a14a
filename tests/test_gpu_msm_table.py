"""MSM through the fixed-base window table (msm_table_kernel: T[w][i] = 2^(c*w) * P_i built at `init`, every digit of every
window accumulated into ONE bucket set) vs the oracle, forced on at sizes the oracle finishes quickly, both curves:
random scalars, sub-ranges (varMsm workloads), infinity bases, repeated bases (P + P, P - P in buckets and in the
reduction), skewed scalars (heavy buckets), and the table switched off again."""
import numpy as np
import pytest

from distributed_plonk_amd._ffi import MsmWorkload

pytestmark = pytest.mark.gpu


def _affine_eq(w, oracle, cid, got_jac, want_jac):
    g, gi = w.g1_to_affine(got_jac)
    o, oi = oracle.jac_to_affine(cid, want_jac)
    return gi == oi and np.array_equal(g, o)


@pytest.fixture
def forced_table(gpu_workers):
    used = []

    def get(curve):
        w = gpu_workers(curve)
        w.set_option("msm_precompute", 2)
        used.append(w)
        return w

    yield get
    for w in used:
        w.set_option("msm_precompute", 0)
        w.set_option("msm_table_c", 0)
        w.set_option("msm_table_sets", 0)


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("n", [2, 33, 1000, 1 << 13])
def test_table_msm_matches_oracle(forced_table, oracle, curve, cid, n):
    w = forced_table(curve)
    bases = oracle.gen_bases(cid, 19, min(n, 50), n)             # duplicates -> same-x additions
    inf = np.zeros(n, dtype=np.uint8)
    if n > 3:
        bases[3] = 0
        inf[3] = 1
    sc = oracle.from_mont(cid, oracle.rand_fr(cid, 23, n))
    sc[0] = 0
    sc[1] = oracle.field_const(cid, 0, 0) - np.array([1, 0, 0, 0], dtype=np.uint64)       # p - 1
    w.init(bases, 0, 0)
    got = w.var_msm(MsmWorkload(0, n), sc)
    assert _affine_eq(w, oracle, cid, got, oracle.msm(cid, bases, sc, inf, threads=8))
    if n >= 1000:                                                # sub-range: planes are addressed at start + w*stride
        lo, hi = n // 3, n // 3 + n // 2
        got = w.var_msm(MsmWorkload(lo, hi), sc[lo:hi])
        assert _affine_eq(w, oracle, cid, got, oracle.msm(cid, bases[lo:hi], sc[lo:hi], inf[lo:hi], threads=8))
        # commit_polynomial path (into_repr + MSM over the whole SRS prefix)
        coeffs = oracle.rand_fr(cid, 29, n - 5)
        got = w.commit(coeffs)
        assert _affine_eq(w, oracle, cid, got, oracle.commit_polynomial(cid, bases, coeffs, inf=inf, threads=8))


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_table_single_repeated_base_and_skew(forced_table, oracle, curve, cid):
    w = forced_table(curve)
    n = 4096
    bases = oracle.gen_bases(cid, 4, 1, n)                       # all bases equal
    p = int.from_bytes(oracle.field_const(cid, 0, 0).tobytes(), "little")
    w.init(bases, 0, 0)
    rnd = oracle.from_mont(cid, oracle.rand_fr(cid, 6, n))
    tiny = np.zeros((n, 4), dtype=np.uint64); tiny[:, 0] = np.arange(n) % 3
    same = np.repeat(rnd[:1], n, axis=0)
    for name, sc in (("random", rnd), ("tiny", tiny), ("all-equal", same)):
        tot = sum(int.from_bytes(s.tobytes(), "little") for s in sc) % p
        k = np.array([(tot >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
        got, gi = w.g1_to_affine(w.var_msm(MsmWorkload(0, n), sc))
        exp, ei = oracle.jac_to_affine(cid, oracle.scalar_mul(cid, bases[0], k))
        assert gi == ei and np.array_equal(got, exp), name


def test_table_on_off_agree_and_forced_window_bypasses_it(forced_table, oracle):
    w = forced_table("bn254")
    n = 1 << 15
    bases = oracle.gen_bases(0, 31, 300, n)
    sc = oracle.from_mont(0, oracle.rand_fr(0, 32, n))
    want = oracle.msm(0, bases, sc, threads=8)
    w.init(bases, 0, 0)
    a = w.var_msm(MsmWorkload(0, n), sc)
    try:
        w.set_option("msm_window", 11)                           # an explicit window uses plane 0 only
        b = w.var_msm(MsmWorkload(0, n), sc)
    finally:
        w.set_option("msm_window", 0)
    w.set_option("msm_precompute", 0)
    w.init(bases, 0, 0)
    c = w.var_msm(MsmWorkload(0, n), sc)
    w.set_option("msm_precompute", 1)                        # cost-model mode: builds the table only when it predicts a gain
    w.init(bases, 0, 0)
    assert _affine_eq(w, oracle, 0, w.var_msm(MsmWorkload(0, n), sc), want)
    for got in (a, b, c):
        assert _affine_eq(w, oracle, 0, got, want)


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("c,sets", [(8, 1), (8, 3), (8, 5), (8, 16), (11, 2), (11, 7), (13, 4)])
def test_table_with_several_bucket_sets(forced_table, oracle, curve, cid, c, sets):
    """Partial tables: T = ceil(W / G) planes 2^(c*G*t) * P_i, G bucket sets per scalar (window g + t*G -> set g, plane t) — the shape the
    plan picks when the full table does not fit its budget.  Every (c, G) pinned through the options, single MSMs, sub-ranges and a
    batched round (K vectors -> K*G sets), duplicated bases and an infinity base, against the oracle."""
    w = forced_table(curve)
    w.set_option("msm_table_c", c)
    w.set_option("msm_table_sets", sets)
    n = 3000
    bases = oracle.gen_bases(cid, 77, 120, n)
    inf = np.zeros(n, dtype=np.uint8)
    bases[11] = 0
    inf[11] = 1
    w.init(bases, 0, 0)
    sc = oracle.from_mont(cid, oracle.rand_fr(cid, 78, n))
    sc[5] = 0
    assert _affine_eq(w, oracle, cid, w.var_msm(MsmWorkload(0, n), sc), oracle.msm(cid, bases, sc, inf, threads=8))
    lo, hi = 700, 2900
    assert _affine_eq(w, oracle, cid, w.var_msm(MsmWorkload(lo, hi), sc[lo:hi]), oracle.msm(cid, bases[lo:hi], sc[lo:hi], inf[lo:hi], threads=8))
    vecs = [oracle.rand_fr(cid, 80 + k, ln) for k, ln in enumerate((n, n - 700, 0, 17))]
    bufs = [w.alloc(max(len(v), 1) * 32) for v in vecs]
    for d, v in zip(bufs, vecs):
        if len(v):
            d.upload(v)
    pts = w.commit_many_dev([(d.ptr, len(v)) for d, v in zip(bufs, vecs)])
    for v, p in zip(vecs, pts):
        if len(v):
            assert _affine_eq(w, oracle, cid, p, oracle.commit_polynomial(cid, bases, v, inf=inf, threads=8))
        else:
            assert w.g1_to_affine(p)[1]                          # the empty polynomial commits to zero
    for d in bufs:
        d.free()
