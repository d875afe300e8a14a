"""Pin the C oracle (oracle/plonk_oracle.c) before trusting it.

The reference has no golden vectors or KATs for this path ("parity unpinned" by data),
so the oracle is pinned by (1) the constants of SURVEY Appendix B, (2) an independent pure-Python
big-int statement of the same published algorithms (oracle/bigint_ref.py), (3) O(N^2) DFTs and
double-and-add, (4) the committed tests/golden/*.json, and (5) the identities the reference's own
tests assert (playground.rs:95-102, dispatcher.rs:240,334-342)."""
import random

import numpy as np
import pytest

from oracle import bigint_ref as B
from oracle import oracle as O

import golden_util as G

CURVES = [("bn254", O.BN254, B.BN254), ("bls12_381", O.BLS12_381, B.BLS12_381)]
MODES = [(False, False), (True, False), (False, True), (True, True)]


def ints(a):
    return [B.from_limbs(r) for r in a]


def to_limbs(vals, n=4):
    return np.array([B.to_limbs(v, n) for v in vals], dtype=np.uint64)


@pytest.mark.parametrize("name,cid,cv", CURVES)
def test_constants_match_survey_appendix_b(name, cid, cv):
    f = cv.fr
    assert B.from_limbs(O.field_const(cid, 0, 0)) == f.p
    assert B.from_limbs(O.field_const(cid, 1, 0)) == cv.fq.p
    assert B.from_limbs(O.field_const(cid, 0, 1)) == pow(2, 256, f.p)
    assert B.from_limbs(O.field_const(cid, 1, 1)) == pow(2, 64 * cv.fq.limbs64, cv.fq.p)
    w = f.from_mont(B.from_limbs(O.field_const(cid, 0, 3)))
    assert pow(w, 1 << f.two_adicity, f.p) == 1 and pow(w, 1 << (f.two_adicity - 1), f.p) == f.p - 1
    if name == "bn254":
        assert f.two_adicity == 28 and f.generator == 5
        assert w == 19103219067921713944291392827692070036145651957329286315305642004821462161904
        assert O.field_inv64(cid, 0) == 0xc2e1f593efffffff
    else:
        assert f.two_adicity == 32 and f.generator == 7
        assert w == 10238227357739495823651030575849232062558860180284477541189508159991286009131
        assert O.field_inv64(cid, 0) == 0xfffffffeffffffff
    assert O.on_curve(cid, O.generator(cid))


@pytest.mark.parametrize("name,cid,cv", CURVES)
def test_field_ops_vs_bigint(name, cid, cv):
    rng = random.Random(5)
    for fld, ff in ((0, cv.fr), (1, cv.fq)):
        n64 = ff.limbs64
        a = [rng.randrange(ff.p) for _ in range(40)] + [0, 1, ff.p - 1]
        b = [rng.randrange(ff.p) for _ in range(40)] + [ff.p - 1, 0, ff.p - 1]
        A, Bm = to_limbs(a, n64), to_limbs(b, n64)
        rinv = pow(ff.R, -1, ff.p)
        assert ints(O.field_op(cid, fld, "mul", A, Bm)) == [x * y * rinv % ff.p for x, y in zip(a, b)]
        assert ints(O.field_op(cid, fld, "add", A, Bm)) == [(x + y) % ff.p for x, y in zip(a, b)]
        assert ints(O.field_op(cid, fld, "sub", A, Bm)) == [(x - y) % ff.p for x, y in zip(a, b)]
        assert ints(O.field_op(cid, fld, "from_mont", A)) == [x * rinv % ff.p for x in a]
        assert ints(O.field_op(cid, fld, "to_mont", A)) == [x * ff.R % ff.p for x in a]
        nz = [x for x in a if x]
        assert ints(O.field_op(cid, fld, "inv", to_limbs(nz, n64))) == [pow(x * rinv, -1, ff.p) * ff.R % ff.p for x in nz]


@pytest.mark.parametrize("name,cid,cv", CURVES)
def test_ntt_vs_bigint_and_naive_dft(name, cid, cv):
    f = cv.fr
    rng = random.Random(6)
    for n in (1, 2, 4, 32, 256, 2048):
        v = [rng.randrange(f.p) for _ in range(n)]
        vm = to_limbs([f.to_mont(x) for x in v])
        d = B.Radix2Domain(f, n)
        for inv, coset in MODES:
            ref = {(False, False): d.fft, (True, False): d.ifft, (False, True): d.coset_fft, (True, True): d.coset_ifft}[(inv, coset)](v)
            assert ints(O.ntt(cid, vm, inv, coset)) == [f.to_mont(x) for x in ref], (n, inv, coset)
        if 2 <= n <= 256:
            assert np.array_equal(O.naive_dft(cid, vm, False), O.ntt(cid, vm, False, False))
            assert np.array_equal(O.naive_dft(cid, vm, True), O.ntt(cid, vm, True, False))


@pytest.mark.parametrize("name,cid,cv", CURVES)
def test_reference_2d_decomposition_equals_direct(name, cid, cv):
    """playground.rs:95-99, dispatcher.rs:334-342 (2^11 and 2^13: odd log N, c = 2r), dispatcher2.rs:1200-1208."""
    for log_n in (4, 7, 11, 13):
        v = O.rand_fr(cid, 40 + log_n, 1 << log_n)
        for inv, coset in MODES:
            want = O.ntt(cid, v, inv, coset)
            assert np.array_equal(O.fourstep(cid, v, inv, coset), want)
            for S in (1, 2, 4):
                assert np.array_equal(O.distributed_fft(cid, v, S, inv, coset), want)


def test_playground_identities():
    """playground.rs:100-102."""
    l = 512
    exps = O.rand_fr(O.BLS12_381, 3, l)
    t = np.zeros((2 * l, 4), dtype=np.uint64)
    t[:l] = exps
    assert np.array_equal(O.ntt(1, t, False, True), O.ntt(1, np.vstack([exps, np.zeros_like(exps)]), False, True))
    assert np.array_equal(O.ntt(1, t, False, False), O.fourstep(1, t, False, False))
    assert np.array_equal(O.ntt(1, O.ntt(1, exps, False, True), True, True), exps)


def test_domain_creation_error():
    """BN254's quotient domain for n = 2^28 does not exist (dispatcher2.rs:246-247)."""
    buf = np.zeros((2, 4), dtype=np.uint64)
    assert O.lib().orc_ntt(O.BN254, buf.ctypes.data, 29, 0, 0, 1) == -1          # two-adicity 28
    assert O.lib().orc_ntt(O.BLS12_381, buf.ctypes.data, 33, 0, 0, 1) == -1      # two-adicity 32
    with pytest.raises(ValueError):
        B.Radix2Domain(B.BN254_FR, 1 << 29)


@pytest.mark.parametrize("name,cid,cv", CURVES)
def test_msm_vs_bigint(name, cid, cv):
    rng = random.Random(8)
    q = cv.fq.limbs64
    bases = O.gen_bases(cid, 5, 8, 40)                              # 8 unique points tiled (dispatcher.rs:190-196)
    pts = [(cv.fq.from_mont(B.from_limbs(b[:q])), cv.fq.from_mont(B.from_limbs(b[q:]))) for b in bases]
    assert all(B.on_curve(cv, P) for P in pts)
    inf = np.zeros(40, dtype=np.uint8)
    inf[3] = 1
    pts[3] = None                                                   # infinity base (dispatcher2.rs:1101)
    sc = [rng.randrange(cv.fr.p) for _ in pts]
    sc[0], sc[1], sc[2] = 0, 1, cv.fr.p - 1
    scl = to_limbs(sc)
    ref = B.msm_naive(cv, pts, sc)
    for res in (O.msm(cid, bases, scl, inf, threads=3), O.msm_naive(cid, bases, scl, inf), O.sharded_msm(cid, bases, scl, 4, inf)):
        xy, isinf = O.jac_to_affine(cid, res)
        assert not isinf
        assert (cv.fq.from_mont(B.from_limbs(xy[:q])), cv.fq.from_mont(B.from_limbs(xy[q:]))) == ref
    # commit_polynomial = into_repr + zero-pad + MSM (worker.rs:117-123)
    coeffs = O.rand_fr(cid, 77, 30)
    cm = O.commit_polynomial(cid, bases, coeffs, inf)
    sc2 = [cv.fr.from_mont(x) for x in ints(coeffs)]
    xy, isinf = O.jac_to_affine(cid, cm)
    want = B.msm_naive(cv, pts[:30], sc2)
    assert (cv.fq.from_mont(B.from_limbs(xy[:q])), cv.fq.from_mont(B.from_limbs(xy[q:]))) == want


@pytest.mark.parametrize("name,cid,cv", CURVES)
def test_oracle_matches_golden_fixtures(name, cid, cv):
    g = G.load(name)
    assert int(g["fr_modulus"], 16) == cv.fr.p and int(g["fr_R"], 16) == cv.fr.R
    for e in g["field_mul"]:
        fld = 0 if e["field"] == "fr" else 1
        n64 = 4 if fld == 0 else cv.fq.limbs64
        got = O.field_op(cid, fld, "mul", G.limbs([e["a"]], n64), G.limbs([e["b"]], n64))
        assert B.from_limbs(got[0]) == int(e["mont_mul"], 16)
    for e in g["ntt"]:
        v = G.limbs(e["input_mont"])
        for key, (inv, coset) in {"fft": (False, False), "ifft": (True, False), "coset_fft": (False, True), "coset_ifft": (True, True)}.items():
            assert np.array_equal(O.ntt(cid, v, inv, coset), G.limbs(e[key])), (e["log_n"], key)
    q = cv.fq.limbs64
    for e in g["msm"]:
        bases, inf = G.bases_from_golden(e, q)
        sc = G.limbs(e["scalars"])
        xy, isinf = O.jac_to_affine(cid, O.msm(cid, bases, sc, inf, threads=2))
        want = e["result_affine_mont"]
        assert not isinf and B.from_limbs(xy[:q]) == int(want[0], 16) and B.from_limbs(xy[q:]) == int(want[1], 16)


def test_rand_fr_is_uniform_montgomery_draw():
    for cid, cv in ((O.BN254, B.BN254), (O.BLS12_381, B.BLS12_381)):
        r = ints(O.rand_fr(cid, 9, 2000))
        assert all(x < cv.fr.p for x in r) and len(set(r)) == 2000
        assert np.array_equal(O.rand_fr(cid, 9, 10), O.rand_fr(cid, 9, 2000)[:10])       # per-index streams


@pytest.mark.parametrize("name,cid,cv", CURVES)
def test_quotient_evals_vs_bigint(name, cid, cv):
    """Next row: dispatcher2.rs:362-504 restated in C vs the pure-Python statement."""
    f = cv.fr
    rng = random.Random(21)
    for log_n in (1, 3):
        n, m = 1 << log_n, 8 << log_n
        rv = lambda k: [[rng.randrange(f.p) for _ in range(m)] for _ in range(k)]
        sel, sig, wire, z, pi = rv(13), rv(5), rv(5), rv(1)[0], rv(1)[0]
        al, be, ga = (rng.randrange(f.p) for _ in range(3))
        k = [rng.randrange(f.p) for _ in range(5)]
        ref = B.quotient_evals(f, n, sel, sig, wire, z, pi, al, be, ga, k)
        M = lambda v: to_limbs([f.to_mont(x) for x in v])
        got = O.quotient_evals(cid, log_n, np.stack([M(v) for v in sel]), np.stack([M(v) for v in sig]), np.stack([M(v) for v in wire]),
                               M(z), M(pi), M([al])[0], M([be])[0], M([ga])[0], M(k), threads=2)
        assert ints(got) == [f.to_mont(x) for x in ref]


def test_bn254_group_law_published_points():
    """Published known answers for the BN254 (alt_bn128) group law — 2G and 3G for G = (1, 2), the values the EIP-196 precompile
    tests use — for both oracle layers.  (The reference itself holds no vectors; these pin affine doubling / addition and the
    Jacobian scalar multiplication to something outside this repository.)"""
    cv = B.BN254
    g2 = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
          9918110051302171585080402603319702774565515993150576347155970296011118125764)
    g3 = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
          19321533766552368860946552437480515441416830039777911637913418824951667761761)
    G = (cv.gx, cv.gy)
    assert B.affine_add(cv, G, G) == g2
    assert B.affine_add(cv, g2, G) == g3
    assert B.scalar_mul(cv, 3, G) == g3
    fq = cv.fq
    for k, want in ((2, g2), (3, g3)):
        jac = O.scalar_mul(0, O.generator(0), np.array([k, 0, 0, 0], dtype=np.uint64))
        xy, inf = O.jac_to_affine(0, jac)
        assert not inf
        assert (fq.from_mont(B.from_limbs(xy[:4])), fq.from_mont(B.from_limbs(xy[4:]))) == want
    # the MSM agrees: 1*G + 2*G = 3G
    bases = np.stack([O.generator(0), O.generator(0)])
    sc = np.array([[1, 0, 0, 0], [2, 0, 0, 0]], dtype=np.uint64)
    xy, inf = O.jac_to_affine(0, O.msm(0, bases, sc))
    assert (fq.from_mont(B.from_limbs(xy[:4])), fq.from_mont(B.from_limbs(xy[4:]))) == g3


def test_bls12_381_group_law_published_points():
    """Published known answers for BLS12-381 G1: the 48-byte compressed encodings (ZCash / IETF BLS-signature format: big-endian x,
    bit 7 = compressed, bit 5 = y is the larger root) of 1G, 2G and 3G — the public keys of the secret keys 1, 2, 3 that every
    BLS12-381 signature test suite (eth2, IETF draft) carries.  Pins affine doubling / addition, the Jacobian scalar multiplication and
    the MSM of both oracle layers for the 381-bit field to values from outside this repository, as EIP-196's 2G / 3G do for BN254."""
    cv = B.BLS12_381
    fq = cv.fq
    published = {
        1: "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb",
        2: "a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e",
        3: "89ece308f9d1f0131765212deca99697b112d61f9be9a5f1f3780a51335b3ff981747a0b2ca2179b96d2c0c9024e5224",
    }

    def zcash_compressed(pt):
        x, y = pt
        b = bytearray(x.to_bytes(48, "big"))
        b[0] |= 0x80 | (0x20 if y > fq.p - y else 0)
        return b.hex()

    G = (cv.gx, cv.gy)
    g2 = B.affine_add(cv, G, G)
    g3 = B.affine_add(cv, g2, G)
    assert zcash_compressed(G) == published[1] and zcash_compressed(g2) == published[2] and zcash_compressed(g3) == published[3]
    assert B.scalar_mul(cv, 3, G) == g3
    for k in (1, 2, 3):
        jac = O.scalar_mul(1, O.generator(1), np.array([k, 0, 0, 0], dtype=np.uint64))
        xy, inf = O.jac_to_affine(1, jac)
        assert not inf
        assert zcash_compressed((fq.from_mont(B.from_limbs(xy[:6])), fq.from_mont(B.from_limbs(xy[6:])))) == published[k]
    bases = np.stack([O.generator(1), O.generator(1)])
    sc = np.array([[1, 0, 0, 0], [2, 0, 0, 0]], dtype=np.uint64)
    xy, inf = O.jac_to_affine(1, O.msm(1, bases, sc))
    assert zcash_compressed((fq.from_mont(B.from_limbs(xy[:6])), fq.from_mont(B.from_limbs(xy[6:])))) == published[3]


@pytest.mark.parametrize("name,cid,cv", CURVES)
def test_expected_msm_helpers_match_a_direct_oracle_msm(name, cid, cv):
    """oracle/checks.py (bench.py's verification leg and the full-size GPU tests rely on it): the aggregated-scalar shortcuts for the
    two synthetic base distributions equal a direct oracle MSM over the explicitly constructed bases."""
    from oracle import checks
    q = cv.fq.limbs64
    one = O.field_const(cid, 1, 1)[:q]
    # tiled: P_i = T[i % u]
    n, u = 1536, 256
    sc = O.from_mont(cid, O.rand_fr(cid, 31, n))
    bases = O.gen_bases(cid, 77, u, n)
    a = O.jac_to_affine(cid, O.msm(cid, bases, sc, threads=2))
    b = O.jac_to_affine(cid, checks.msm_expected_tiled(cid, 77, u, sc, threads=2))
    assert a[1] == b[1] and np.array_equal(a[0], b[0])
    # distinct: P_i = A[i % 4096] + B[i / 4096]
    n = 4096 + 900
    sc = O.from_mont(cid, O.rand_fr(cid, 32, n))
    A, Bp = O.gen_bases(cid, 5, checks.NA, checks.NA), O.gen_bases(cid, 6, 2, 2)
    bases = np.zeros((n, 2 * q), dtype=np.uint64)
    for i in range(n):
        s = O.jac_add(cid, np.concatenate([A[i % checks.NA], one]), np.concatenate([Bp[i // checks.NA], one]))
        bases[i] = O.jac_to_affine(cid, s)[0]
    a = O.jac_to_affine(cid, O.msm(cid, bases, sc, threads=2))
    b = O.jac_to_affine(cid, checks.msm_expected_distinct(cid, 5, sc, threads=2))
    assert a[1] == b[1] and np.array_equal(a[0], b[0])
