"""Pin the oracle to THIRD-PARTY code.

The reference asserts its own distributed path differentially against arkworks (`dispatcher.rs:240`: sharded MSM == monolithic MSM;
`dispatcher.rs:334-342`, `playground.rs:95-99`: 4-step == `domain.{fft, ifft, coset_fft, coset_ifft}`).  arkworks cannot run here, but
the image holds independent implementations of the same mathematics that this repository neither wrote nor ships:
`sympy.discrete.transforms.ntt / intt` (iterative radix-2 over w = primitive_root(p)^((p-1)/n); primitive_root returns the SMALLEST
generator = arkworks' `GENERATOR` 5 / 7, hence the same w) and `sympy.ntheory.elliptic_curve.EllipticCurve` (affine chord-and-tangent).
Here both restatements — the C oracle and oracle/bigint_ref.py — are run against sympy LIVE and against the committed
sympy-generated fixtures (tests/golden/sympy_*.json, tools/gen_golden_sympy.py).  If either drifts from sympy, this file fails."""
import random

import numpy as np
import pytest

sympy = pytest.importorskip("sympy")
from sympy.discrete.transforms import intt, ntt            # noqa: E402
from sympy.ntheory import primitive_root                   # noqa: E402
from sympy.ntheory.elliptic_curve import EllipticCurve     # noqa: E402

from oracle import bigint_ref as B                          # noqa: E402
from oracle import oracle as O                              # noqa: E402

import golden_util as G                                     # noqa: E402

CURVES = [("bn254", O.BN254, B.BN254, 3), ("bls12_381", O.BLS12_381, B.BLS12_381, 4)]
MODES = {"fft": (False, False), "ifft": (True, False), "coset_fft": (False, True), "coset_ifft": (True, True)}


def sympy_modes(a, r, g):
    """ark-poly's four entry points over sympy's transform: the coset wrappers are the only glue."""
    ginv = pow(g, r - 2, r)
    return {"fft": ntt(a, r), "ifft": intt(a, r),
            "coset_fft": ntt([x * pow(g, i, r) % r for i, x in enumerate(a)], r),
            "coset_ifft": [x * pow(ginv, i, r) % r for i, x in enumerate(intt(a, r))]}


def from_mont_ints(a, p, n64=4):
    rinv = pow(1 << (64 * n64), -1, p)
    return [B.from_limbs(row) * rinv % p for row in a]


@pytest.mark.parametrize("name,cid,cv,b", CURVES)
def test_sympy_uses_the_reference_generator_and_root(name, cid, cv, b):
    r = cv.fr.p
    g = primitive_root(r)
    assert g == cv.fr.generator == (5 if name == "bn254" else 7)
    s = cv.fr.two_adicity
    w_sympy = pow(g, (r - 1) >> s, r)                 # what sympy's ntt uses for n = 2^s
    w_oracle = from_mont_ints(O.field_const(cid, 0, 3).reshape(1, 4), r)[0]
    assert w_sympy == w_oracle
    doc = G.load_sympy(name)
    assert int(doc["two_adic_root"], 16) == w_oracle and doc["coset_generator"] == g and int(doc["fr_modulus"], 16) == r


@pytest.mark.parametrize("name,cid,cv,b", CURVES)
@pytest.mark.parametrize("log_n", range(1, 11))
def test_oracle_ntt_equals_sympy_live(name, cid, cv, b, log_n):
    """orc_ntt (the checker of every GPU transform test), the 4-step spec (playground.rs:21-80), the distributed helpers
    (worker.rs:66-115) and bigint_ref's domain vs sympy, N = 2 ... 2^10, all four modes."""
    r, g = cv.fr.p, cv.fr.generator
    rng = random.Random(1000 * cid + log_n)
    n = 1 << log_n
    a = [rng.randrange(r) for _ in range(n)]
    if log_n == 3:
        a[:4] = [0, 1, r - 1, r - 2]
    A = G.mont_limbs(a, r)
    want = sympy_modes(a, r, g)
    d = B.Radix2Domain(cv.fr, n)
    ref = {"fft": d.fft, "ifft": d.ifft, "coset_fft": d.coset_fft, "coset_ifft": d.coset_ifft}
    for key, (inv, coset) in MODES.items():
        exp = G.mont_limbs(want[key], r)
        assert np.array_equal(O.ntt(cid, A, inv, coset), exp), (key, "orc_ntt")
        assert ref[key](a) == want[key], (key, "bigint_ref")
        if log_n >= 2:
            assert np.array_equal(O.fourstep(cid, A, inv, coset), exp), (key, "orc_fourstep")
            for S in (1, 2):
                if (1 << (log_n // 2)) >= S:
                    assert np.array_equal(O.distributed_fft(cid, A, S, inv, coset), exp), (key, S, "orc_distributed_fft")


@pytest.mark.parametrize("name,cid,cv,b", CURVES)
def test_oracle_matches_the_sympy_fixtures(name, cid, cv, b):
    """The committed sympy-generated vectors (what the GPU golden test consumes) against both restatements, including the
    2^7 ... 2^12-point digests."""
    doc = G.load_sympy(name)
    r = cv.fr.p
    for e in doc["ntt"]:
        v = G.limbs(e["input_mont"])
        assert from_mont_ints(v, r) == G.sympy_ntt_input(doc, e["log_n"])
        for key, (inv, coset) in MODES.items():
            assert np.array_equal(O.ntt(cid, v, inv, coset), G.limbs(e[key])), (e["log_n"], key)
    for e in doc["ntt_digest"]:
        v = G.mont_limbs(G.sympy_ntt_input(doc, e["log_n"]), r)
        assert G.sha256_limbs(v) == e["input_sha256"]
        for key, (inv, coset) in MODES.items():
            assert G.sha256_limbs(O.ntt(cid, v, inv, coset)) == e[key + "_sha256"], (e["log_n"], key)


def jac_of(xy, inf, cid, q64):
    one = O.field_const(cid, 1, 1)[:q64]
    if inf:
        return np.concatenate([one, one, np.zeros(q64, dtype=np.uint64)])         # arkworks' zero (1, 1, 0)
    return np.concatenate([xy, one])


@pytest.mark.parametrize("name,cid,cv,b", CURVES)
def test_oracle_group_law_matches_the_sympy_fixtures(name, cid, cv, b):
    """orc_jac_add / orc_scalar_mul / orc_msm / orc_msm_naive / orc_sharded_msm and bigint_ref's group law vs sympy's points:
    P + Q, P + P, P - P, the identity on either side, k in {0, 1, 2, r - 1, r, random}, MSMs with an infinity base,
    duplicated bases, P and -P with equal scalars, scalars 0 / 1 / r - 1."""
    doc = G.load_sympy(name)
    q64 = O.FQ_LIMBS[cid]
    q = cv.fq.p

    def aff(j):
        xy, inf = O.jac_to_affine(cid, j)
        return None if inf else xy

    def same(j, P):
        got = aff(j)
        want, winf = G.point_limbs(P, q64)
        return (got is None and winf) or (got is not None and not winf and np.array_equal(got, want))

    for e in doc["group"]:
        a, ainf = G.point_limbs(e["a"], q64)
        if e["op"] == "add":
            bb, binf = G.point_limbs(e["b"], q64)
            assert same(O.jac_add(cid, jac_of(a, ainf, cid, q64), jac_of(bb, binf, cid, q64)), e["out"]), e
        else:
            k = int(e["k"], 16)
            if k < cv.fr.p:                                       # orc_scalar_mul takes a canonical 256-bit scalar
                assert same(O.scalar_mul(cid, a, G.limbs([hex(k)])[0]), e["out"]), e
            # the pure-Python statement, any k
            Pa = tuple(B.from_limbs(x) * pow(cv.fq.R, -1, q) % q for x in (a[:q64], a[q64:]))
            got = B.scalar_mul(cv, k, Pa)
            want = None if e["out"] is None else tuple(int(h, 16) * pow(cv.fq.R, -1, q) % q for h in e["out"])
            assert got == want, e
    for e in doc["msm"]:
        bases, inf = G.bases_from_golden(e, q64)
        sc = G.limbs(e["scalars"])
        for j in (O.msm(cid, bases, sc, inf, threads=1), O.msm(cid, bases, sc, inf, threads=3), O.msm_naive(cid, bases, sc, inf),
                  O.sharded_msm(cid, bases, sc, 3, inf), O.sharded_msm(cid, bases, sc, 4, inf, threads=2)):
            assert same(j, e["result_affine_mont"]), e["case"]


@pytest.mark.parametrize("name,cid,cv,b", CURVES)
def test_oracle_msm_equals_sympy_live(name, cid, cv, b):
    """A fresh MSM every run (not from the fixture): oracle Pippenger == sum_i k_i * P_i by sympy's double-and-add."""
    q64 = O.FQ_LIMBS[cid]
    q, r = cv.fq.p, cv.fr.p
    E = EllipticCurve(0, b, modulus=q)
    gx, gy = from_mont_ints(O.generator(cid).reshape(2, q64), q, q64)
    Gp = E(gx, gy)
    assert (r * Gp).z == 0
    rng = random.Random(77 + cid)
    n = 10
    pts = [rng.randrange(1, r) * Gp for _ in range(n)]
    pts[4] = pts[2]
    sc = [rng.randrange(r) for _ in range(n)]
    sc[5] = 1
    acc = E(0, 1, 0)
    for P, k in zip(pts, sc):
        acc = acc + k * P
    bases = np.zeros((n, 2 * q64), dtype=np.uint64)
    for i, P in enumerate(pts):
        bases[i, :q64] = G.mont_limbs([int(P.x / P.z)], q, q64)[0]
        bases[i, q64:] = G.mont_limbs([int(P.y / P.z)], q, q64)[0]
        assert O.on_curve(cid, bases[i])
    xy, inf = O.jac_to_affine(cid, O.msm(cid, bases, G.limbs([hex(k) for k in sc]), threads=2))
    assert not inf
    assert np.array_equal(xy[:q64], G.mont_limbs([int(acc.x / acc.z)], q, q64)[0])
    assert np.array_equal(xy[q64:], G.mont_limbs([int(acc.y / acc.z)], q, q64)[0])


@pytest.mark.parametrize("name,cid,cv,b", CURVES)
def test_oracle_polynomial_rows_equal_sympy_galoistools(name, cid, cv, b):
    """SURVEY §8f rank 3 (dispatcher2.rs:545-555, 566-633, 651-666) and the blinding of worker.rs:396-406: the oracle's Horner evaluation,
    linear combination, division by (X - z) and (b_0 + b_1 X + ...)(X^n - 1) + p(X) against sympy's dense GF(p) polynomial arithmetic
    (sympy.polys.galoistools: gf_eval, gf_div, gf_mul, gf_add — coefficients highest degree first)."""
    from sympy.polys.domains import ZZ
    from sympy.polys.galoistools import gf_add, gf_div, gf_eval, gf_mul, gf_mul_ground, gf_strip
    r = cv.fr.p
    rng = random.Random(4242 + cid)

    def hi_first(c):            # our vectors are lowest degree first
        return gf_strip([ZZ(x) for x in reversed(c)])

    def lo_first(f, length):
        out = [int(x) % r for x in reversed(f)]
        return out + [0] * (length - len(out))

    for n in (1, 2, 7, 64, 259):
        c = [rng.randrange(r) for _ in range(n)]
        z = rng.randrange(1, r)
        C, Z = G.mont_limbs(c, r), G.mont_limbs([z], r)[0]
        assert from_mont_ints(O.poly_eval(cid, C, Z).reshape(1, 4), r)[0] == int(gf_eval(hi_first(c), ZZ(z), r, ZZ)) % r, ("eval", n)
        if n >= 2:
            q, rem = gf_div(hi_first(c), [ZZ(1), ZZ(-z % r)], r, ZZ)
            assert from_mont_ints(O.poly_div_linear(cid, C, Z), r) == lo_first(q, n - 1), ("div", n)
            assert [int(x) % r for x in rem] in ([], [int(gf_eval(hi_first(c), ZZ(z), r, ZZ)) % r])        # remainder = p(z)
    # linear combination of polynomials of different lengths
    polys = [[rng.randrange(r) for _ in range(ln)] for ln in (5, 9, 1, 12)]
    ks = [rng.randrange(r) for _ in polys]
    acc = []
    for p_, k_ in zip(polys, ks):
        acc = gf_add(acc, gf_mul_ground(hi_first(p_), ZZ(k_), r, ZZ), r, ZZ)
    got = O.poly_lincomb(cid, [G.mont_limbs(p_, r) for p_ in polys], G.mont_limbs(ks, r))
    assert from_mont_ints(got, r) == lo_first(acc, 12)
    # blinding: (b_0 + b_1 X [+ b_2 X^2]) * (X^n - 1) + p(X)
    for n, kb in ((8, 2), (16, 3)):
        p_ = [rng.randrange(r) for _ in range(n)]
        bl = [rng.randrange(r) for _ in range(kb)]
        zh = [ZZ(1)] + [ZZ(0)] * (n - 1) + [ZZ(r - 1)]
        want = gf_add(gf_mul(hi_first(bl), zh, r, ZZ), hi_first(p_), r, ZZ)
        assert from_mont_ints(O.blind(cid, G.mont_limbs(p_, r), n, G.mont_limbs(bl, r)), r) == lo_first(want, n + kb)


@pytest.mark.parametrize("name,cid,cv,b", CURVES)
def test_committed_sympy_fixture_is_what_the_generator_writes(name, cid, cv, b):
    """The fixture the GPU golden test consumes IS sympy's output: the generator's own functions, run here on the smaller entries, reproduce the
    committed file (a hand-edited or stale tests/golden/sympy_*.json fails) — transforms up to 2^5 points in full, the 2^7-point digests."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_golden_sympy", os.path.join(root, "tools", "gen_golden_sympy.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    doc = G.load_sympy(name)
    r = cv.fr.p
    R = pow(2, 256, r)
    assert gen.CURVES[name]["r"] == r and doc["ntt_seed"] == gen.NTT_SEED
    for e in doc["ntt"]:
        if e["log_n"] > 5:
            continue
        a = gen.ntt_input(r, e["log_n"])
        modes, _ = gen.four_modes(a, r, primitive_root(r))
        assert e["input_mont"] == [hex(x * R % r) for x in a]
        for k, v in modes.items():
            assert e[k] == [hex(int(x) * R % r) for x in v], (e["log_n"], k)
    e = next(x for x in doc["ntt_digest"] if x["log_n"] == 7)
    a = gen.ntt_input(r, 7)
    modes, _ = gen.four_modes(a, r, primitive_root(r))
    assert e["input_sha256"] == gen.digest(a, R, r) and all(e[k + "_sha256"] == gen.digest(v, R, r) for k, v in modes.items())
