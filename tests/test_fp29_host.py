"""The 9 x 29-bit limb arithmetic of csrc/fp29.hpp compiled for the HOST (its functions are __host__ __device__) against Python
integers — no GPU: the precomputed-quotient ("Shoup") constant multiplier the NTT butterflies use on BN254, its constant
preparation, the Montgomery multiplier on the same operands, and the product-free canonicalisation up to the bound the Shoup
butterflies reach (< 48p).  Operand ranges are the ones a butterfly produces: un-normalised limbs < 2^31, value < 2^259.4."""
import ctypes as C
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
     1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}
MASK = (1 << 29) - 1
R256 = 1 << 256


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fp29") / "fp29_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "host_cpp", "fp29_host.cpp"), "-o", so])
    return C.CDLL(so)


def limbs29(v, top_free=False):
    out = [(v >> (29 * k)) & MASK for k in range(9)]
    if top_free:
        out[8] = v >> (29 * 8)
    return out


def value(l):
    return sum(int(x) << (29 * k) for k, x in enumerate(l))


def lazy_limbs(rng, bound):
    """a value below `bound` written with un-normalised limbs below 2^31 (what a butterfly hands to its product)"""
    v = rng.randrange(bound)
    l = limbs29(v, top_free=True)
    for k in range(8):
        d = min(l[k + 1], 3, ((1 << 31) - 1 - l[k]) >> 29)
        d = rng.randrange(d + 1) if d > 0 else 0
        l[k] += d << 29
        l[k + 1] -= d
    assert value(l) == v and all(0 <= x < (1 << 31) for x in l)
    return l, v


@pytest.mark.parametrize("curve", [0, 1])
def test_shoup_multiplier_against_integers(lib, curve):
    p = P[curve]
    n = 40000
    rng = random.Random(0x5A0F + curve)
    # BN254: f29_mul's documented contract.  BLS12-381 (p = 2^254.86): the Shoup butterflies reach 1.6p + 4p * 9 < 38p there, so both
    # multipliers are checked up to 40p = 2^260.2 (the Shoup quotient estimate needs x < 2^261; the Montgomery result is < x*p/2^261 + p < 2p)
    bound = int(2 ** 259.4) if curve == 0 else 40 * p
    pbar = (C.c_uint32 * 9)()
    lib.get_pbar(curve, pbar)
    assert value(pbar) == (1 << 261) - p and all(x <= MASK for x in pbar)
    edge_c = [0, 1, 2, p - 1, p - 2, (p + 1) // 2, 1 << 253]
    edge_x = [0, 1, p - 1, p, 2 * p, bound - 1]
    xs, cs, vals = [], [], []
    for i in range(n):
        c = edge_c[i % len(edge_c)] if i < 64 else rng.randrange(p)
        if i < 64:
            xv = edge_x[(i // len(edge_c)) % len(edge_x)]
            xl = limbs29(xv, top_free=True)
        else:
            xl, xv = lazy_limbs(rng, bound)
        xs.append(xl); cs.append(c); vals.append(xv)
    A = (C.c_uint32 * (9 * n))(*[w for l in xs for w in l])
    c29, cq29 = (C.c_uint32 * (9 * n))(), (C.c_uint32 * (9 * n))()
    one_c, one_q, c8 = (C.c_uint32 * 9)(), (C.c_uint32 * 9)(), (C.c_uint32 * 8)()
    for k, c in enumerate(cs):
        cm = c * R256 % p                                   # the reference's Montgomery form, as the tables are built from
        for i in range(8):
            c8[i] = (cm >> (32 * i)) & 0xffffffff
        lib.shoup_const(curve, c8, one_c, one_q)
        assert value(one_c) == c and value(one_q) == (c << 261) // p, "prepared constant"
        c29[9 * k:9 * k + 9] = one_c[:]
        cq29[9 * k:9 * k + 9] = one_q[:]
    R = (C.c_uint32 * (9 * n))()
    lib.shoup_mul(curve, A, c29, cq29, R, C.c_long(n))
    for k in range(n):
        r = R[9 * k:9 * k + 9]
        assert all(x <= MASK for x in r), "normalised limbs"
        rv = value(r)
        assert rv % p == vals[k] * cs[k] % p, (curve, k)
        assert 0 <= rv < 3 * p, (curve, k, rv / p)
    cm = (C.c_uint32 * (9 * n))(*[w for c in cs for w in limbs29((c << 261) % p)])
    M = (C.c_uint32 * (9 * n))()
    lib.mont_mul(curve, A, cm, M, C.c_long(n))
    for k in range(0, n, 7):
        mv = value(M[9 * k:9 * k + 9])
        assert mv % p == vals[k] * cs[k] % p and mv < 2 * p


@pytest.mark.parametrize("curve", [0, 1])
def test_canon_lazy_up_to_48p(lib, curve):
    p = P[curve]
    n = 60000
    rng = random.Random(77 + curve)
    top = min(48 * p, (1 << 261) - 1)
    vals = [0, 1, p - 1, p, p + 1, 2 * p - 1, 2 * p, 24 * p, 36 * p + 5, top - 1] + [k * p + d for k in range(1, 48) for d in (-1, 0, 1) if 0 <= k * p + d < top]
    vals += [rng.randrange(top) for _ in range(n - len(vals))]
    A = (C.c_uint32 * (9 * len(vals)))(*[w for v in vals for w in limbs29(v, top_free=True)])
    Rr = (C.c_uint32 * (9 * len(vals)))()
    lib.canon_lazy(curve, A, Rr, C.c_long(len(vals)))
    for k, v in enumerate(vals):
        r = Rr[9 * k:9 * k + 9]
        assert value(r) == v % p and all(x <= MASK for x in r), (curve, v // p)
