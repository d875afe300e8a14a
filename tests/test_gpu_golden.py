"""HIP path vs the committed golden fixtures (tests/golden/*.json from tools/gen_golden.py = the repo's own big-int statement;
tests/golden/sympy_*.json from tools/gen_golden_sympy.py = THIRD-PARTY sympy transforms and group law; and sympy live) — and the
arkworks in-memory SRS layout that `init` receives on the wire."""
import numpy as np
import pytest

import golden_util as G
from distributed_plonk_amd import _ffi
from distributed_plonk_amd._ffi import MsmWorkload

pytestmark = pytest.mark.gpu

KEYS = {"fft": (False, False), "ifft": (True, False), "coset_fft": (False, True), "coset_ifft": (True, True)}


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_golden_field_ntt_msm(gpu_workers, curve):
    w = gpu_workers(curve)
    g = G.load(curve)
    q = w.q64
    for e in g["field_mul"]:
        fld = 0 if e["field"] == "fr" else 1
        n64 = 4 if fld == 0 else q
        got = w.field_op(fld, 0, G.limbs([e["a"]], n64), G.limbs([e["b"]], n64))
        assert np.array_equal(got, G.limbs([e["mont_mul"]], n64))
    for e in g["ntt"]:
        v = G.limbs(e["input_mont"])
        for key, (inv, coset) in KEYS.items():
            assert np.array_equal(w.ntt(v, inv, coset), G.limbs(e[key])), (e["log_n"], key)
    for e in g["msm"]:
        bases, inf = G.bases_from_golden(e, q)
        sc = G.limbs(e["scalars"])
        w.init(bases, 0, 0)                       # infinity = (0, 0) in the XY layout
        xy, isinf = w.g1_to_affine(w.var_msm(MsmWorkload(0, len(sc)), sc))
        want = e["result_affine_mont"]
        assert not isinf
        assert np.array_equal(xy[:q], G.limbs([want[0]], q)[0]) and np.array_equal(xy[q:], G.limbs([want[1]], q)[0])


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_init_accepts_arkworks_memory_layout(gpu_workers, oracle, curve, cid):
    """`init` ships `serialize(&[G1Affine])` (utils.rs:27-33, dispatcher.rs:58-64): x, y, infinity: bool, padded
    to 8 bytes — 72 B (BN254) / 104 B (BLS12-381) per point; arkworks' affine zero is (0, 1, true)."""
    w = gpu_workers(curve)
    q = w.q64
    n = 300
    bases = oracle.gen_bases(cid, 31, 50, n)
    inf = np.zeros(n, dtype=np.uint8)
    inf[[0, 17, 299]] = 1
    stride = 16 * q + 8
    raw = np.zeros((n, stride), dtype=np.uint8)
    raw[:, :16 * q] = bases.view(np.uint8).reshape(n, 16 * q)
    one = oracle.field_const(cid, 1, 1).view(np.uint8)
    for i in np.nonzero(inf)[0]:
        raw[i, :8 * q] = 0
        raw[i, 8 * q:16 * q] = one          # (0, 1, true)
        raw[i, 16 * q] = 1
    raw[:, 16 * q + 1:] = 0xAB                # padding bytes are undefined in Rust: must be ignored
    sc = oracle.from_mont(cid, oracle.rand_fr(cid, 32, n))
    w.init(raw, 0, 0, layout=_ffi.PLONK_BASES_ARK)
    got, gi = w.g1_to_affine(w.var_msm(MsmWorkload(0, n), sc))
    exp, ei = oracle.jac_to_affine(cid, oracle.msm(cid, bases, sc, inf, threads=4))
    assert gi == ei and np.array_equal(got, exp)


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_init_refuses_a_mislaid_srs(gpu_workers, oracle, curve, cid):
    """VERDICT r4 item 4 / SURVEY a15: `G1Affine {x, y, infinity}` is a default-repr Rust struct (field order not guaranteed,
    utils.rs:27-43) and the worker reinterprets its memory (worker.rs:136-141).  plonk_init checks every base: swapped coordinates, an
    ark buffer read at a shifted offset, a flag byte that is no bool, an unreduced coordinate and a single corrupted point are all
    PLONK_ERR_ARG naming the offender — and leave the context without bases — instead of commitments that are garbage with PLONK_OK."""
    from distributed_plonk_amd._ffi import PlonkError
    w = gpu_workers(curve)
    q = w.q64
    n = 257
    bases = oracle.gen_bases(cid, 77, 40, n)                      # (n, 2 * 4q) u64: x || y
    sc = oracle.from_mont(cid, oracle.rand_fr(cid, 78, n))

    def refused(arr, layout, needle, index=None):
        with pytest.raises(PlonkError) as e:
            w.init(arr, 0, 0, layout=layout)
        assert e.value.code == -1 and needle in str(e.value), str(e.value)
        if index is not None:
            assert f"base {index} of" in str(e.value), str(e.value)
        with pytest.raises(PlonkError):                           # nothing was installed
            w.var_msm(MsmWorkload(0, n), sc)

    # XY layout: y || x
    half = bases.shape[1] // 2
    swapped = np.concatenate([bases[:, half:], bases[:, :half]], axis=1)
    refused(np.ascontiguousarray(swapped), _ffi.PLONK_BASES_XY, "y^2 != x^3 + b", 0)
    # one corrupted point deep in the vector
    bad = bases.copy()
    bad[200, 0] ^= np.uint64(1)
    refused(bad, _ffi.PLONK_BASES_XY, "y^2 != x^3 + b", 200)
    # an unreduced coordinate: x + p has the same residue but is not what ark-ff stores
    bad = bases.copy()
    bad[5, :bases.shape[1] // 2] = 0xFFFFFFFFFFFFFFFF
    refused(bad, _ffi.PLONK_BASES_XY, "not below the modulus", 5)
    # the arkworks layout, read one byte late (a mis-declared stride / offset on the Rust side) and with a non-bool flag byte
    stride = 16 * q + 8
    raw = np.zeros((n, stride), dtype=np.uint8)
    raw[:, :16 * q] = bases.view(np.uint8).reshape(n, 16 * q)
    shifted = np.zeros(n * stride, dtype=np.uint8)
    shifted[:-1] = raw.reshape(-1)[1:]
    refused(shifted.reshape(n, stride), _ffi.PLONK_BASES_ARK, "not a curve point")
    flagged = raw.copy()
    flagged[9, 16 * q] = 7
    refused(flagged, _ffi.PLONK_BASES_ARK, "neither 0 nor 1", 9)
    # (0, 1) without the flag is a finite off-curve point in XY (Prover's docstring): refused; with the flag in ARK it is infinity: accepted
    zero_one = bases.copy()
    zero_one[3] = 0
    zero_one[3, bases.shape[1] // 2:] = oracle.field_const(cid, 1, 1)[:bases.shape[1] // 2]
    refused(zero_one, _ffi.PLONK_BASES_XY, "y^2 != x^3 + b", 3)
    # the check can be switched off, and the good SRS still installs
    w.set_option("check_bases", 0)
    w.init(np.ascontiguousarray(swapped), 0, 0, layout=_ffi.PLONK_BASES_XY)
    w.set_option("check_bases", 1)
    w.init(bases, 0, 0)
    got, gi = w.g1_to_affine(w.var_msm(MsmWorkload(0, n), sc))
    exp, ei = oracle.jac_to_affine(cid, oracle.msm(cid, bases, sc, threads=4))
    assert gi == ei and np.array_equal(got, exp)


# ---- third-party expected values (tools/gen_golden_sympy.py): sympy's transform and group law, not this repository's restatement -----

def _same_point(w, jac, P):
    q = w.q64
    xy, isinf = w.g1_to_affine(jac)
    want, winf = G.point_limbs(P, q)
    return (isinf and winf) or (not isinf and not winf and np.array_equal(xy, want))


def _jac(w, oracle_one, P):
    q = w.q64
    xy, inf = G.point_limbs(P, q)
    if inf:
        return np.concatenate([oracle_one, oracle_one, np.zeros(q, dtype=np.uint64)])        # arkworks' zero (1, 1, 0)
    return np.concatenate([xy, oracle_one])


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_sympy_generated_fixtures(gpu_workers, curve):
    """The HIP path against tests/golden/sympy_*.json: every transform mode for 2 ... 2^6 points element by element, 2^7 ... 2^12 by
    SHA-256 of the output bytes, the device group law (P + Q, P + P, P - P, identity) and MSMs with an infinity base, duplicated
    bases, P / -P pairs, scalars 0 / 1 / r - 1 and the reference's tiled-bases shape (dispatcher.rs:190-200)."""
    w = gpu_workers(curve)
    doc = G.load_sympy(curve)
    r, qmod, q = int(doc["fr_modulus"], 16), int(doc["fq_modulus"], 16), w.q64
    for e in doc["ntt"]:
        v = G.limbs(e["input_mont"])
        for key, (inv, coset) in KEYS.items():
            assert np.array_equal(w.ntt(v, inv, coset), G.limbs(e[key])), (e["log_n"], key)
    for e in doc["ntt_digest"]:
        v = G.mont_limbs(G.sympy_ntt_input(doc, e["log_n"]), r)
        assert G.sha256_limbs(v) == e["input_sha256"]
        for key, (inv, coset) in KEYS.items():
            assert G.sha256_limbs(w.ntt(v, inv, coset)) == e[key + "_sha256"], (e["log_n"], key)
    one = G.limbs([hex((1 << (64 * q)) % qmod)], q)[0]
    for e in doc["group"]:
        if e["op"] == "add":
            assert _same_point(w, w.g1_add(_jac(w, one, e["a"]), _jac(w, one, e["b"])), e["out"]), e
        elif int(e["k"], 16) < r:                                  # k * P as a one-term MSM (var_msm takes canonical scalars)
            bases, inf = G.bases_from_golden({"bases_mont": [e["a"]]}, q)
            w.init(bases, 0, 0)
            assert _same_point(w, w.var_msm(MsmWorkload(0, 1), G.limbs([e["k"]])), e["out"]), e
    for e in doc["msm"]:
        bases, inf = G.bases_from_golden(e, q)
        sc = G.limbs(e["scalars"])
        w.init(bases, 0, 0)                                       # infinity = (0, 0) in the XY layout
        assert _same_point(w, w.var_msm(MsmWorkload(0, len(sc)), sc), e["result_affine_mont"]), e["case"]
        # sharded by index range like dispatcher.rs:218-238, partial points added on the device
        acc = None
        step = (len(sc) + 2) // 3
        for s in range(0, len(sc), step):
            part = w.var_msm(MsmWorkload(s, min(s + step, len(sc))), sc[s:s + step])
            acc = part if acc is None else w.g1_add(acc, part)
        assert _same_point(w, acc, e["result_affine_mont"]), (e["case"], "sharded")


@pytest.mark.parametrize("curve,b", [("bn254", 3), ("bls12_381", 4)])
def test_hip_equals_sympy_live(gpu_workers, curve, b):
    """No oracle and no fixture in between: fresh inputs, sympy computes the expected transform and the expected MSM point here."""
    sympy = pytest.importorskip("sympy")
    import random
    from sympy.discrete.transforms import intt, ntt
    from sympy.ntheory import primitive_root
    from sympy.ntheory.elliptic_curve import EllipticCurve
    w = gpu_workers(curve)
    doc = G.load_sympy(curve)
    r, qmod, q = int(doc["fr_modulus"], 16), int(doc["fq_modulus"], 16), w.q64
    g = primitive_root(r)
    ginv = pow(g, r - 2, r)
    rng = random.Random()                                          # unseeded on purpose; the inputs are printed on failure
    for log_n in (1, 4, 9, 11, 13):
        a = [rng.randrange(r) for _ in range(1 << log_n)]
        v = G.mont_limbs(a, r)
        want = {"fft": ntt(a, r), "ifft": intt(a, r), "coset_fft": ntt([x * pow(g, i, r) % r for i, x in enumerate(a)], r),
                "coset_ifft": [x * pow(ginv, i, r) % r for i, x in enumerate(intt(a, r))]}
        for key, (inv, coset) in KEYS.items():
            assert np.array_equal(w.ntt(v, inv, coset), G.mont_limbs(want[key], r)), (log_n, key, a[:4])
    E = EllipticCurve(0, b, modulus=qmod)
    gen = E(1, 2) if curve == "bn254" else E(
        0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
        0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)
    n = 16
    ks = [rng.randrange(1, r) for _ in range(n)]
    pts = [k * gen for k in ks]
    sc = [rng.randrange(r) for _ in range(n)]
    acc = E(0, 1, 0)
    for P, k in zip(pts, sc):
        acc = acc + k * P
    bases = np.zeros((n, 2 * q), dtype=np.uint64)
    for i, P in enumerate(pts):
        bases[i, :q] = G.mont_limbs([int(P.x / P.z)], qmod, q)[0]
        bases[i, q:] = G.mont_limbs([int(P.y / P.z)], qmod, q)[0]
    w.init(bases, 0, 0)
    xy, isinf = w.g1_to_affine(w.var_msm(MsmWorkload(0, n), G.limbs([hex(k) for k in sc])))
    assert not isinf, (ks, sc)
    assert np.array_equal(xy[:q], G.mont_limbs([int(acc.x / acc.z)], qmod, q)[0]), (ks, sc)
    assert np.array_equal(xy[q:], G.mont_limbs([int(acc.y / acc.z)], qmod, q)[0]), (ks, sc)
