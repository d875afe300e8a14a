#!/usr/bin/env python3
"""Generate tests/golden/sympy_*.json from THIRD-PARTY code: sympy's number-theoretic transform and elliptic-curve group law.

Why a second fixture set: the reference holds no golden vectors and its arithmetic (arkworks 0.3.0) cannot be run
here, so tests/golden/{bn254,bls12_381}.json come from this repository's own big-integer restatement (tools/gen_golden.py).  These
files do not: every expected value below is computed by `sympy.discrete.transforms.ntt / intt` and by
`sympy.ntheory.elliptic_curve.EllipticCurve` (affine chord-and-tangent over a ModularInteger field) — code this repository neither
wrote nor uses anywhere else.  Nothing from oracle/ or distributed_plonk_amd/ is imported.

What this script itself contributes is only the glue the reference's callers also apply around the library calls:
  * the moduli and the curve equations y^2 = x^3 + 3 / + 4,
  * ark-poly's coset wrappers: coset_fft = fft of (a_i * g^i), coset_ifft = (ifft)_i * g^-i   (Radix2EvaluationDomain),
  * the Montgomery encoding a -> a * 2^(64 * limbs) mod p that the reference's in-memory layout uses (utils.rs:27-43).
sympy's transform takes w = primitive_root(p)^((p - 1) / n): primitive_root returns the SMALLEST generator, which is arkworks'
`GENERATOR` (5 on BN254 Fr, 7 on BLS12-381 Fr) — asserted below — so it is the same w ark-poly derives from TWO_ADIC_ROOT_OF_UNITY.

Transforms up to 2^6 points are stored in full; 2^7 ... 2^12 as the SHA-256 of the little-endian Montgomery output bytes (the
inputs of those are regenerated from the recorded seed by the same `random.Random` calls — see tests/golden_util.sympy_ntt_input).

    python tools/gen_golden_sympy.py
"""
import hashlib
import json
import os
import random

from sympy.discrete.transforms import intt, ntt
from sympy.ntheory import isprime, primitive_root, sqrt_mod
from sympy.ntheory.elliptic_curve import EllipticCurve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

CURVES = {
    "bn254": dict(
        r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
        q=21888242871839275222246405745257275088696311157297823662689037894645226208583,
        b=3, gen=(1, 2), q64=4, two_adicity=28),
    "bls12_381": dict(
        r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
        q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
        b=4, gen=(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
                  0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
        q64=6, two_adicity=32),
}
FULL_UP_TO = 6
DIGEST_UP_TO = 12
NTT_SEED = 0x5A17


def hx(v):
    return hex(int(v))


def ntt_input(r, log_n):
    """Canonical inputs of the 2^log_n-point vectors: the fixture's readers regenerate them with the same calls."""
    rng = random.Random((NTT_SEED << 8) | log_n)
    return [rng.randrange(r) for _ in range(1 << log_n)]


def four_modes(a, r, g):
    """ark-poly's four entry points on top of sympy's ntt / intt."""
    n = len(a)
    ginv = pow(g, r - 2, r)
    shifted = [x * pow(g, i, r) % r for i, x in enumerate(a)]
    return {
        "fft": ntt(a, r),
        "ifft": intt(a, r),
        "coset_fft": ntt(shifted, r),
        "coset_ifft": [x * pow(ginv, i, r) % r for i, x in enumerate(intt(a, r))],
    }, n


def digest(vals, R, p):
    h = hashlib.sha256()
    for v in vals:
        h.update((int(v) * R % p).to_bytes(32, "little"))
    return h.hexdigest()


def coords(P):
    """sympy point -> (x, y) integers or None for the point at infinity."""
    if P.z == 0:
        return None
    return int(P.x / P.z) , int(P.y / P.z)


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, cv in CURVES.items():
        r, q, q64 = cv["r"], cv["q"], cv["q64"]
        assert isprime(r) and isprime(q)
        R, Rq = pow(2, 256, r), pow(2, 64 * q64, q)
        g = primitive_root(r)
        assert g == (5 if name == "bn254" else 7), "sympy's smallest primitive root must be arkworks' GENERATOR"
        s = cv["two_adicity"]
        assert (r - 1) % (1 << s) == 0 and ((r - 1) >> s) & 1
        doc = {"generated_by": "tools/gen_golden_sympy.py (sympy %s: discrete.transforms.ntt/intt, ntheory.elliptic_curve)"
                               % __import__("sympy").__version__,
               "curve": name, "fr_modulus": hx(r), "fq_modulus": hx(q), "coset_generator": g,
               "two_adic_root": hx(pow(g, (r - 1) >> s, r)), "ntt_seed": NTT_SEED, "ntt": [], "ntt_digest": [],
               "group": [], "msm": []}

        # ---- transforms -------------------------------------------------------------------------------------------
        for log_n in range(1, DIGEST_UP_TO + 1):
            a = ntt_input(r, log_n)
            modes, _ = four_modes(a, r, g)
            if log_n <= FULL_UP_TO:
                e = {"log_n": log_n, "input_mont": [hx(x * R % r) for x in a]}
                for k, v in modes.items():
                    e[k] = [hx(x * R % r) for x in v]
                doc["ntt"].append(e)
            else:
                e = {"log_n": log_n, "input_sha256": digest(a, R, r)}
                for k, v in modes.items():
                    e[k + "_sha256"] = digest(v, R, r)
                doc["ntt_digest"].append(e)

        # ---- group law ----------------------------------------------------------------------------------------------
        E = EllipticCurve(0, cv["b"], modulus=q)
        G = E(*cv["gen"])
        assert (r * G).z == 0, "the generator has order r"
        rng = random.Random(0xEC0 + q64)

        def mont_pt(P):
            c = coords(P)
            return None if c is None else [hx(c[0] * Rq % q), hx(c[1] * Rq % q)]

        ks = [rng.randrange(1, r) for _ in range(6)]
        pts = [k * G for k in ks]
        # a point outside the prime-order subgroup on BLS12-381 would not be a legal base; BN254 has cofactor 1
        for (i, j) in ((0, 1), (2, 3), (4, 4), (5, 5)):
            doc["group"].append({"op": "add", "a": mont_pt(pts[i]), "b": mont_pt(pts[j]), "out": mont_pt(pts[i] + pts[j])})
        doc["group"].append({"op": "add", "a": mont_pt(pts[0]), "b": mont_pt(-pts[0]), "out": None})            # P - P
        doc["group"].append({"op": "add", "a": mont_pt(pts[1]), "b": None, "out": mont_pt(pts[1])})             # P + inf
        doc["group"].append({"op": "add", "a": None, "b": mont_pt(pts[2]), "out": mont_pt(pts[2])})             # inf + P
        for k in (0, 1, 2, r - 1, r, rng.randrange(r), rng.randrange(1 << 64)):
            doc["group"].append({"op": "mul", "a": mont_pt(pts[3]), "k": hx(k), "out": mont_pt(k * pts[3])})

        # ---- multi-scalar multiplications (double-and-add by sympy, summed by sympy) ----------------------------------
        for case, n in (("mixed", 12), ("tiled", 40)):
            if case == "mixed":
                bases = [rng.randrange(1, r) * G for _ in range(n)]
                bases[3] = E(0, 1, 0)                 # an infinity base (dispatcher2.rs:1101)
                bases[7] = bases[6]                   # P + P inside one bucket
                bases[9] = -bases[8]                  # P - P
                sc = [rng.randrange(r) for _ in range(n)]
                sc[0], sc[1], sc[2] = 0, 1, r - 1
                sc[7] = sc[6]
                sc[9] = sc[8]
            else:                                     # the reference's test shape: a few points tiled (dispatcher.rs:190-200)
                few = [rng.randrange(1, r) * G for _ in range(5)]
                bases = [few[i % 5] for i in range(n)]
                sc = [rng.randrange(r) for _ in range(n)]
            acc = E(0, 1, 0)
            for P, k in zip(bases, sc):
                acc = acc + k * P
            doc["msm"].append({"case": case, "bases_mont": [mont_pt(P) for P in bases], "scalars": [hx(k) for k in sc],
                               "result_affine_mont": mont_pt(acc)})

        path = os.path.join(OUT, f"sympy_{name}.json")
        with open(path, "w") as fh:
            json.dump(doc, fh, indent=1)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
