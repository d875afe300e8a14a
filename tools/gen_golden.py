#!/usr/bin/env python3
"""Generate tests/golden/*.json from the pure-Python big-int statement (oracle/bigint_ref.py).

The reference holds no golden vectors and cannot be run here, so these are NOT
reference outputs: they are exact-integer results of the published arkworks algorithms, committed so
that the C oracle and the HIP path are both pinned to fixed bytes (and so that a change in either shows
up as a diff).  Values are hex strings of the reference's in-memory representation: Fr = a*2^256 mod r
(Montgomery), scalars canonical, points affine (x, y) Montgomery or null for infinity.

    python tools/gen_golden.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bigint_ref as B  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def hx(v):
    return hex(v)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = random.Random(0xC0FFEE)
    for cname, cv in B.CURVES.items():
        f = cv.fr
        doc = {"curve": cname, "fr_modulus": hx(f.p), "fq_modulus": hx(cv.fq.p), "fr_R": hx(f.R), "fq_R": hx(cv.fq.R),
               "two_adicity": f.two_adicity, "coset_generator": f.generator,
               "two_adic_root": hx(pow(f.generator, (f.p - 1) >> f.two_adicity, f.p)), "ntt": [], "msm": [], "field_mul": []}
        # field multiplication KATs (Montgomery in/out)
        for fld, ff in (("fr", f), ("fq", cv.fq)):
            for _ in range(4):
                a, b = rng.randrange(ff.p), rng.randrange(ff.p)
                doc["field_mul"].append({"field": fld, "a": hx(a), "b": hx(b), "mont_mul": hx(a * b * pow(ff.R, -1, ff.p) % ff.p)})
        # NTT KATs: canonical values -> Montgomery limbs
        for log_n in (1, 3, 6):
            n = 1 << log_n
            v = [rng.randrange(f.p) for _ in range(n)]
            d = B.Radix2Domain(f, n)
            entry = {"log_n": log_n, "input_mont": [hx(f.to_mont(x)) for x in v]}
            for name, fn in (("fft", d.fft), ("ifft", d.ifft), ("coset_fft", d.coset_fft), ("coset_ifft", d.coset_ifft)):
                entry[name] = [hx(f.to_mont(x)) for x in fn(v)]
            if n >= 4:     # the reference's 2-D decomposition must agree (playground.rs:95-99)
                assert B.fourstep(f, n, v, False, False) == d.fft(v)
                assert B.distributed_fft(f, n, v, 2, True, True) == d.coset_ifft(v)
            doc["ntt"].append(entry)
        # MSM KATs: duplicated bases, an infinity base, scalars 0, 1, r-1
        pts = B.rand_points(cv, 4, rng) * 3
        pts[5] = None
        sc = [rng.randrange(f.p) for _ in pts]
        sc[0], sc[1], sc[2] = 0, 1, f.p - 1
        res = B.msm_naive(cv, pts, sc)
        assert res == B.msm_pippenger(cv, pts, sc) == B.sharded_msm(cv, pts, sc, 3)
        q = cv.fq
        doc["msm"].append({
            "bases_mont": [None if P is None else [hx(q.to_mont(P[0])), hx(q.to_mont(P[1]))] for P in pts],
            "scalars": [hx(s) for s in sc],
            "result_affine_mont": None if res is None else [hx(q.to_mont(res[0])), hx(q.to_mont(res[1]))]})
        with open(os.path.join(OUT, f"{cname}.json"), "w") as fh:
            json.dump(doc, fh, indent=1)
        print("wrote", cname)


if __name__ == "__main__":
    main()
