"""EXPERIMENT (round-3 preparation, CPU only): correctness of the precomputed-quotient constant multiplier of f29_shoup.hpp against
Python integers, on the operand ranges the NTT butterflies produce (limbs < 2^31, value < 2^259.4), both scalar fields.
    python tools/experiments/check_f29_shoup.py [count]"""
import ctypes as C
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join("/tmp", "f29_shoup_host.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(HERE, "f29_shoup_host.cpp"), "-o", SO])
lib = C.CDLL(SO)
P = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
     1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}
MASK = (1 << 29) - 1


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
    for k in range(8):                         # push random excess down: l[k] += d * 2^29, l[k+1] -= d, keeping every limb in range
        d = min(l[k + 1], 3, ((1 << 31) - 1 - l[k]) >> 29)
        d = rng.randrange(d + 1) if d > 0 else 0
        l[k] += d << 29
        l[k + 1] -= d
    assert value(l) == v and all(0 <= x < (1 << 31) for x in l)
    return l, v


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rng = random.Random(0x5A0F)
    for curve, p in P.items():
        bound = int(2 ** 259.4)
        xs, cs, vals = [], [], []
        edge_c = [0, 1, 2, p - 1, p - 2, (p + 1) // 2, 1 << 253]
        edge_x = [0, 1, p - 1, p, 2 * p, bound - 1]
        for i in range(count):
            c = edge_c[i % len(edge_c)] if i < 64 else rng.randrange(p)
            if i < 64:
                xv = edge_x[(i // len(edge_c)) % len(edge_x)]
                xl = limbs29(xv, top_free=True)
            else:
                xl, xv = lazy_limbs(rng, bound)
            xs.append(xl); cs.append(c); vals.append(xv)
        n = count
        A = (C.c_uint32 * (9 * n))(*[w for l in xs for w in l])
        c29 = (C.c_uint32 * (9 * n))()
        cq29 = (C.c_uint32 * (9 * n))()
        one_c, one_q, c8 = (C.c_uint32 * 9)(), (C.c_uint32 * 9)(), (C.c_uint32 * 8)()
        for k, c in enumerate(cs):
            for i in range(8):
                c8[i] = (c >> (32 * i)) & 0xffffffff
            lib.shoup_const(curve, c8, one_c, one_q)
            assert value(one_c) == c and value(one_q) == (c << 261) // p, "quotient constant"
            c29[9 * k:9 * k + 9] = one_c[:]
            cq29[9 * k:9 * k + 9] = one_q[:]
        R = (C.c_uint32 * (9 * n))()
        lib.shoup_mul(curve, A, c29, cq29, R, C.c_long(n))
        worst = 0
        for k in range(n):
            r = R[9 * k:9 * k + 9]
            assert all(x <= MASK for x in r), "normalised limbs"
            rv = value(r)
            exact = vals[k] * cs[k]
            assert rv % p == exact % p, (curve, k)
            assert 0 <= rv < 3 * p, (curve, k, rv / p)
            worst = max(worst, (rv - exact % p) // p)
        # the product path's multiplier on the same operands gives the same residue (different representative)
        cm = (C.c_uint32 * (9 * n))(*[w for c in cs for w in limbs29((c << 261) % p)])
        M = (C.c_uint32 * (9 * n))()
        lib.mont_mul(curve, A, cm, M, C.c_long(n))
        for k in range(0, n, 97):
            assert value(M[9 * k:9 * k + 9]) % p == (vals[k] * cs[k]) % p
        print(f"curve {curve}: {n} products ok, result < 3p everywhere (largest multiple of p above the canonical value: {worst})")


if __name__ == "__main__":
    main()
