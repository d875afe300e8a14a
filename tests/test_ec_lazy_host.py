"""The MSM's device arithmetic compiled for the HOST against Python integers — no GPU: the unsaturated-limb Montgomery field of
csrc/flimb.hpp (9 x 29-bit limbs on BN254 Fq, 14 x 28-bit on BLS12-381 Fq; product, square, the two-product `fl_dot2`, the lifted
subtraction constants) and the lazy XYZZ formulas of csrc/ec_lazy.hpp that `msm_accumulate_kernel`, its redo / heavy kernels and the
reduction pyramid run (mixed addition with and without the fused Y3, complete addition, doubling, the fast paths' same-x abort).
Every result is compared with affine integer arithmetic on the curve, and the invariants ec_lazy.hpp states (X < 5.2p, Y < 3.3p,
ZZ, ZZZ < 2p, normalised limbs) are checked after every operation — also when the operands are lifted to the TOP of those ranges,
which is what the column accumulators' 64-bit head-room was sized for.  The GPU parity tests (test_gpu_msm.py) cover the kernels
around this arithmetic; this file keeps the arithmetic itself under test where no GPU exists."""
import ctypes as C
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEOM = {
    0: dict(name="bn254", NL=9, B=29, N=8, b=3, g=(1, 2),
            p=21888242871839275222246405745257275088696311157297823662689037894645226208583),
    1: dict(name="bls12_381", NL=14, B=28, N=12, b=4,
            g=(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
               0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
            p=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab),
}
# ec_lazy.hpp's accumulator invariants, in units of p
BOUND = dict(x=5.2, y=3.3, zz=2.0, zzz=2.0)


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("ecl") / "ec_lazy_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas",
                           os.path.join(ROOT, "tests", "host_cpp", "ec_lazy_host.cpp"), "-o", so])
    lib_ = C.CDLL(so)
    lib_.ecl_curve_op.restype = C.c_int
    return lib_


class Field:
    def __init__(self, curve):
        g = GEOM[curve]
        self.curve, self.NL, self.B, self.N, self.p = curve, g["NL"], g["B"], g["N"], g["p"]
        self.Rp = 1 << (self.B * self.NL)              # R'
        self.R = 1 << (32 * self.N)                    # the reference's Montgomery radix
        self.mask = (1 << self.B) - 1

    def limbs(self, v):
        """normalised limbs: B bits each, the excess in the top limb"""
        out = [(v >> (self.B * k)) & self.mask for k in range(self.NL - 1)]
        top = v >> (self.B * (self.NL - 1))
        assert top < (1 << 32), "value does not fit the limb form"
        return out + [top]

    def value(self, l):
        return sum(int(x) << (self.B * k) for k, x in enumerate(l))

    def normalised(self, l):
        return all(int(x) <= self.mask for x in l[:-1])

    def enc(self, x, lift=0):
        """x (plain residue) -> limbs of x*R' mod p + lift*p"""
        return self.limbs(x * self.Rp % self.p + lift * self.p)

    def dec(self, l):
        return self.value(l) * pow(self.Rp, -1, self.p) % self.p

    def arr(self, rows):
        flat = [w for r in rows for w in r]
        return (C.c_uint32 * len(flat))(*flat)


# ------------------------------------------------------------------------------------------------ affine integers
def ec_add(F, P, Q):
    p = F.p
    if P is None:
        return Q
    if Q is None:
        return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def ec_neg(F, P):
    return None if P is None else (P[0], (-P[1]) % F.p)


def ec_mul(F, k, P):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(F, acc, P)
        P = ec_add(F, P, P)
        k >>= 1
    return acc


def points(F, rng, count):
    g = GEOM[F.curve]
    assert (g["g"][1] ** 2 - g["g"][0] ** 3 - g["b"]) % F.p == 0
    return [ec_mul(F, rng.randrange(1, 1 << 64), g["g"]) for _ in range(count)]


# ------------------------------------------------------------------------------------------------ limb-form points
def aff_limbs(F, P, lift_y=0):
    if P is None:
        return [0] * (2 * F.NL)
    return F.enc(P[0]) + F.enc(P[1], lift_y)


def acc_limbs(F, P, rng=None, lift=None):
    """an XYZZ accumulator of the group element P with a random Z (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2); `lift` = (kx, ky, kzz, kzzz)
    multiples of p added to the canonical residues"""
    if P is None:
        return [0] * (4 * F.NL)
    z = rng.randrange(1, F.p) if rng else 1
    zz, zzz = z * z % F.p, z * z * z % F.p
    kx, ky, kzz, kzzz = lift or (0, 0, 0, 0)
    return F.enc(P[0] * zz % F.p, kx) + F.enc(P[1] * zzz % F.p, ky) + F.enc(zz, kzz) + F.enc(zzz, kzzz)


def acc_point(F, l, where=""):
    """checks the invariants of ec_lazy.hpp on an accumulator and returns its group element"""
    NL, p = F.NL, F.p
    X, Y, ZZ, ZZZ = (l[i * NL:(i + 1) * NL] for i in range(4))
    if all(int(w) == 0 for w in ZZ):
        return None
    for name, c in (("x", X), ("y", Y), ("zz", ZZ), ("zzz", ZZZ)):
        assert F.normalised(c), (where, name, "limbs not normalised")
        assert F.value(c) < BOUND[name] * p, (where, name, F.value(c) / p)
    zz, zzz = F.dec(ZZ), F.dec(ZZZ)
    assert zz % p != 0 and pow(zz, 3, p) == zzz * zzz % p, (where, "ZZ^3 != ZZZ^2")
    return F.dec(X) * pow(zz, -1, p) % p, F.dec(Y) * pow(zzz, -1, p) % p


def run(lib, F, op, a, b):
    out = (C.c_uint32 * (4 * F.NL))()
    ok = lib.ecl_curve_op(F.curve, op, F.arr([a]), F.arr([b]), out)
    assert ok >= 0
    return ok, list(out)


MADD_FAST, MADD_FUSED, MADD, ADD, ADD_FAST, DBL, DBL_AFF, NEG = range(8)


# ------------------------------------------------------------------------------------------------ tests
@pytest.mark.parametrize("curve", [0, 1])
def test_limb_parameters(lib, curve):
    F = Field(curve)
    NL, p = F.NL, F.p
    raw = (C.c_uint32 * (8 * NL + 1))()
    lib.ecl_params(curve, raw)
    row = lambda i: [int(x) for x in raw[i * NL:(i + 1) * NL]]
    pl, p2, c2, c4, c8, one, r_std, r2fix = (row(i) for i in range(8))
    assert F.value(pl) == p and F.normalised(pl) and F.value(p2) == 2 * p and F.normalised(p2)
    for k, c in ((2, c2), (4, c4), (8, c8)):
        assert F.value(c) == k * p                      # the lift moves 2^31 into every limb without changing the value
        # a + C - b must not underflow in any limb for b limbs < 3 * 2^B (PPP + 2Q is the widest subtrahend of ec_lazy.hpp)
        assert all(x >= 3 << F.B for x in c[:-1]) and c[-1] < (1 << 31)
    assert F.value(one) == F.Rp % p and F.value(r_std) == F.R % p
    assert F.value(r2fix) == F.Rp * F.Rp * pow(F.R, -1, p) % p
    assert (int(raw[8 * NL]) * pl[0] + 1) % (1 << F.B) == 0          # inv = -p^-1 mod 2^B


@pytest.mark.parametrize("curve", [0, 1])
def test_field_products_up_to_the_lazy_bounds(lib, curve):
    F = Field(curve)
    p, NL = F.p, F.NL
    rng = random.Random(0xF1 + curve)
    n = 4000
    Rinv = pow(F.Rp, -1, p)
    # operand ranges of the curve formulas: P < 9.2p, T < 9.1p, R < 5.2p, accumulator X < 5.2p, (4p - Y) < 4p + ...
    top = int(9.2 * p)
    edge = [0, 1, p - 1, p, p + 1, 2 * p, top - 1, int(5.2 * p) - 1]

    def operands(k, bound):
        return [edge[(i // (len(edge) ** k)) % len(edge)] % bound if i < len(edge) ** 2 else rng.randrange(bound) for i in range(n)]

    a, b = operands(0, top), operands(1, top)
    out = (C.c_uint32 * (NL * n))()
    lib.ecl_field_op(curve, 0, F.arr([F.limbs(v) for v in a]), F.arr([F.limbs(v) for v in b]), None, None, out, C.c_long(n))
    for k in range(n):
        r = out[NL * k:NL * k + NL]
        v = F.value(r)
        assert F.normalised(r) and v % p == a[k] * b[k] * Rinv % p and v < a[k] * b[k] // F.Rp + p + 1, ("mul", k)
    lib.ecl_field_op(curve, 1, F.arr([F.limbs(v) for v in a]), None, None, None, out, C.c_long(n))
    for k in range(n):
        r = out[NL * k:NL * k + NL]
        v = F.value(r)
        assert F.normalised(r) and v % p == a[k] * a[k] * Rinv % p and v < a[k] * a[k] // F.Rp + p + 1, ("sqr", k)
    # Y3 = R*T + (4p - Y1)*PPP under one reduction: R < 5.2p, T < 9.1p, 4p - Y1 < 4p (lifted: up to 4p + 2^31 in the limbs' slack), PPP < 1.1p
    r_, t_ = [rng.randrange(int(5.2 * p)) for _ in range(n)], [rng.randrange(int(9.1 * p)) for _ in range(n)]
    ny, pp_ = [rng.randrange(4 * p + 1) for _ in range(n)], [rng.randrange(int(1.1 * p)) for _ in range(n)]
    r_[0], t_[0], ny[0], pp_[0] = int(5.2 * p) - 1, int(9.1 * p) - 1, 4 * p, int(1.1 * p) - 1
    lib.ecl_field_op(curve, 2, F.arr([F.limbs(v) for v in r_]), F.arr([F.limbs(v) for v in t_]), F.arr([F.limbs(v) for v in ny]),
                     F.arr([F.limbs(v) for v in pp_]), out, C.c_long(n))
    for k in range(n):
        r = out[NL * k:NL * k + NL]
        v = F.value(r)
        s = r_[k] * t_[k] + ny[k] * pp_[k]
        assert F.normalised(r) and v % p == s * Rinv % p and v < s // F.Rp + p + 1, ("dot2", k)
        assert v < 1.4 * p                                                            # the bound xyzzl_madd_fast<FUSED_Y3> relies on


@pytest.mark.parametrize("curve", [0, 1])
def test_standard_form_round_trip(lib, curve):
    """bases_to_limbs_kernel / store_std: the reference's R = 2^(32N) Montgomery residues <-> the resident R' limb form"""
    F = Field(curve)
    rng = random.Random(5 + curve)
    for x in [0, 1, F.p - 1] + [rng.randrange(F.p) for _ in range(300)]:
        std = x * F.R % F.p
        s = (C.c_uint32 * F.N)(*[(std >> (32 * i)) & 0xffffffff for i in range(F.N)])
        l = (C.c_uint32 * F.NL)()
        lib.ecl_from_std(curve, s, l)
        assert F.value(l) == x * F.Rp % F.p and F.normalised(l)
        for lift in (0, 1):                                                            # store_std takes lazy values
            back = (C.c_uint32 * F.N)()
            lib.ecl_to_std(curve, F.arr([F.limbs(F.value(l) + lift * F.p)]), back)
            assert sum(int(w) << (32 * i) for i, w in enumerate(back)) == std


@pytest.mark.parametrize("curve", [0, 1])
def test_bucket_accumulation_chain(lib, curve):
    """What one lane of msm_accumulate_kernel does: a run of mixed additions of +/- bases into an XYZZ accumulator, both Y3
    formulations, invariants after every step; the same-x cases abort untouched and are finished by the complete formula."""
    F = Field(curve)
    rng = random.Random(0xACC + curve)
    pts = points(F, rng, 24)
    for fused in (MADD_FAST, MADD_FUSED):
        acc, want = [0] * (4 * F.NL), None
        for step in range(160):
            P = pts[rng.randrange(len(pts))]
            q = aff_limbs(F, P)
            if rng.random() < 0.5:                                                     # a negative digit: y -> 2p - y, in (p, 2p]
                _, qn = run(lib, F, NEG, acc, q)
                q = qn[:2 * F.NL]
                assert F.value(q[F.NL:]) == 2 * F.p - P[1] * F.Rp % F.p and F.normalised(q[F.NL:])
                P = ec_neg(F, P)
            ok, new = run(lib, F, fused, acc, q)
            if want is not None and want[0] == P[0]:                                   # P + P or P + (-P): the fast path must refuse
                assert ok == 0 and new == acc, "same-x addition must leave the accumulator untouched"
                ok, new = run(lib, F, MADD, acc, q)                                    # msm_accumulate_redo_kernel
            assert ok == 1
            want = ec_add(F, want, P)
            acc = new
            assert acc_point(F, acc, (fused, step)) == want
        assert want is not None


@pytest.mark.parametrize("curve", [0, 1])
def test_exceptional_cases(lib, curve):
    F = Field(curve)
    rng = random.Random(0xE + curve)
    P, Q = points(F, rng, 2)
    inf_acc, inf_aff = [0] * (4 * F.NL), [0] * (2 * F.NL)
    a = acc_limbs(F, P, rng)
    # mixed: acc + infinity base (complete path only; the kernel filters infinity for the fast one), infinity acc + base
    assert acc_point(F, run(lib, F, MADD, a, inf_aff)[1]) == P
    for op in (MADD_FAST, MADD_FUSED, MADD):
        ok, o = run(lib, F, op, inf_acc, aff_limbs(F, Q))
        assert ok == 1 and acc_point(F, o) == Q
    # mixed doubling and cancellation
    for op in (MADD_FAST, MADD_FUSED):
        assert run(lib, F, op, a, aff_limbs(F, P))[0] == 0
        assert run(lib, F, op, a, aff_limbs(F, ec_neg(F, P)))[0] == 0
    assert acc_point(F, run(lib, F, MADD, a, aff_limbs(F, P))[1]) == ec_add(F, P, P)
    assert acc_point(F, run(lib, F, MADD, a, aff_limbs(F, ec_neg(F, P)))[1]) is None
    assert acc_point(F, run(lib, F, DBL_AFF, inf_acc, aff_limbs(F, P))[1]) == ec_add(F, P, P)
    # accumulator + accumulator (the pyramid, the heavy-bucket tree, msm_points_sum_kernel)
    b = acc_limbs(F, Q, rng)
    same, opp = acc_limbs(F, P, rng), acc_limbs(F, ec_neg(F, P), rng)           # other Z: a different representation of the same x
    assert acc_point(F, run(lib, F, ADD, a, b)[1]) == ec_add(F, P, Q)
    ok, o = run(lib, F, ADD_FAST, a, b)
    assert ok == 1 and acc_point(F, o) == ec_add(F, P, Q)
    assert acc_point(F, run(lib, F, ADD, a, same)[1]) == ec_add(F, P, P)
    assert acc_point(F, run(lib, F, ADD, a, opp)[1]) is None
    for other in (same, opp):
        ok, o = run(lib, F, ADD_FAST, a, other)
        assert ok == 0 and o == a
    assert acc_point(F, run(lib, F, ADD, a, inf_acc)[1]) == P and acc_point(F, run(lib, F, ADD, inf_acc, b)[1]) == Q
    for x, y, w in ((a, inf_acc, P), (inf_acc, b, Q), (inf_acc, inf_acc, None)):
        ok, o = run(lib, F, ADD_FAST, x, y)
        assert ok == 1 and acc_point(F, o) == w
    assert acc_point(F, run(lib, F, DBL, a, a)[1]) == ec_add(F, P, P)
    assert acc_point(F, run(lib, F, DBL, inf_acc, inf_acc)[1]) is None


@pytest.mark.parametrize("curve", [0, 1])
def test_operands_at_the_top_of_their_ranges(lib, curve):
    """The bound bookkeeping of ec_lazy.hpp: accumulators whose coordinates sit just under the stated invariants (X < 5.2p, Y < 3.3p,
    ZZ, ZZZ < 2p — canonical residue plus the largest multiple of p that fits) and negated bases (y in (p, 2p]) still give the right
    group element and results INSIDE the invariants: no column accumulator wrapped, no lifted subtraction underflowed."""
    F = Field(curve)
    rng = random.Random(0xB0 + curve)
    pts = points(F, rng, 12)
    lifts = [(4, 2, 1, 1), (4, 0, 0, 0), (0, 2, 0, 0), (0, 0, 1, 1), (3, 1, 1, 0)]
    for trial in range(40):
        P, Q = rng.sample(pts, 2)
        la, lb = lifts[trial % len(lifts)], lifts[(trial // len(lifts)) % len(lifts)]
        a, b = acc_limbs(F, P, rng, la), acc_limbs(F, Q, rng, lb)
        assert acc_point(F, a, "lifted a") == P and acc_point(F, b, "lifted b") == Q
        q = aff_limbs(F, Q)
        if trial & 1:
            q = run(lib, F, NEG, a, q)[1][:2 * F.NL]
        Qs = ec_neg(F, Q) if trial & 1 else Q
        for op in (MADD_FAST, MADD_FUSED, MADD):
            ok, o = run(lib, F, op, a, q)
            assert ok == 1 and acc_point(F, o, (op, trial)) == ec_add(F, P, Qs)
        for op in (ADD, ADD_FAST):
            ok, o = run(lib, F, op, a, b)
            assert ok == 1 and acc_point(F, o, (op, trial)) == ec_add(F, P, Q)
        assert acc_point(F, run(lib, F, DBL, a, a)[1], ("dbl", trial)) == ec_add(F, P, P)
        assert acc_point(F, run(lib, F, MADD, a, aff_limbs(F, P, lift_y=0))[1], ("madd dbl", trial)) == ec_add(F, P, P)
        same = acc_limbs(F, P, rng, lb)
        assert acc_point(F, run(lib, F, ADD, a, same)[1], ("add dbl", trial)) == ec_add(F, P, P)


@pytest.mark.parametrize("curve", [0, 1])
def test_reduction_pyramid_chunk(lib, curve):
    """One lane of msm_reduce_level_kernel: over a strided chunk E_0 .. E_(K-1) it emits acc = sum_t t * E_t and S = sum_t E_t
    with 2K additions; the identity  sum_j (j+1) E_j = sum_ch (ch+1) S_ch + nch * sum_ch acc_ch  the host's Horner relies on is
    checked on a small window of buckets."""
    F = Field(curve)
    rng = random.Random(0x9e + curve)
    K, nch = 4, 3
    nb = K * nch
    pts = points(F, rng, nb)
    pts[5] = None                                                                      # an empty bucket
    E = [acc_limbs(F, P, rng) for P in pts]
    inf = [0] * (4 * F.NL)
    S, A = [], []
    for ch in range(nch):
        running, acc = inf, inf
        for d in reversed(range(K)):
            ok, acc = run(lib, F, ADD_FAST, acc, running)
            assert ok == 1
            ok, running = run(lib, F, ADD_FAST, running, E[ch + d * nch])
            assert ok == 1
        S.append(acc_point(F, running, ("S", ch)))
        A.append(acc_point(F, acc, ("A", ch)))
        want_s, want_a = None, None
        for t in range(K):
            want_s = ec_add(F, want_s, pts[ch + t * nch])
            want_a = ec_add(F, want_a, ec_mul(F, t, pts[ch + t * nch]) if pts[ch + t * nch] else None)
        assert S[ch] == want_s and A[ch] == want_a
    lhs = None
    for j, P in enumerate(pts):
        lhs = ec_add(F, lhs, ec_mul(F, j + 1, P) if P else None)
    rhs = None
    for ch in range(nch):
        rhs = ec_add(F, rhs, ec_mul(F, ch + 1, S[ch]) if S[ch] else None)
        rhs = ec_add(F, rhs, ec_mul(F, nch, A[ch]) if A[ch] else None)
    assert lhs == rhs
