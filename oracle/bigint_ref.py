"""Pure-Python big-integer statement of the hot path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the reference holds no golden vectors / KATs for this path,
and its Rust sources cannot be built here.  This file restates, with Python ints, the
*published* algorithms of the un-vendored dependencies the reference calls
(ark-ff / ark-poly / ark-ec 0.3.0, Cargo.lock:99-102,149-152,190-193) and the reference's own
in-tree orchestration.  It is used only by tests/ to pin the faster C oracle
(oracle/plonk_oracle.c) and to generate tests/golden/*.json — never by the product path.

Reference call sites restated here (all under /root/reference/src):
  * playground.rs:21-80        4-step fft / ifft / coset_fft / coset_ifft
  * worker.rs:66-94            fft1_helper   (row pass)
  * worker.rs:96-115           fft2_helper   (column pass)
  * worker.rs:327-330,432-435  fft2_prepare pack / fft_exchange scatter-transpose
  * dispatcher2.rs:732-787     Prover::fft orchestration
  * dispatcher.rs:218-240      sharded MSM + reduce
  * worker.rs:117-123          commit_polynomial
  * dispatcher2.rs:362-504     quotient polynomial coset evaluations
"""
from __future__ import annotations

import random
from dataclasses import dataclass


# --------------------------------------------------------------------------- fields
@dataclass(frozen=True)
class Field:
    name: str
    p: int
    limbs64: int          # ark-ff BigInteger width (u64 limbs)
    generator: int        # multiplicative generator (coset shift), 0 for base fields
    two_adicity: int

    @property
    def R(self) -> int:   # Montgomery radix (ark-ff 0.3.0: R = 2^(64*limbs))
        return pow(2, 64 * self.limbs64, self.p)

    @property
    def bits(self) -> int:
        return self.p.bit_length()

    def to_mont(self, a: int) -> int:
        return a * self.R % self.p

    def from_mont(self, a: int) -> int:
        return a * pow(self.R, -1, self.p) % self.p

    def root_of_unity(self, n: int) -> int:
        """ark-ff FftField::get_root_of_unity(n): TWO_ADIC_ROOT squared down to order n."""
        log = n.bit_length() - 1
        assert 1 << log == n and log <= self.two_adicity
        w = pow(self.generator, (self.p - 1) >> self.two_adicity, self.p)
        for _ in range(self.two_adicity - log):
            w = w * w % self.p
        return w


BN254_FR = Field("bn254_fr",
                 21888242871839275222246405745257275088548364400416034343698204186575808495617,
                 4, 5, 28)
BN254_FQ = Field("bn254_fq",
                 21888242871839275222246405745257275088696311157297823662689037894645226208583,
                 4, 0, 1)
BLS12_381_FR = Field("bls12_381_fr",
                     0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
                     4, 7, 32)
BLS12_381_FQ = Field("bls12_381_fq",
                     0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
                     6, 0, 1)


@dataclass(frozen=True)
class Curve:
    name: str
    fr: Field
    fq: Field
    b: int                # y^2 = x^3 + b
    gx: int
    gy: int


BN254 = Curve("bn254", BN254_FR, BN254_FQ, 3, 1, 2)
BLS12_381 = Curve(
    "bls12_381", BLS12_381_FR, BLS12_381_FQ, 4,
    0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
    0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)

CURVES = {"bn254": BN254, "bls12_381": BLS12_381}


# --------------------------------------------------------------------------- radix-2 domain
class Radix2Domain:
    """ark-poly 0.3.0 Radix2EvaluationDomain (values are plain residues, not Montgomery)."""

    def __init__(self, f: Field, num_coeffs: int):
        size = 1
        while size < num_coeffs:
            size <<= 1
        self.f, self.size = f, size
        self.log_size = size.bit_length() - 1
        if self.log_size > f.two_adicity:
            raise ValueError("DomainCreationError: log size > TWO_ADICITY")
        self.group_gen = f.root_of_unity(size)
        self.group_gen_inv = pow(self.group_gen, -1, f.p)
        self.size_inv = pow(size, -1, f.p)
        self.generator_inv = pow(f.generator, -1, f.p)

    # in-order radix-2; any exact algorithm yields the same residues
    def _ntt(self, v, w):
        p, n = self.f.p, self.size
        v = list(v) + [0] * (n - len(v))
        assert len(v) == n
        # bit reversal
        j = 0
        for i in range(1, n):
            bit = n >> 1
            while j & bit:
                j ^= bit
                bit >>= 1
            j |= bit
            if i < j:
                v[i], v[j] = v[j], v[i]
        length = 2
        while length <= n:
            wl = pow(w, n // length, p)
            for s in range(0, n, length):
                t = 1
                for k in range(length // 2):
                    a, b = v[s + k], v[s + k + length // 2] * t % p
                    v[s + k] = (a + b) % p
                    v[s + k + length // 2] = (a - b) % p
                    t = t * wl % p
            length <<= 1
        return v

    def fft(self, v):
        return self._ntt(v, self.group_gen)

    def ifft(self, v):
        p = self.f.p
        return [x * self.size_inv % p for x in self._ntt(v, self.group_gen_inv)]

    def coset_fft(self, v):
        return self.fft(distribute_powers(self.f, v, self.f.generator))

    def coset_ifft(self, v):
        return distribute_powers(self.f, self.ifft(v), self.generator_inv)


def distribute_powers(f: Field, v, g: int):
    out, t = [], 1
    for x in v:
        out.append(x * t % f.p)
        t = t * g % f.p
    return out


def naive_dft(f: Field, v, w):
    n = len(v)
    return [sum(v[j] * pow(w, j * k, f.p) for j in range(n)) % f.p for k in range(n)]


# --------------------------------------------------------------------------- 4-step (playground.rs)
def _transpose(rows):
    return [list(col) for col in zip(*rows)]


def fourstep(f: Field, size: int, coeffs, is_inv: bool, is_coset: bool):
    """playground.rs:21-80 (fft/ifft/coset_fft/coset_ifft) as one function."""
    dom = Radix2Domain(f, size)
    r = 1 << (dom.log_size >> 1)
    c = dom.size // r
    r_dom, c_dom = Radix2Domain(f, r), Radix2Domain(f, c)
    v = list(coeffs) + [0] * (dom.size - len(coeffs))
    if is_coset and not is_inv:
        v = distribute_powers(f, v, f.generator)                      # playground.rs:69
    t = _transpose([v[k:k + r] for k in range(0, dom.size, r)])       # :29 / :52
    w = dom.group_gen_inv if is_inv else dom.group_gen
    for i, group in enumerate(t):
        group[:] = c_dom.ifft(group) if is_inv else c_dom.fft(group)  # :31 / :54
        for j in range(len(group)):
            group[j] = group[j] * pow(w, i * j, f.p) % f.p             # :35 / :58
    groups = _transpose(t)
    groups = [r_dom.ifft(g) if is_inv else r_dom.fft(g) for g in groups]  # :40 / :63
    out = [x for row in _transpose(groups) for x in row]              # :41 / :64
    if is_coset and is_inv:
        out = distribute_powers(f, out, dom.generator_inv)            # :74-78
    return out


# --------------------------------------------------------------------------- distributed (worker.rs)
@dataclass
class FftWorkload:            # utils.rs:3-19
    row_start: int
    row_end: int
    col_start: int
    col_end: int


def make_workloads(r: int, c: int, S: int):
    """dispatcher2.rs:272-291 / dispatcher.rs:278-285."""
    return [FftWorkload(i * r // S, (i + 1) * r // S, i * c // S, (i + 1) * c // S) for i in range(S)]


def fft1_helper(f, v, i, is_coset, is_inv, dom, c_dom, r_dom):
    """worker.rs:66-94."""
    p = f.p
    v = list(v)
    if is_coset and not is_inv:
        v = [u * pow(f.generator, i + j * r_dom.size, p) % p for j, u in enumerate(v)]
    v = c_dom.ifft(v) if is_inv else c_dom.fft(v)
    w = dom.group_gen_inv if is_inv else dom.group_gen
    return [u * pow(w, i * j, p) % p for j, u in enumerate(v)]


def fft2_helper(f, v, i, is_coset, is_inv, c_dom, r_dom):
    """worker.rs:96-115."""
    p = f.p
    v = r_dom.ifft(v) if is_inv else r_dom.fft(v)
    if is_coset and is_inv:
        ginv = pow(f.generator, -1, p)
        v = [u * pow(ginv, i + j * c_dom.size, p) % p for j, u in enumerate(v)]
    return v


def distributed_fft(f: Field, size: int, coeffs, S: int, is_inv: bool, is_coset: bool):
    """dispatcher2.rs:732-787 with S in-process workers (worker.rs:187-381,412-438)."""
    dom = Radix2Domain(f, size)
    r = 1 << (dom.log_size >> 1)
    c = dom.size // r
    r_dom, c_dom = Radix2Domain(f, r), Radix2Domain(f, c)
    wl = make_workloads(r, c, S)
    v = list(coeffs) + [0] * (dom.size - len(coeffs))
    t = _transpose([v[k:k + r] for k in range(0, dom.size, r)])     # dispatcher2.rs:754
    # fft1 on each worker (one call per row)
    rows = []
    for s in range(S):
        mine = []
        for j in range(wl[s].row_end - wl[s].row_start):
            gi = j + wl[s].row_start                                 # worker.rs:267
            mine.append(fft1_helper(f, t[gi], gi, is_coset, is_inv, dom, c_dom, r_dom))
        rows.append(mine)
    # fft2_prepare + fft_exchange
    cols = [[[0] * r for _ in range(wl[s].col_end - wl[s].col_start)] for s in range(S)]
    for src in range(S):
        for dst in range(S):
            blk = [x for row in rows[src] for x in row[wl[dst].col_start:wl[dst].col_end]]  # :327-330
            nc = wl[dst].col_end - wl[dst].col_start
            for i, x in enumerate(blk):                              # :432-435
                cols[dst][i % nc][wl[src].row_start + i // nc] = x
    # fft2
    u = [None] * c
    for s in range(S):
        for i, col in enumerate(cols[s]):
            gi = i + wl[s].col_start                                 # worker.rs:369
            u[gi] = fft2_helper(f, col, gi, is_coset, is_inv, c_dom, r_dom)
    return [x for row in _transpose(u) for x in row]                 # dispatcher2.rs:786


# --------------------------------------------------------------------------- curve (short Weierstrass, a = 0)
INF = None  # affine point at infinity


def on_curve(cv: Curve, P):
    if P is INF:
        return True
    x, y = P
    return (y * y - x * x * x - cv.b) % cv.fq.p == 0


def affine_add(cv: Curve, P, Q):
    p = cv.fq.p
    if P is INF:
        return Q
    if Q is INF:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return (x3, (lam * (x1 - x3) - y1) % p)


def affine_neg(cv, P):
    return INF if P is INF else (P[0], (-P[1]) % cv.fq.p)


def scalar_mul(cv: Curve, k: int, P):
    acc, base = INF, P
    while k:
        if k & 1:
            acc = affine_add(cv, acc, base)
        base = affine_add(cv, base, base)
        k >>= 1
    return acc


def msm_naive(cv: Curve, bases, scalars):
    acc = INF
    for P, s in zip(bases, scalars):
        acc = affine_add(cv, acc, scalar_mul(cv, s, P))
    return acc


def ln_without_floats(a: int) -> int:
    """ark-std: log2_ceil-ish * 69 / 100 (used for the Pippenger window)."""
    log2 = (a - 1).bit_length() if a > 1 else 0   # ark_std::log2 = ceil(log2(a))
    return log2 * 69 // 100


def ark_window(size: int) -> int:
    return 3 if size < 32 else ln_without_floats(size) + 2


def msm_pippenger(cv: Curve, bases, scalars):
    """ark-ec 0.3.0 VariableBaseMSM::multi_scalar_mul, affine bookkeeping."""
    size = min(len(bases), len(scalars))
    pairs = [(bases[i], scalars[i]) for i in range(size) if scalars[i] != 0]
    c = ark_window(size)
    num_bits = cv.fr.bits
    window_sums = []
    for w_start in range(0, num_bits, c):
        res = INF
        buckets = [INF] * ((1 << c) - 1)
        for P, s in pairs:
            if s == 1:
                if w_start == 0:
                    res = affine_add(cv, res, P)
            else:
                d = (s >> w_start) % (1 << c)
                if d:
                    buckets[d - 1] = affine_add(cv, buckets[d - 1], P)
        running = INF
        for b in reversed(buckets):
            running = affine_add(cv, running, b)
            res = affine_add(cv, res, running)
        window_sums.append(res)
    lowest = window_sums[0]
    total = INF
    for ws in reversed(window_sums[1:]):
        total = affine_add(cv, total, ws)
        for _ in range(c):
            total = affine_add(cv, total, total)
    return affine_add(cv, lowest, total)


def sharded_msm(cv: Curve, bases, scalars, S: int):
    """dispatcher.rs:218-238: contiguous index shards + reduce(a+b)."""
    n = len(scalars)
    acc = INF
    for i in range(S):
        lo, hi = i * n // S, (i + 1) * n // S
        acc = affine_add(cv, acc, msm_pippenger(cv, bases[lo:hi], scalars[lo:hi]))
    return acc


def commit_polynomial(cv: Curve, bases, coeffs_mont):
    """worker.rs:117-123: into_repr every coeff, zero-pad to bases.len(), MSM."""
    sc = [cv.fr.from_mont(x) for x in coeffs_mont]
    sc += [0] * (len(bases) - len(sc))
    return msm_pippenger(cv, bases, sc)


# --------------------------------------------------------------------------- limb (de)serialisation (utils.rs:27-43)
def to_limbs(x: int, n64: int):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n64)]


def from_limbs(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def rand_points(cv: Curve, n: int, rng: random.Random):
    G = (cv.gx, cv.gy)
    return [scalar_mul(cv, rng.randrange(1, cv.fr.p), G) for _ in range(n)]


def quotient_evals(f: Field, n: int, sel, sig, wire, z, pi, alpha, beta, gamma, k):
    """dispatcher2.rs:362-504 on plain residues.  sel[13][m], sig[5][m], wire[5][m], z[m], pi[m]."""
    p = f.p
    m = 8 * n
    dom = Radix2Domain(f, m)
    a2n = alpha * alpha * pow(n, -1, p) % p
    xs = [f.generator * pow(dom.group_gen, i, p) % p for i in range(m)]
    ratio = m // n
    zh_inv = [pow(pow(xs[i], n, p) - 1, -1, p) for i in range(ratio)]
    out = []
    for i in range(m):
        x = xs[i]
        a, b, c, d, e = (wire[j][i] for j in range(5))
        ab, cd = a * b % p, c * d % p
        gate = (sel[11][i] + pi[i] + sel[0][i] * a + sel[1][i] * b + sel[2][i] * c + sel[3][i] * d + sel[4][i] * ab + sel[5][i] * cd
                + sel[12][i] * ab * cd * e + sel[6][i] * pow(a, 5, p) + sel[7][i] * pow(b, 5, p) + sel[8][i] * pow(c, 5, p)
                + sel[9][i] * pow(d, 5, p) - sel[10][i] * e) % p
        acc1, acc2 = z[i], z[(i + ratio) % m]
        for j in range(5):
            t = (wire[j][i] + gamma) % p
            acc1 = acc1 * (t + k[j] * x * beta) % p
            acc2 = acc2 * (t + sig[j][i] * beta) % p
        perm = alpha * (acc1 - acc2) % p
        l1 = a2n * (z[i] - 1) * pow(x - 1, -1, p) % p
        out.append((zh_inv[i % ratio] * (gate + perm) + l1) % p)
    return out


# --------------------------------------------------------------------------- permutation grand product
def perm_product(f: Field, n: int, wires, id_perm, perm_idx, beta: int, gamma: int):
    """dispatcher2.rs:329-344 on plain residues.  wires[5][n] = witness[wire_variables[i][j]],
    id_perm[5n] = extended_id_permutation, perm_idx[5n] = perm_i*n + perm_j of wire_permutation[i*n+j].
    Returns product_vec (n evaluations, product_vec[0] = 1).  A zero denominator panics in the reference
    (Fp Div unwraps the inverse); here it raises ZeroDivisionError/ValueError from pow()."""
    p = f.p
    out = [1]
    for j in range(n - 1):
        a = b = 1
        for i in range(len(wires)):
            t = (wires[i][j] + gamma) % p
            a = a * (t + beta * id_perm[i * n + j]) % p
            b = b * (t + beta * id_perm[perm_idx[i * n + j]]) % p
        out.append(out[j] * a % p * pow(b, -1, p) % p)
    return out


# --------------------------------------------------------------------------- round 4/5 polynomial ops
def poly_eval(f: Field, coeffs, z: int) -> int:
    """DensePolynomial::evaluate (Horner), dispatcher2.rs:545-555."""
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * z + c) % f.p
    return acc


def poly_lincomb(f: Field, polys, coeffs):
    """sum_k coeffs[k] * polys[k]  (DensePolynomial Mul<Fr> / Add), dispatcher2.rs:566-633,646-649."""
    out = [0] * max((len(q) for q in polys), default=0)
    for q, c in zip(polys, coeffs):
        for i, v in enumerate(q):
            out[i] = (out[i] + c * v) % f.p
    return out


def poly_div_linear(f: Field, coeffs, z: int):
    """The synthetic-division loop of dispatcher2.rs:651-666 / :672-688: quotient of poly / (X - z), remainder dropped.
    Restated literally (remainder vector, pop of trailing zeros) so the trimmed-degree behaviour is the reference's."""
    p = f.p
    rem = list(coeffs)
    while rem and rem[-1] == 0:
        rem.pop()
    if len(rem) <= 1:
        return []
    quot = [0] * (len(rem) - 1)
    while rem and len(rem) - 1 >= 1:
        cq = rem[-1]
        d = len(rem) - 2
        quot[d] = cq
        rem[d] = (rem[d] + cq * z) % p
        rem[d + 1] = (rem[d + 1] - cq) % p
        while rem and rem[-1] == 0:
            rem.pop()
    return quot


def blind(f: Field, coeffs, n: int, blinders):
    """DensePolynomial::rand(k-1).mul_by_vanishing_poly(domain) + poly  (dispatcher2.rs:311-312,347-348):
    (sum b_i X^i)(X^n - 1) + poly."""
    out = list(coeffs) + [0] * (n + len(blinders) - len(coeffs))
    for i, b in enumerate(blinders):
        out[i] = (out[i] - b) % f.p
        out[n + i] = (out[n + i] + b) % f.p
    return out


def prove_rounds(cv: Curve, n: int, ck, circuit: dict, blinders: dict, ch: dict):
    """Rounds 1-5 of dispatcher2.rs::Prover::prove (:296-712) on plain residues, with the transcript challenges
    (beta, gamma, alpha, zeta, v) and the blinding polynomials supplied by the caller.
    circuit: wires[5][n] evaluations, selectors[13] / sigmas[5] coefficient vectors (ProvingKey polys), id_perm[5n],
    perm_idx[5n], pub_input[n] evaluations, k[5].  ck: affine points (None = infinity), already padded (:207-208).
    Returns a dict with every commitment (affine) and evaluation of `Proof` plus intermediate polynomials."""
    f = cv.fr
    p = f.p
    dom, qdom = Radix2Domain(f, n), Radix2Domain(f, 6 * n + 7)
    m = qdom.size
    assert m == 8 * n

    def commit(coeffs):
        return commit_polynomial(cv, ck, [f.to_mont(c) for c in coeffs])

    # Round 1 (:296-322)
    wire_polys = [blind(f, dom.ifft(circuit["wires"][i]), n, blinders["wires"][i]) for i in range(5)]
    wires_poly_comms = [commit(q) for q in wire_polys]
    # Round 2 (:325-357)
    beta, gamma = ch["beta"], ch["gamma"]
    prod = perm_product(f, n, circuit["wires"], circuit["id_perm"], circuit["perm_idx"], beta, gamma)
    perm_poly = blind(f, dom.ifft(prod), n, blinders["perm"])
    prod_perm_poly_comm = commit(perm_poly)
    # Round 3 (:360-533)
    alpha = ch["alpha"]

    def cfft(coeffs):
        return qdom.coset_fft(list(coeffs) + [0] * (m - len(coeffs)))

    pi_poly = dom.ifft(circuit["pub_input"])
    evals = quotient_evals(f, n, [cfft(q) for q in circuit["selectors"]], [cfft(q) for q in circuit["sigmas"]],
                           [cfft(q) for q in wire_polys], cfft(perm_poly), cfft(pi_poly), alpha, beta, gamma, circuit["k"])
    quot = qdom.coset_ifft(evals)
    while quot and quot[-1] == 0:
        quot.pop()
    expected_degree = 5 * (n + 1) + 2
    if len(quot) - 1 != expected_degree:
        raise ValueError(f"WrongQuotientPolyDegree({len(quot) - 1}, {expected_degree})")
    split = [quot[i:i + n + 2] for i in range(0, len(quot), n + 2)]
    split_quot_poly_comms = [commit(q) for q in split]
    # Round 4 (:536-555)
    zeta = ch["zeta"]
    wires_evals = [poly_eval(f, q, zeta) for q in wire_polys]
    wire_sigma_evals = [poly_eval(f, q, zeta) for q in circuit["sigmas"][:4]]
    perm_next_eval = poly_eval(f, perm_poly, zeta * dom.group_gen % p)
    # Round 5 (:558-690)
    vanish = (pow(zeta, n, p) - 1) % p
    a, b, c, d, e = wires_evals
    ab, cd = a * b % p, c * d % p
    sel = circuit["selectors"]
    polys = list(sel[:13])
    coeffs = [a, b, c, d, ab, cd, pow(a, 5, p), pow(b, 5, p), pow(c, 5, p), pow(d, 5, p), (-e) % p, 1, ab * cd % p * e % p]
    l1 = vanish * pow(n * (zeta - 1) % p, -1, p) % p
    acc = alpha
    for w, k in zip(wires_evals, circuit["k"]):
        acc = acc * ((w + beta * k % p * zeta + gamma) % p) % p
    polys.append(perm_poly)
    coeffs.append((acc + alpha * alpha % p * l1) % p)
    acc = alpha * beta % p * perm_next_eval % p
    for w, s in zip(wires_evals[:4], wire_sigma_evals):
        acc = acc * ((w + beta * s + gamma) % p) % p
    polys.append(circuit["sigmas"][4])
    coeffs.append((-acc) % p)
    z_n2 = (vanish + 1) * zeta % p * zeta % p
    cq = 1
    for q in split:
        polys.append(q)
        coeffs.append((-vanish) * cq % p)
        cq = cq * z_n2 % p
    lin_poly = poly_lincomb(f, polys, coeffs)
    v = ch["v"]
    bp = [lin_poly] + wire_polys + list(circuit["sigmas"][:4])
    batch_poly = poly_lincomb(f, bp, [pow(v, i, p) for i in range(len(bp))])
    opening_proof = commit(poly_div_linear(f, batch_poly, zeta))
    shifted_opening_proof = commit(poly_div_linear(f, perm_poly, zeta * dom.group_gen % p))
    return dict(wires_poly_comms=wires_poly_comms, prod_perm_poly_comm=prod_perm_poly_comm,
                split_quot_poly_comms=split_quot_poly_comms, opening_proof=opening_proof,
                shifted_opening_proof=shifted_opening_proof, wires_evals=wires_evals, wire_sigma_evals=wire_sigma_evals,
                perm_next_eval=perm_next_eval, wire_polys=wire_polys, perm_poly=perm_poly, quot_poly=quot, lin_poly=lin_poly,
                batch_poly=batch_poly)


def make_circuit(f: Field, n: int, rng: random.Random, num_inputs: int = 2):
    """A random SATISFIED TurboPlonk instance with non-trivial copy constraints (test input; the reference takes
    its circuit from jf-plonk's PlonkCircuit, which is not part of this path).  Gate j:
      q_c + pi + sum q_lc_i w_i + q_mul0 ab + q_mul1 cd + q_ecc abcde + sum q_hash_i w_i^5 - q_o e = 0
    a..d reference a pool of free variables or outputs of earlier gates; e is the gate's output variable."""
    p = f.p
    dom = Radix2Domain(f, n)
    witness = [rng.randrange(p) for _ in range(max(2, n // 2))]
    wire_vars = [[0] * n for _ in range(5)]
    sel_ev = [[rng.randrange(p) for _ in range(n)] for _ in range(13)]
    pi = [rng.randrange(p) if j < num_inputs else 0 for j in range(n)]
    for j in range(n):
        for i in range(4):
            wire_vars[i][j] = rng.randrange(len(witness))
        a, b, c, d = (witness[wire_vars[i][j]] for i in range(4))
        s = [sel_ev[t][j] for t in range(13)]
        rest = (s[11] + pi[j] + s[0] * a + s[1] * b + s[2] * c + s[3] * d + s[4] * a * b + s[5] * c * d
                + s[6] * pow(a, 5, p) + s[7] * pow(b, 5, p) + s[8] * pow(c, 5, p) + s[9] * pow(d, 5, p)) % p
        den = (s[10] - s[12] * a * b % p * c % p * d) % p
        witness.append(rest * pow(den, -1, p) % p)
        wire_vars[4][j] = len(witness) - 1
    wires = [[witness[wire_vars[i][j]] for j in range(n)] for i in range(5)]
    k = [1] + [rng.randrange(2, p) for _ in range(4)]
    id_perm = [k[i] * pow(dom.group_gen, j, p) % p for i in range(5) for j in range(n)]
    occ = {}
    for i in range(5):
        for j in range(n):
            occ.setdefault(wire_vars[i][j], []).append(i * n + j)
    perm_idx = [0] * (5 * n)
    for positions in occ.values():
        for t, pos in enumerate(positions):
            perm_idx[pos] = positions[(t + 1) % len(positions)]
    sigmas = [dom.ifft([id_perm[perm_idx[i * n + j]] for j in range(n)]) for i in range(5)]
    selectors = [dom.ifft(ev) for ev in sel_ev]
    return dict(wires=wires, selectors=selectors, sigmas=sigmas, id_perm=id_perm, perm_idx=perm_idx, pub_input=pi, k=k)
