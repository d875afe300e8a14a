"""Rounds 1-5 of the reference prover restated over the C oracle's primitives (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED (no reference golden vectors; see oracle/plonk_oracle.c).  Follows
/root/reference/src/dispatcher2.rs::Prover::prove (:296-712) with the transcript challenges and the blinding
polynomials supplied by the caller (merlin transcript = SURVEY §8f rank 4, out of scope).  Every array is numpy
uint64 limbs, Fr in Montgomery form (utils.rs:27-43).  Scalar (challenge) arithmetic uses Python ints.
tests/ pins this against oracle/bigint_ref.py::prove_rounds on small circuits.
"""
from __future__ import annotations

import numpy as np

from . import bigint_ref as B
from . import oracle as O

CURVE_OBJ = {O.BN254: B.BN254, O.BLS12_381: B.BLS12_381}


def fr_to_limbs(f: B.Field, x: int) -> np.ndarray:
    """plain residue -> Montgomery limbs (4,)"""
    return np.array(B.to_limbs(f.to_mont(x % f.p), 4), dtype=np.uint64)


def fr_from_limbs(f: B.Field, l) -> int:
    return f.from_mont(B.from_limbs(l))


def fr_vec_to_limbs(f: B.Field, xs) -> np.ndarray:
    return np.array([B.to_limbs(f.to_mont(x % f.p), 4) for x in xs], dtype=np.uint64).reshape(len(xs), 4)


def make_circuit(curve: int, log_n: int, seed: int, num_inputs: int = 2):
    """Vectorised counterpart of bigint_ref.make_circuit (a random SATISFIED TurboPlonk instance): gates in the first
    half read a pool of free variables, gates in the second half also read first-half outputs, so copy constraints
    tie all five wire columns together.  Returns Montgomery limb arrays:
    wires (5,n,4), selectors (13,n,4) coefficient form, sigmas (5,n,4) coefficient form, id_perm (5n,4),
    perm_idx (5n,) u64, pub_input (n,4) evaluations, k (5,4)."""
    f = CURVE_OBJ[curve].fr
    n = 1 << log_n
    rs = np.random.RandomState(seed)
    pool = max(2, n // 2)
    h = n // 2
    fop = lambda op, a, b=None: O.field_op(curve, 0, op, a, b)
    MUL, ADD, SUB, INV = "mul", "add", "sub", "inv"
    wit_pool = O.rand_fr(curve, seed * 7 + 1, pool)
    sel_ev = O.rand_fr(curve, seed * 7 + 2, 13 * n).reshape(13, n, 4)
    pi = np.zeros((n, 4), dtype=np.uint64)
    pi[:num_inputs] = O.rand_fr(curve, seed * 7 + 3, num_inputs)
    wire_vars = np.zeros((5, n), dtype=np.int64)      # variable ids: [0,pool) free, pool + j = output of gate j
    wire_vars[:4, :h] = rs.randint(0, pool, size=(4, h))
    wire_vars[:4, h:] = rs.randint(0, pool + h, size=(4, n - h))
    wire_vars[4] = pool + np.arange(n)
    values = np.zeros((pool + n, 4), dtype=np.uint64)
    values[:pool] = wit_pool

    def solve(lo, hi):
        a, b, c, d = (values[wire_vars[i, lo:hi]] for i in range(4))
        s = sel_ev[:, lo:hi]
        p5 = lambda x: fop(MUL, fop(MUL, fop(MUL, x, x), fop(MUL, x, x)), x)
        ab, cd = fop(MUL, a, b), fop(MUL, c, d)
        rest = fop(ADD, s[11], pi[lo:hi])
        for t, w in ((0, a), (1, b), (2, c), (3, d), (4, ab), (5, cd), (6, p5(a)), (7, p5(b)), (8, p5(c)), (9, p5(d))):
            rest = fop(ADD, rest, fop(MUL, s[t], w))
        den = fop(SUB, s[10], fop(MUL, s[12], fop(MUL, ab, cd)))
        values[pool + lo:pool + hi] = fop(MUL, rest, fop(INV, den))

    solve(0, h)
    solve(h, n)
    wires = np.stack([values[wire_vars[i]] for i in range(5)])
    dom = B.Radix2Domain(f, n)
    kk = [1] + [int(x) for x in rs.randint(2, 1 << 62, size=4)]
    k = fr_vec_to_limbs(f, kk)
    # id_perm[i*n + j] = k_i * w^j : powers of w by repeated vector doubling
    pw = np.zeros((n, 4), dtype=np.uint64)
    pw[0] = fr_to_limbs(f, 1)
    filled = 1
    while filled < n:
        step = fr_to_limbs(f, pow(dom.group_gen, filled, f.p))
        cnt = min(filled, n - filled)
        pw[filled:filled + cnt] = fop(MUL, pw[:cnt], np.broadcast_to(step, (cnt, 4)).copy())
        filled += cnt
    id_perm = np.concatenate([fop(MUL, pw, np.broadcast_to(k[i], (n, 4)).copy()) for i in range(5)])
    # copy-constraint cycles: next occurrence of the same variable
    flat = wire_vars.reshape(-1)
    order = np.argsort(flat, kind="stable")
    sv = flat[order]
    nxt = np.roll(order, -1)
    starts = np.flatnonzero(np.r_[True, sv[1:] != sv[:-1]])
    ends = np.r_[starts[1:], len(sv)] - 1
    nxt[ends] = order[starts]
    perm_idx = np.empty(5 * n, dtype=np.uint64)
    perm_idx[order] = nxt.astype(np.uint64)
    sig_ev = id_perm[perm_idx.astype(np.int64)].reshape(5, n, 4)
    sigmas = np.stack([O.ntt(curve, sig_ev[i], True, False) for i in range(5)])
    selectors = np.stack([O.ntt(curve, sel_ev[t], True, False) for t in range(13)])
    return dict(wires=wires, selectors=selectors, sigmas=sigmas, id_perm=id_perm, perm_idx=perm_idx, pub_input=pi, k=k)


def circuit_to_ints(curve: int, circ: dict) -> dict:
    """Montgomery limb arrays -> plain residues for bigint_ref.prove_rounds."""
    f = CURVE_OBJ[curve].fr
    cv = lambda a: [fr_from_limbs(f, r) for r in a]
    return dict(wires=[cv(w) for w in circ["wires"]], selectors=[cv(s) for s in circ["selectors"]],
                sigmas=[cv(s) for s in circ["sigmas"]], id_perm=cv(circ["id_perm"]), perm_idx=[int(x) for x in circ["perm_idx"]],
                pub_input=cv(circ["pub_input"]), k=cv(circ["k"]))


def make_ck(curve: int, n: int, seed: int, unique: int = 64):
    """commit key: n+3 powers padded with the point at infinity to a multiple of 32 (dispatcher2.rs:206-208).
    -> (bases (N, 2Q) x||y, inf flags (N,) u8)."""
    cnt = n + 3
    N = ((cnt + 31) >> 5) << 5
    Q = O.FQ_LIMBS[curve]
    bases = np.zeros((N, 2 * Q), dtype=np.uint64)
    bases[:cnt] = O.gen_bases(curve, seed, min(unique, cnt), cnt)
    inf = np.zeros(N, dtype=np.uint8)
    inf[cnt:] = 1
    return bases, inf


def make_ck_trapdoor(curve: int, n: int, tau: int):
    """universal_setup with a KNOWN trapdoor (dispatcher2.rs:1278 draws tau from rng and forgets it): P_i = tau^i * G for the
    n + 3 powers the prover needs, padded with the point at infinity to a multiple of 32 (dispatcher2.rs:206-208).  With it the
    final pairing check of the verifier becomes an equation in G1 (oracle/verifier_ref.py).  -> (bases x||y, inf flags)."""
    f = CURVE_OBJ[curve].fr
    cnt = n + 3
    N = ((cnt + 31) >> 5) << 5
    Q = O.FQ_LIMBS[curve]
    bases = np.zeros((N, 2 * Q), dtype=np.uint64)
    g = O.generator(curve)
    s = 1
    for i in range(cnt):
        k = np.array(B.to_limbs(s, 4), dtype=np.uint64)             # canonical scalar
        bases[i], is_inf = O.jac_to_affine(curve, O.scalar_mul(curve, g, k))
        assert not is_inf
        s = s * tau % f.p
    inf = np.zeros(N, dtype=np.uint8)
    inf[cnt:] = 1
    return bases, inf


def prove_rounds(curve: int, log_n: int, ck, ck_inf, circ: dict, blinders: dict, ch, threads: int = 1):
    """dispatcher2.rs:296-712.  blinders: {"wires": (5,2,4), "perm": (3,4)}; ch: {"beta","gamma","alpha","zeta","v"}
    as Montgomery limbs (4,), or a callable (label, proof_so_far) -> limbs that is asked where `prove` draws each challenge
    (a Fiat-Shamir transcript).  Returns commitments as affine (xy, is_inf) plus evaluations and intermediate polys."""
    if callable(ch):
        draw, ch, so_far = ch, {}, {}

        def need(label):
            ch[label] = draw(label, so_far)
    else:
        so_far = {}

        def need(label):
            pass
    f = CURVE_OBJ[curve].fr
    p = f.p
    n = 1 << log_n
    m = 8 * n
    dom = B.Radix2Domain(f, n)
    L = lambda x: fr_to_limbs(f, x)
    I = lambda l: fr_from_limbs(f, l)

    def commit(coeffs):
        return O.jac_to_affine(curve, O.commit_polynomial(curve, ck, coeffs, inf=ck_inf, threads=threads))

    def cfft(coeffs):
        v = np.zeros((m, 4), dtype=np.uint64)
        v[:coeffs.shape[0]] = coeffs
        return O.ntt(curve, v, False, True, threads=threads)

    # Round 1
    wire_polys = [O.blind(curve, O.ntt(curve, circ["wires"][i], True, False, threads=threads), n, blinders["wires"][i]) for i in range(5)]
    wires_poly_comms = so_far["wires_poly_comms"] = [commit(q) for q in wire_polys]
    # Round 2
    need("beta"); need("gamma")
    prod = O.perm_product(curve, circ["wires"], circ["id_perm"], circ["perm_idx"], ch["beta"], ch["gamma"])
    perm_poly = O.blind(curve, O.ntt(curve, prod, True, False, threads=threads), n, blinders["perm"])
    prod_perm_poly_comm = so_far["prod_perm_poly_comm"] = commit(perm_poly)
    # Round 3
    need("alpha")
    pi_poly = O.ntt(curve, circ["pub_input"], True, False, threads=threads)
    evals = O.quotient_evals(curve, log_n, np.stack([cfft(q) for q in circ["selectors"]]), np.stack([cfft(q) for q in circ["sigmas"]]),
                             np.stack([cfft(q) for q in wire_polys]), cfft(perm_poly), cfft(pi_poly), ch["alpha"], ch["beta"], ch["gamma"],
                             circ["k"], threads=threads)
    quot = O.ntt(curve, evals, True, True, threads=threads)
    deg = m - 1
    while deg > 0 and not quot[deg].any():
        deg -= 1
    expected = 5 * (n + 1) + 2
    if deg != expected:
        raise ValueError(f"WrongQuotientPolyDegree({deg}, {expected})")
    split = [quot[i:min(i + n + 2, deg + 1)] for i in range(0, deg + 1, n + 2)]
    split_quot_poly_comms = so_far["split_quot_poly_comms"] = [commit(q) for q in split]
    # Round 4
    need("zeta")
    zeta = ch["zeta"]
    zeta_w = L(I(zeta) * dom.group_gen)
    wires_evals = [O.poly_eval(curve, q, zeta) for q in wire_polys]
    wire_sigma_evals = [O.poly_eval(curve, circ["sigmas"][i], zeta) for i in range(4)]
    perm_next_eval = O.poly_eval(curve, perm_poly, zeta_w)
    so_far.update(wires_evals=wires_evals, wire_sigma_evals=wire_sigma_evals, perm_next_eval=perm_next_eval)
    need("v")
    # Round 5 (scalar coefficients in Python ints)
    z, al, be, ga, v = I(zeta), I(ch["alpha"]), I(ch["beta"]), I(ch["gamma"]), I(ch["v"])
    a, b, c, d, e = (I(x) for x in wires_evals)
    sg = [I(x) for x in wire_sigma_evals]
    kk = [I(x) for x in circ["k"]]
    vanish = (pow(z, n, p) - 1) % p
    ab, cd = a * b % p, c * d % p
    polys = [circ["selectors"][t] for t in range(13)]
    coeffs = [a, b, c, d, ab, cd, pow(a, 5, p), pow(b, 5, p), pow(c, 5, p), pow(d, 5, p), (-e) % p, 1, ab * cd % p * e % p]
    l1 = vanish * pow(n * (z - 1) % p, -1, p) % p
    acc = al
    for w, k_ in zip((a, b, c, d, e), kk):
        acc = acc * ((w + be * k_ % p * z + ga) % p) % p
    polys.append(perm_poly)
    coeffs.append((acc + al * al % p * l1) % p)
    acc = al * be % p * I(perm_next_eval) % p
    for w, s in zip((a, b, c, d), sg):
        acc = acc * ((w + be * s + ga) % p) % p
    polys.append(circ["sigmas"][4])
    coeffs.append((-acc) % p)
    z_n2 = (vanish + 1) * z % p * z % p
    cq = 1
    for q in split:
        polys.append(q)
        coeffs.append((-vanish) * cq % p)
        cq = cq * z_n2 % p
    lin_poly = O.poly_lincomb(curve, polys, fr_vec_to_limbs(f, coeffs))
    bp = [lin_poly] + wire_polys + [circ["sigmas"][i] for i in range(4)]
    batch_poly = O.poly_lincomb(curve, bp, fr_vec_to_limbs(f, [pow(v, i, p) for i in range(len(bp))]))
    opening_proof = commit(O.poly_div_linear(curve, batch_poly, zeta))
    shifted_opening_proof = commit(O.poly_div_linear(curve, perm_poly, zeta_w))
    return dict(wires_poly_comms=wires_poly_comms, prod_perm_poly_comm=prod_perm_poly_comm,
                split_quot_poly_comms=split_quot_poly_comms, opening_proof=opening_proof,
                shifted_opening_proof=shifted_opening_proof, wires_evals=wires_evals, wire_sigma_evals=wire_sigma_evals,
                perm_next_eval=perm_next_eval, wire_polys=wire_polys, perm_poly=perm_poly, perm_product=prod, quot_poly=quot[:deg + 1],
                lin_poly=lin_poly, batch_poly=batch_poly)
