"""CPU pinning of the oracle's §8f rank 2/3 restatements: the C oracle (oracle/plonk_oracle.c) against the independent
Python big-integer statement (oracle/bigint_ref.py), plus algebraic anchors that do not depend on either (the division
identity, Horner vs direct sum, and the quotient-degree check of a satisfied circuit, dispatcher2.rs:511-518).
PARITY UNPINNED by reference data: the reference holds no vectors for these rows."""
import random

import numpy as np
import pytest

from oracle import bigint_ref as B
from oracle import prover_ref as P

CURVES = [(0, B.BN254), (1, B.BLS12_381)]


def _ints(f, arr):
    return [P.fr_from_limbs(f, r) for r in np.asarray(arr).reshape(-1, 4)]


@pytest.mark.parametrize("cid,cv", CURVES)
@pytest.mark.parametrize("n", [2, 3, 16, 100])
def test_perm_product_c_vs_bigint(oracle, cid, cv, n):
    f = cv.fr
    rs = np.random.RandomState(n)
    wires = oracle.rand_fr(cid, 1, 5 * n).reshape(5, n, 4)
    id_perm = oracle.rand_fr(cid, 2, 5 * n)
    perm_idx = rs.permutation(5 * n).astype(np.uint64)
    beta, gamma = oracle.rand_fr(cid, 3, 2)
    got = oracle.perm_product(cid, wires, id_perm, perm_idx, beta, gamma)
    want = B.perm_product(f, n, [_ints(f, w) for w in wires], _ints(f, id_perm), [int(x) for x in perm_idx],
                          P.fr_from_limbs(f, beta), P.fr_from_limbs(f, gamma))
    assert _ints(f, got) == want


@pytest.mark.parametrize("cid,cv", CURVES)
def test_poly_ops_c_vs_bigint(oracle, cid, cv):
    f = cv.fr
    for length in (1, 2, 9, 300):
        poly = oracle.rand_fr(cid, 40 + length, length)
        z = oracle.rand_fr(cid, 41, 1)[0]
        pi, zi = _ints(f, poly), P.fr_from_limbs(f, z)
        assert P.fr_from_limbs(f, oracle.poly_eval(cid, poly, z)) == B.poly_eval(f, pi, zi) == sum(c * pow(zi, i, f.p) for i, c in enumerate(pi)) % f.p
        q = _ints(f, oracle.poly_div_linear(cid, poly, z))
        assert q == B.poly_div_linear(f, pi, zi)
        # (X - z) q + poly(z) == poly
        back = [0] * length
        for i, c in enumerate(q):
            back[i + 1] = (back[i + 1] + c) % f.p
            back[i] = (back[i] - c * zi) % f.p
        back[0] = (back[0] + B.poly_eval(f, pi, zi)) % f.p
        assert back == pi
    polys = [oracle.rand_fr(cid, 50 + i, L) for i, L in enumerate((5, 9, 1, 9))]
    cf = oracle.rand_fr(cid, 60, 4)
    assert _ints(f, oracle.poly_lincomb(cid, polys, cf)) == B.poly_lincomb(f, [_ints(f, q) for q in polys], _ints(f, cf))
    bl = oracle.rand_fr(cid, 61, 3)
    assert _ints(f, oracle.blind(cid, polys[1][:8], 8, bl)) == B.blind(f, _ints(f, polys[1][:8]), 8, _ints(f, bl))


def test_poly_div_trailing_zero_coefficients(oracle):
    """DensePolynomial trims trailing zeros before the loop of dispatcher2.rs:651-666; the dense recurrence must give the
    same quotient followed by zeros."""
    f = B.BN254.fr
    poly = oracle.rand_fr(0, 70, 12)
    poly[-3:] = 0
    z = oracle.rand_fr(0, 71, 1)[0]
    q = _ints(f, oracle.poly_div_linear(0, poly, z))
    lit = B.poly_div_linear(f, _ints(f, poly), P.fr_from_limbs(f, z))
    assert q[:len(lit)] == lit and all(c == 0 for c in q[len(lit):])


@pytest.mark.parametrize("cid,cv", CURVES)
def test_prove_rounds_c_vs_bigint(oracle, cid, cv):
    """Rounds 1-5 (dispatcher2.rs:296-712) on a satisfied random circuit: the C-primitive prover and the big-integer prover
    agree on every commitment and evaluation; both pass the reference's quotient-degree check (:511-518), which only holds
    when grand product, quotient evaluations and all NTTs are mutually consistent."""
    f, fq = cv.fr, cv.fq
    log_n, n = 3, 8
    circ = P.make_circuit(cid, log_n, seed=3)
    ck, inf = P.make_ck(cid, n, seed=9, unique=8)
    bl = dict(wires=oracle.rand_fr(cid, 11, 10).reshape(5, 2, 4), perm=oracle.rand_fr(cid, 12, 3))
    ch = {k: oracle.rand_fr(cid, 20 + i, 1)[0] for i, k in enumerate(("beta", "gamma", "alpha", "zeta", "v"))}
    pr = P.prove_rounds(cid, log_n, ck, inf, circ, bl, ch)
    Q = oracle.FQ_LIMBS[cid]
    pts = [None if inf[i] else (fq.from_mont(B.from_limbs(ck[i][:Q])), fq.from_mont(B.from_limbs(ck[i][Q:]))) for i in range(len(ck))]
    bli = dict(wires=[_ints(f, w) for w in bl["wires"]], perm=_ints(f, bl["perm"]))
    pb = B.prove_rounds(cv, n, pts, P.circuit_to_ints(cid, circ), bli, {k: P.fr_from_limbs(f, v) for k, v in ch.items()})

    def aff(x):
        xy, isinf = x
        return None if isinf else (fq.from_mont(B.from_limbs(xy[:Q])), fq.from_mont(B.from_limbs(xy[Q:])))

    for key in ("wires_poly_comms", "split_quot_poly_comms"):
        assert [aff(x) for x in pr[key]] == pb[key], key
    for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
        assert aff(pr[key]) == pb[key], key
    for key in ("wires_evals", "wire_sigma_evals"):
        assert [P.fr_from_limbs(f, x) for x in pr[key]] == pb[key], key
    assert P.fr_from_limbs(f, pr["perm_next_eval"]) == pb["perm_next_eval"]
    assert _ints(f, pr["batch_poly"]) == pb["batch_poly"]
    assert len(pb["quot_poly"]) == 5 * n + 8


def test_unsatisfied_circuit_fails_degree_check(oracle):
    """Flip one witness value: the quotient no longer divides and the reference returns WrongQuotientPolyDegree."""
    cid, log_n = 0, 3
    circ = P.make_circuit(cid, log_n, seed=4)
    circ["wires"] = circ["wires"].copy()
    circ["wires"][4, 5] = oracle.rand_fr(cid, 99, 1)[0]
    ck, inf = P.make_ck(cid, 8, seed=9, unique=8)
    bl = dict(wires=oracle.rand_fr(cid, 11, 10).reshape(5, 2, 4), perm=oracle.rand_fr(cid, 12, 3))
    ch = {k: oracle.rand_fr(cid, 20 + i, 1)[0] for i, k in enumerate(("beta", "gamma", "alpha", "zeta", "v"))}
    with pytest.raises(ValueError, match="WrongQuotientPolyDegree"):
        P.prove_rounds(cid, log_n, ck, inf, circ, bl, ch)
