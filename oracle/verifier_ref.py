"""Pairing-free TurboPlonk verifier for a KNOWN-TRAPDOOR SRS (TEST INFRASTRUCTURE ONLY — never imported by the product).

The reference's only end-to-end test proves a circuit and hands the proof to jf-plonk's verifier
(/root/reference/src/dispatcher2.rs:1273-1295 `test2`, /root/reference/src/dispatcher.rs:1118-1134):
`PlonkKzgSnark::verify::<StandardTranscript>(&vk, &public_inputs, &proof)`.  That verifier lives in the un-vendored jellyfish
dependency and ends in a pairing check  e(A, [tau]_2) = e(B, [1]_2).  With an SRS whose trapdoor is known to the test
(P_i = tau^i * G, `trapdoor_srs`) the same statement is  tau * A == B  in G1 — no pairing, no G2.

This file is written from the PROTOCOL, not from the prover: it never looks at how `Prover::prove` builds lin_poly / batch_poly;
it states the constraint system the quotient encodes (dispatcher2.rs:459-504: the gate equation, the permutation argument and the
L1 term that the reference divides by Z_H) and derives what a verifier must check:

  quotient identity at zeta (everything the prover claims, all mod r):
      t(zeta) * Z_H(zeta) = gate(zeta) + alpha * [ z(zeta) * prod_{i<5} (w_i + beta*k_i*zeta + gamma)
                                                  - z(zeta*w) * prod_{i<5} (w_i + beta*sigma_i(zeta) + gamma) ]
                            + alpha^2 * L1(zeta) * (z(zeta) - 1),              t(X) = sum_i X^{i(n+2)} t_i(X)   (:519-523)
  The proof does not carry z(zeta), sigma_4(zeta), the selector values or t_i(zeta): those stay polynomials ("linearisation"),
      r(X) = sum_s c_s(evals) * q_s(X) + [alpha * prod_{i<5}(w_i + beta*k_i*zeta + gamma) + alpha^2 * L1(zeta)] * z(X)
             - [alpha * beta * z_w * prod_{i<4}(w_i + beta*sigma_i + gamma)] * sigma_4(X) - Z_H(zeta) * sum_i zeta^{i(n+2)} t_i(X)
  and the identity becomes a claim about ONE value:
      r(zeta) = -PI(zeta) + alpha * z_w * (w_4 + gamma) * prod_{i<4}(w_i + beta*sigma_i + gamma) + alpha^2 * L1(zeta).
  [r] is a linear combination of commitments the verifier holds, so the claim, the five wire evaluations and the four sigma
  evaluations are checked with one batched KZG opening at zeta (powers of v), and z(zeta*w) with a second opening:
      (tau - zeta)   * W_zeta   == [r] + sum_j v^j [f_j] - (r(zeta) + sum_j v^j f_j(zeta)) * G
      (tau - zeta*w) * W_zeta_w == [z] - z_w * G
  (jf-plonk folds both into one pairing with a further challenge u: tau*(W_z + u*W_zw) == zeta*W_z + u*zeta*w*W_zw + F - E; with the
  trapdoor the two equations are checked separately, which is strictly stronger.  `verify` also evaluates the folded form.)

Challenges are re-derived by the verifier itself from a fresh transcript in the order `Prover::prove` draws them
(dispatcher2.rs:238-241, 323-328, 356-361, 533-543, 555-634); u is drawn after absorbing the two opening proofs under the labels
jf-plonk's verifier uses [upstream, from memory: b"open_proof", b"shifted_open_proof", b"u"] — u only mixes two equations that are
also checked one by one here, so its derivation cannot mask a failure.

Arithmetic: Python integers and the affine group law of oracle/bigint_ref.py (no C oracle, no GPU code): a third, independent
implementation next to plonk_oracle.c and the HIP kernels.
"""
from __future__ import annotations

from . import bigint_ref as B

NUM_WIRE_TYPES = 5
NUM_SELECTORS = 13        # q_lc[4], q_mul[2], q_hash[4], q_o, q_c, q_ecc  (dispatcher2.rs:443-456)


class VerificationError(Exception):
    pass


# ------------------------------------------------------------------------------------------------ conversions (limbs <-> ints)
def fr_int(cv: B.Curve, limbs) -> int:
    return cv.fr.from_mont(B.from_limbs([int(x) for x in limbs]))


def fr_limbs(cv: B.Curve, x: int):
    import numpy as np
    return np.array(B.to_limbs(cv.fr.to_mont(x % cv.fr.p), 4), dtype=np.uint64)


def point_int(cv: B.Curve, pt):
    """(xy Montgomery limbs, is_infinity) as the C ABI / the C oracle return affine points -> (x, y) ints or INF."""
    xy, inf = pt
    if inf:
        return B.INF
    q = cv.fq.limbs64
    xy = [int(v) for v in xy]
    P = (cv.fq.from_mont(B.from_limbs(xy[:q])), cv.fq.from_mont(B.from_limbs(xy[q:])))
    if not B.on_curve(cv, P):
        raise VerificationError("commitment is not on the curve")
    return P


def point_limbs(cv: B.Curve, P):
    import numpy as np
    q = cv.fq.limbs64
    if P is B.INF:
        return np.zeros(2 * q, dtype=np.uint64), True
    return np.array(B.to_limbs(cv.fq.to_mont(P[0]), q) + B.to_limbs(cv.fq.to_mont(P[1]), q), dtype=np.uint64), False


# ------------------------------------------------------------------------------------------------ G1 helpers
def g1_lincomb(cv: B.Curve, terms):
    """sum_k s_k * P_k for a handful of (scalar, point) pairs."""
    acc = B.INF
    r = cv.fr.p
    for s, P in terms:
        s %= r
        if s and P is not B.INF:
            acc = B.affine_add(cv, acc, B.scalar_mul(cv, s, P))
    return acc


def commit_by_trapdoor(cv: B.Curve, coeffs_int, tau: int):
    """KZG commitment of a polynomial under the trapdoor SRS without touching the SRS: f(tau) * G."""
    r = cv.fr.p
    acc = 0
    for c in reversed(coeffs_int):
        acc = (acc * tau + c) % r
    return B.scalar_mul(cv, acc, (cv.gx, cv.gy)) if acc else B.INF


def trapdoor_srs(cv: B.Curve, tau: int, count: int):
    """[G, tau G, tau^2 G, ...] as affine int points — universal_setup with a published trapdoor (test2 draws it from rng and
    forgets it).  Pure Python: use for small sizes; the tests build larger keys with the C oracle's scalar_mul."""
    out, s = [], 1
    for _ in range(count):
        out.append(B.scalar_mul(cv, s, (cv.gx, cv.gy)))
        s = s * tau % cv.fr.p
    return out


# ------------------------------------------------------------------------------------------------ the verifier
def lagrange_pi_eval(cv: B.Curve, n: int, public_inputs, zeta: int) -> int:
    """PI(zeta) for the public-input polynomial whose evaluations on H are public_inputs followed by zeros (dispatcher2.rs:426):
    sum_i pi_i * L_i(zeta),  L_i(X) = w^i (X^n - 1) / (n (X - w^i))."""
    f = cv.fr
    r = f.p
    w = f.root_of_unity(n)
    zh = (pow(zeta, n, r) - 1) % r
    acc, wi = 0, 1
    for x in public_inputs:
        if zeta == wi:
            return x % r if zh == 0 else 0      # zeta on the domain: not reachable with a random challenge
        acc = (acc + x * wi % r * zh % r * pow(n * (zeta - wi) % r, -1, r)) % r
        wi = wi * w % r
    return acc


def derive_challenges(transcript, vk: dict, public_inputs_limbs, proof: dict) -> dict:
    """Fiat-Shamir on the verifier's side.  `transcript`: a fresh object with the methods of dispatcher2.rs:44-154
    (append_vk_and_pub_input, append_commitments, append_commitment, append_proof_evaluations, get_and_append_challenge)."""
    t = transcript
    t.append_vk_and_pub_input(vk["domain_size"], len(public_inputs_limbs), list(vk["k"]), vk["selector_comms"], vk["sigma_comms"],
                              list(public_inputs_limbs))                                   # :238-241
    t.append_commitments(b"witness_poly_comms", proof["wires_poly_comms"])               # :323
    ch = {"beta": t.get_and_append_challenge(b"beta"), "gamma": t.get_and_append_challenge(b"gamma")}   # :327-328
    t.append_commitment(b"perm_poly_comms", proof["prod_perm_poly_comm"])                # :356
    ch["alpha"] = t.get_and_append_challenge(b"alpha")                                    # :361
    t.append_commitments(b"quot_poly_comms", proof["split_quot_poly_comms"])             # :533
    ch["zeta"] = t.get_and_append_challenge(b"zeta")                                      # :543
    t.append_proof_evaluations(proof["wires_evals"], proof["wire_sigma_evals"], proof["perm_next_eval"])    # :555
    ch["v"] = t.get_and_append_challenge(b"v")                                            # :634
    t.append_commitment(b"open_proof", proof["opening_proof"])                            # [upstream jf-plonk verifier]
    t.append_commitment(b"shifted_open_proof", proof["shifted_opening_proof"])
    ch["u"] = t.get_and_append_challenge(b"u")
    return ch


def verify(cv: B.Curve, vk: dict, public_inputs_limbs, proof: dict, tau: int, transcript=None, challenges: dict | None = None) -> dict:
    """vk: {"domain_size": n, "k": 5 Fr limbs, "selector_comms": 13 points, "sigma_comms": 5 points}; points are
    (xy Montgomery limbs, is_infinity).  proof: the fields of `Proof` (dispatcher2.rs:699-710) in the same encodings.
    public_inputs_limbs: `circuit.public_input()` (NOT padded).  Challenges come from `transcript` (fresh) or, for provers run
    with caller-chosen challenges, from `challenges` (limbs; "u" optional).  Raises VerificationError; returns the intermediate
    values on success."""
    f = cv.fr
    r = f.p
    n = int(vk["domain_size"])
    if n & (n - 1) or n < 2:
        raise VerificationError("domain size")
    if (len(proof["wires_poly_comms"]) != NUM_WIRE_TYPES or len(proof["split_quot_poly_comms"]) != NUM_WIRE_TYPES
            or len(proof["wires_evals"]) != NUM_WIRE_TYPES or len(proof["wire_sigma_evals"]) != NUM_WIRE_TYPES - 1
            or len(vk["selector_comms"]) != NUM_SELECTORS or len(vk["sigma_comms"]) != NUM_WIRE_TYPES):
        raise VerificationError("proof / key shape")
    if challenges is None:
        if transcript is None:
            raise ValueError("verify needs a transcript or explicit challenges")
        challenges = derive_challenges(transcript, vk, public_inputs_limbs, proof)
    I = lambda l: fr_int(cv, l)
    beta, gamma, alpha, zeta, v = (I(challenges[k]) for k in ("beta", "gamma", "alpha", "zeta", "v"))
    u = I(challenges["u"]) if "u" in challenges else 0x5EED
    G = (cv.gx, cv.gy)
    P = lambda pt: point_int(cv, pt)
    w_c = [P(c) for c in proof["wires_poly_comms"]]
    z_c = P(proof["prod_perm_poly_comm"])
    t_c = [P(c) for c in proof["split_quot_poly_comms"]]
    W_z, W_zw = P(proof["opening_proof"]), P(proof["shifted_opening_proof"])
    q_c = [P(c) for c in vk["selector_comms"]]
    s_c = [P(c) for c in vk["sigma_comms"]]
    k = [I(x) for x in vk["k"]]
    a, b, c, d, e = wv = [I(x) for x in proof["wires_evals"]]
    sg = [I(x) for x in proof["wire_sigma_evals"]]
    z_w = I(proof["perm_next_eval"])
    pis = [I(x) for x in public_inputs_limbs]

    omega = f.root_of_unity(n)
    zh = (pow(zeta, n, r) - 1) % r
    if zh == 0 or zeta == 1:
        raise VerificationError("zeta lies on the evaluation domain")
    l1 = zh * pow(n * (zeta - 1) % r, -1, r) % r
    pi_z = lagrange_pi_eval(cv, n, pis, zeta)

    # gate(X) linearised: the wire values are numbers, the selectors stay commitments.   (gate equation: dispatcher2.rs:465-477)
    ab, cd = a * b % r, c * d % r
    sel_coeff = [a, b, c, d,                                         # q_lc[0..3] * w_i
                 ab, cd,                                             # q_mul[0] * ab, q_mul[1] * cd
                 pow(a, 5, r), pow(b, 5, r), pow(c, 5, r), pow(d, 5, r),   # q_hash[i] * w_i^5
                 (-e) % r,                                           # - q_o * e
                 1,                                                  # q_c
                 ab * cd % r * e % r]                                # q_ecc * a b c d e
    terms = list(zip(sel_coeff, q_c))
    # permutation argument, the factor multiplying z(X) and the one multiplying sigma_4(X)            (:478-492)
    pz = alpha
    for wi, ki in zip(wv, k):
        pz = pz * ((wi + beta * ki % r * zeta + gamma) % r) % r
    terms.append(((pz + alpha * alpha % r * l1) % r, z_c))                                           # + alpha^2 L1 z(X)  (:494-500)
    ps = alpha * z_w % r
    for wi, si in zip(wv[:4], sg):
        ps = ps * ((wi + beta * si + gamma) % r) % r
    terms.append(((-ps * beta) % r, s_c[4]))
    # - Z_H(zeta) * t(X),  t = sum_i X^{i(n+2)} t_i                                                  (:519-523)
    zn2 = pow(zeta, n + 2, r)
    cq = 1
    for tc in t_c:
        terms.append(((-zh * cq) % r, tc))
        cq = cq * zn2 % r
    D = g1_lincomb(cv, terms)                                        # [r]
    r_zeta = (-pi_z + ps * ((e + gamma) % r) + alpha * alpha % r * l1) % r

    # batched opening at zeta: r, w_0..w_4, sigma_0..sigma_3 with powers of v
    polys = [D] + w_c + s_c[:4]
    evals = [r_zeta] + wv + sg
    F = g1_lincomb(cv, [(pow(v, j, r), Pj) for j, Pj in enumerate(polys)])
    E = sum(pow(v, j, r) * ej for j, ej in enumerate(evals)) % r
    lhs1 = g1_lincomb(cv, [((tau - zeta) % r, W_z)])
    rhs1 = g1_lincomb(cv, [(1, F), ((-E) % r, G)])
    if lhs1 != rhs1:
        raise VerificationError("opening at zeta rejected: (tau - zeta) * W_zeta != F - E*G")
    zeta_w = zeta * omega % r
    lhs2 = g1_lincomb(cv, [((tau - zeta_w) % r, W_zw)])
    rhs2 = g1_lincomb(cv, [(1, z_c), ((-z_w) % r, G)])
    if lhs2 != rhs2:
        raise VerificationError("opening at zeta*w rejected: (tau - zeta w) * W_zeta_w != [z] - z_w*G")
    # jf-plonk's single folded check, as the pairing would see it
    A = g1_lincomb(cv, [(1, W_z), (u, W_zw)])
    Bp = g1_lincomb(cv, [(zeta, W_z), (u * zeta_w % r, W_zw), (1, F), (u, z_c), ((-(E + u * z_w)) % r, G)])
    if g1_lincomb(cv, [(tau, A)]) != Bp:
        raise VerificationError("folded check rejected")
    return dict(challenges=challenges, lin_comm=D, lin_eval=r_zeta, batch_comm=F, batch_eval=E, pi_eval=pi_z)


def vk_by_trapdoor(cv: B.Curve, n: int, k_limbs, selectors_int, sigmas_int, tau: int) -> dict:
    """The verifying key `preprocess` would produce (dispatcher2.rs:1280): commitments of the 13 selector and 5 sigma polynomials
    (coefficient form, plain residues), computed as f(tau)*G — independent of any MSM."""
    return dict(domain_size=n, k=list(k_limbs),
                selector_comms=[point_limbs(cv, commit_by_trapdoor(cv, q, tau)) for q in selectors_int],
                sigma_comms=[point_limbs(cv, commit_by_trapdoor(cv, s, tau)) for s in sigmas_int])
