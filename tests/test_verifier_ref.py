"""CPU pin of oracle/verifier_ref.py — the pairing-free verifier for a known-trapdoor SRS that plays the part of
`PlonkKzgSnark::verify` in the reference's only end-to-end test (/root/reference/src/dispatcher2.rs:1273-1295).

Here the proofs come from the oracle's restatement of `Prover::prove`; tests/test_gpu_verifier.py feeds it the device prover's
output.  The verifier is written from the protocol (pure Python ints), so a proof it accepts satisfies the PLONK equations for the
selector order, permutation argument and linearisation the reference uses — not merely "equals the builder's restatement"."""
import numpy as np
import pytest

from distributed_plonk_amd.transcript import PlonkTranscript
from oracle import bigint_ref as B
from oracle import verifier_ref as V

CURVES = [("bn254", 0), ("bls12_381", 1)]
TAU = 0x1D0C5EED_0BADC0DE_12345678_9ABCDEF1_0F1E2D3C_4B5A6978        # any non-zero trapdoor


class ProverSideTranscript:
    """What `Prover::prove` appends before each challenge (dispatcher2.rs:323, 327-328, 356, 361, 533, 543, 555, 634)."""

    def __init__(self, curve, vk, pub):
        self.t = PlonkTranscript(curve)
        self.t.append_vk_and_pub_input(vk["domain_size"], len(pub), list(vk["k"]), vk["selector_comms"], vk["sigma_comms"], list(pub))

    def __call__(self, label, proof):
        t = self.t
        if label == "beta":
            t.append_commitments(b"witness_poly_comms", proof["wires_poly_comms"])
        elif label == "alpha":
            t.append_commitment(b"perm_poly_comms", proof["prod_perm_poly_comm"])
        elif label == "zeta":
            t.append_commitments(b"quot_poly_comms", proof["split_quot_poly_comms"])
        elif label == "v":
            t.append_proof_evaluations(proof["wires_evals"], proof["wire_sigma_evals"], proof["perm_next_eval"])
        return t.get_and_append_challenge(label.encode())


def make_instance(oracle, curve, cid, log_n, seed, num_inputs=2):
    from oracle import prover_ref as P
    cv = B.CURVES[curve]
    n = 1 << log_n
    circ = P.make_circuit(cid, log_n, seed=seed, num_inputs=num_inputs)
    ck, inf = P.make_ck_trapdoor(cid, n, TAU)
    ints = P.circuit_to_ints(cid, circ)
    vk = V.vk_by_trapdoor(cv, n, circ["k"], ints["selectors"], ints["sigmas"], TAU)
    bl = dict(wires=oracle.rand_fr(cid, seed + 2, 10).reshape(5, 2, 4), perm=oracle.rand_fr(cid, seed + 3, 3))
    pub = circ["pub_input"][:num_inputs]
    return P, cv, circ, ck, inf, vk, bl, pub


PROOF_KEYS = ("wires_poly_comms", "prod_perm_poly_comm", "split_quot_poly_comms", "opening_proof", "shifted_opening_proof",
              "wires_evals", "wire_sigma_evals", "perm_next_eval")


@pytest.fixture(scope="module")
def proved(oracle):
    cache = {}

    def get(curve, cid, log_n):
        key = (curve, log_n)
        if key not in cache:
            P, cv, circ, ck, inf, vk, bl, pub = make_instance(oracle, curve, cid, log_n, 500 + log_n)
            full = P.prove_rounds(cid, log_n, ck, inf, circ, bl, ProverSideTranscript(curve, vk, pub), threads=8)
            cache[key] = (cv, circ, ck, inf, vk, pub, {k: full[k] for k in PROOF_KEYS})
        return cache[key]
    return get


@pytest.mark.parametrize("curve,cid", CURVES)
def test_trapdoor_key_is_the_kzg_key(oracle, curve, cid):
    """commit(f) under P_i = tau^i G equals f(tau) G: the C oracle's MSM route and the verifier's Horner route agree, and the
    pure-Python SRS agrees with the C one."""
    from oracle import prover_ref as P
    cv = B.CURVES[curve]
    n = 16
    ck, inf = P.make_ck_trapdoor(cid, n, TAU)
    assert inf[:n + 3].sum() == 0 and inf[n + 3:].all()
    srs = V.trapdoor_srs(cv, TAU, 5)
    for i in range(5):
        assert V.point_int(cv, (ck[i], False)) == srs[i]
    poly = oracle.rand_fr(cid, 9, n + 3)
    got = oracle.jac_to_affine(cid, oracle.commit_polynomial(cid, ck, poly, inf=inf, threads=2))
    want = V.commit_by_trapdoor(cv, [V.fr_int(cv, c) for c in poly], TAU)
    assert V.point_int(cv, got) == want
    xy, is_inf = V.point_limbs(cv, want)
    assert not is_inf and np.array_equal(xy, got[0])


@pytest.mark.parametrize("curve,cid", CURVES)
@pytest.mark.parametrize("log_n", [3, 5, 8])
def test_verifier_accepts_oracle_proofs(proved, curve, cid, log_n):
    cv, circ, ck, inf, vk, pub, proof = proved(curve, cid, log_n)
    out = V.verify(cv, vk, pub, proof, TAU, transcript=PlonkTranscript(curve))
    assert set(out["challenges"]) == {"beta", "gamma", "alpha", "zeta", "v", "u"}


@pytest.mark.parametrize("curve,cid", CURVES)
def test_verifier_rejects_tampering(oracle, proved, curve, cid):
    log_n = 5
    cv, circ, ck, inf, vk, pub, proof = proved(curve, cid, log_n)
    ok = V.verify(cv, vk, pub, proof, TAU, transcript=PlonkTranscript(curve))
    fixed = {k: v for k, v in ok["challenges"].items()}
    one = V.fr_limbs(cv, 1)
    bump = lambda x: oracle.field_op(cid, 0, "add", np.asarray(x).reshape(1, 4), one.reshape(1, 4))[0]

    def rejects(p2, vk2=vk, pub2=pub, fs=True):
        # with the verifier's own Fiat-Shamir (every challenge after the tampered element changes) ...
        if fs:
            with pytest.raises(V.VerificationError):
                V.verify(cv, vk2, pub2, p2, TAU, transcript=PlonkTranscript(curve))
        # ... and with the honest run's challenges held fixed (the algebra alone must catch it)
        with pytest.raises(V.VerificationError):
            V.verify(cv, vk2, pub2, p2, TAU, challenges=fixed)

    # one flipped evaluation, each kind
    for key, idx in (("wires_evals", 0), ("wires_evals", 4), ("wire_sigma_evals", 3)):
        ev = [x.copy() for x in proof[key]]
        ev[idx] = bump(ev[idx])
        rejects(dict(proof, **{key: ev}))
    rejects(dict(proof, perm_next_eval=bump(proof["perm_next_eval"])))
    # swapped commitments
    wc = list(proof["wires_poly_comms"]); wc[0], wc[1] = wc[1], wc[0]
    rejects(dict(proof, wires_poly_comms=wc))
    tq = list(proof["split_quot_poly_comms"]); tq[1], tq[2] = tq[2], tq[1]
    rejects(dict(proof, split_quot_poly_comms=tq))
    rejects(dict(proof, opening_proof=proof["shifted_opening_proof"], shifted_opening_proof=proof["opening_proof"]))
    # the selector ORDER matters (dispatcher2.rs:443-456): a key with q_lc[0] <-> q_lc[1], q_o <-> q_c or q_mul[0] <-> q_ecc swapped rejects
    for i, j in ((0, 1), (10, 11), (4, 12)):
        sc = list(vk["selector_comms"]); sc[i], sc[j] = sc[j], sc[i]
        rejects(proof, vk2=dict(vk, selector_comms=sc))
    sg = list(vk["sigma_comms"]); sg[3], sg[4] = sg[4], sg[3]
    rejects(proof, vk2=dict(vk, sigma_comms=sg))
    kk = list(vk["k"]); kk[1], kk[2] = kk[2], kk[1]
    rejects(proof, vk2=dict(vk, k=kk))
    # a different public input
    pub2 = pub.copy(); pub2[1] = bump(pub2[1])
    rejects(proof, pub2=pub2)
    # a wrong trapdoor is a different SRS
    with pytest.raises(V.VerificationError):
        V.verify(cv, vk, pub, proof, TAU + 1, challenges=fixed)
    # a point off the curve
    bad = (proof["opening_proof"][0].copy(), False)
    bad[0][0] ^= np.uint64(1)
    with pytest.raises(V.VerificationError):
        V.verify(cv, vk, pub, dict(proof, opening_proof=bad), TAU, challenges=fixed)


def test_unsatisfied_witness_never_reaches_the_verifier(oracle):
    """A witness with one perturbed wire value violates a gate and a copy constraint: the quotient no longer divides and the
    prover's own degree check (dispatcher2.rs:511-518) fires before any proof exists, with the transcript in the loop."""
    from oracle import prover_ref as P
    curve, cid, log_n = "bn254", 0, 4
    _, cv, circ, ck, inf, vk, bl, pub = make_instance(oracle, curve, cid, log_n, 901)
    wires = circ["wires"].copy()
    wires[0, 3] = oracle.rand_fr(cid, 77, 1)[0]
    with pytest.raises(ValueError, match="WrongQuotientPolyDegree"):
        P.prove_rounds(cid, log_n, ck, inf, dict(circ, wires=wires), bl, ProverSideTranscript(curve, vk, pub), threads=2)


def test_lagrange_public_input_evaluation(oracle):
    """PI(zeta) by the Lagrange formula == Horner evaluation of iNTT(padded inputs) (what the prover interpolates, :426)."""
    cv = B.BN254
    n = 32
    pub = oracle.rand_fr(0, 5, 3)
    padded = np.zeros((n, 4), dtype=np.uint64); padded[:3] = pub
    poly = oracle.ntt(0, padded, True, False)
    zeta = oracle.rand_fr(0, 6, 1)[0]
    want = V.fr_int(cv, oracle.poly_eval(0, poly, zeta))
    assert V.lagrange_pi_eval(cv, n, [V.fr_int(cv, x) for x in pub], V.fr_int(cv, zeta)) == want
