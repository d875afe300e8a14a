"""Rounds 1-5 of the reference prover (dispatcher2.rs:296-712) with every vector on the device, against the oracle's
restatement of the same rounds on the same satisfied circuit, challenges and blinders: every commitment (affine limbs),
every evaluation and the intermediate polynomials must be bit-identical."""
import numpy as np
import pytest

from distributed_plonk_amd.prover import Prover, WrongQuotientPolyDegree

pytestmark = pytest.mark.gpu


def _instance(oracle, cid, log_n, seed):
    from oracle import prover_ref as P
    n = 1 << log_n
    circ = P.make_circuit(cid, log_n, seed=seed)
    ck, inf = P.make_ck(cid, n, seed=seed + 1, unique=min(64, n))
    bl = dict(wires=oracle.rand_fr(cid, seed + 2, 10).reshape(5, 2, 4), perm=oracle.rand_fr(cid, seed + 3, 3))
    ch = {k: oracle.rand_fr(cid, seed + 10 + i, 1)[0] for i, k in enumerate(("beta", "gamma", "alpha", "zeta", "v"))}
    return P, circ, ck, inf, bl, ch


def _same_point(got, want):
    return got[1] == want[1] and np.array_equal(got[0], want[0])


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n,cache", [(3, False), (6, True), (10, False), (12, True)])
def test_prover_rounds_match_oracle(gpu_workers, oracle, curve, cid, log_n, cache):
    P, circ, ck, inf, bl, ch = _instance(oracle, cid, log_n, 40 + log_n)
    n = 1 << log_n
    w = gpu_workers(curve)
    w.init(ck, n, 8 * n)                                   # (0,0) rows = the zero points of dispatcher2.rs:207-208
    pv = Prover(w, log_n, cache_key_cosets=cache)
    try:
        pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
        for _ in range(2 if cache else 1):                 # a second proof reuses the cached key cosets
            got = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label], keep=True)
        want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, ch, threads=16)
        for key in ("wires_poly_comms", "split_quot_poly_comms"):
            assert len(got[key]) == len(want[key]) == 5
            for g, x in zip(got[key], want[key]):
                assert _same_point(g, x), key
        for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
            assert _same_point(got[key], want[key]), key
        for key in ("wires_evals", "wire_sigma_evals"):
            assert np.array_equal(np.stack(got[key]), np.stack(want[key])), key
        assert np.array_equal(got["perm_next_eval"], want["perm_next_eval"])
        for key in ("perm_product", "perm_poly", "quot_poly", "lin_poly", "batch_poly"):
            assert np.array_equal(got["_debug"][key], want[key]), key
        assert set(pv.timings) >= {"round1", "round2", "round3_coset_ffts", "round3_quotient", "round3_commit", "round4", "round5"}
    finally:
        pv.close()


def test_prover_rejects_unsatisfied_circuit(gpu_workers, oracle):
    """dispatcher2.rs:511-518: a witness that violates one gate makes the quotient's degree wrong."""
    P, circ, ck, inf, bl, ch = _instance(oracle, 0, 5, 77)
    n = 32
    w = gpu_workers("bn254")
    w.init(ck, n, 8 * n)
    pv = Prover(w, 5)
    try:
        pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
        wires = circ["wires"].copy()
        wires[4, 7] = oracle.rand_fr(0, 1234, 1)[0]
        with pytest.raises(WrongQuotientPolyDegree):
            pv.prove(wires, circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label])
        with pytest.raises(ValueError, match="WrongQuotientPolyDegree"):
            P.prove_rounds(0, 5, ck, inf, dict(circ, wires=wires), bl, ch)
        # the same prover object still produces a valid proof afterwards (no leaked state)
        pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label])
    finally:
        pv.close()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_prover_with_real_transcript(gpu_workers, oracle, curve, cid):
    """challenges drawn from the merlin transcript exactly where dispatcher2.rs draws them.  The proof
    must be reproducible, and the oracle prover fed with the drawn challenges must agree on every output (so the
    transcript absorbed identical bytes on both sides)."""
    from distributed_plonk_amd.prover import FiatShamir
    log_n = 7
    P, circ, ck, inf, bl, _ = _instance(oracle, cid, log_n, 300)
    n = 1 << log_n
    w = gpu_workers(curve)
    w.init(ck, n, 8 * n)
    pv = Prover(w, log_n)
    try:
        pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
        runs = []
        for _ in range(2):
            fs = pv.fiat_shamir(circ["pub_input"][:2])
            runs.append((pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, fs), fs.drawn))
        (got, ch), (got2, ch2) = runs
        assert set(ch) == {"beta", "gamma", "alpha", "zeta", "v"}
        for k in ch:
            assert np.array_equal(ch[k], ch2[k])
        assert _same_point(got["opening_proof"], got2["opening_proof"])
        # verifying-key commitments the transcript absorbed == oracle commitments of the same polynomials
        vk = pv.verifying_key()
        for j in (0, 12):
            want = oracle.jac_to_affine(cid, oracle.commit_polynomial(cid, ck, circ["selectors"][j], inf=inf, threads=8))
            assert _same_point(vk["selector_comms"][j], want)
        want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, ch, threads=8)
        for key in ("wires_poly_comms", "split_quot_poly_comms"):
            for g, x in zip(got[key], want[key]):
                assert _same_point(g, x), key
        for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
            assert _same_point(got[key], want[key]), key
        assert np.array_equal(np.stack(got["wires_evals"]), np.stack(want["wires_evals"]))
        # a different public input changes every challenge
        fs3 = pv.fiat_shamir(circ["pub_input"][:1])
        assert not np.array_equal(fs3("beta", got), ch["beta"])
    finally:
        pv.close()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n,cache", [(4, False), (8, True), (11, False)])
def test_prover_six_coset_quotient_matches_oracle(gpu_workers, oracle, curve, cid, log_n, cache):
    """quotient_mode="classes6": the quotient polynomial interpolated from 6 of the 8 cosets of H_n in the 8n-point domain
    (6n evaluations instead of 8n).  It is the same polynomial, so every output must still equal the reference's."""
    P, circ, ck, inf, bl, ch = _instance(oracle, cid, log_n, 900 + log_n)
    n = 1 << log_n
    w = gpu_workers(curve)
    w.init(ck, n, 8 * n)
    from distributed_plonk_amd.worker import PlonkWorker
    helper = PlonkWorker(curve=curve) if cache else None   # exercise the two-stream commitments in half of the cases
    if helper is not None:
        helper.init(ck, n, 8 * n)
    pv = Prover(w, log_n, cache_key_cosets=cache, quotient_mode="classes6", commit_helper=helper)
    try:
        pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
        for _ in range(2):
            got = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label], keep=True)
        want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, ch, threads=16)
        for key in ("wires_poly_comms", "split_quot_poly_comms"):
            for g, x in zip(got[key], want[key]):
                assert _same_point(g, x), key
        for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
            assert _same_point(got[key], want[key]), key
        assert np.array_equal(np.stack(got["wires_evals"]), np.stack(want["wires_evals"]))
        for key in ("quot_poly", "lin_poly", "batch_poly"):
            assert np.array_equal(got["_debug"][key], want[key]), key
        # an unsatisfied witness still fails the degree check
        bad = circ["wires"].copy()
        bad[4, 1] = oracle.rand_fr(cid, 4321, 1)[0]
        with pytest.raises(WrongQuotientPolyDegree):
            pv.prove(bad, circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label])
        with pytest.raises(ValueError):
            Prover(w, 3, quotient_mode="classes6")             # n = 8: 5n+7 = 6n-1, the degree check would be vacuous
    finally:
        pv.close()
        if helper is not None:
            helper.close()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n", [5, 11])
def test_prover_key_coset_ffts_on_a_third_context_match_oracle(gpu_workers, oracle, curve, cid, log_n):
    """Prover(fft_helper=...) — what bench.py's proof runs with up to 2^22 gates since round 5 (profiles/r05_opening_measurements.txt): the 18
    proving-key coset FFTs of round 3 issued on a third context by a thread of their own beside rounds 1 and 2.  Same outputs as the oracle's
    restatement of dispatcher2.rs:296-712, bit for bit, twice in a row (work buffers reused); an unsatisfied witness still raises and
    leaves no thread behind."""
    import threading
    from distributed_plonk_amd.worker import PlonkWorker
    P, circ, ck, inf, bl, ch = _instance(oracle, cid, log_n, 1300 + log_n)
    n = 1 << log_n
    w = gpu_workers(curve)
    w.init(ck, n, 8 * n)
    c2, h = PlonkWorker(curve=curve), PlonkWorker(curve=curve)
    for x in (c2, h):
        x.init(ck, n, 8 * n)
    pv = Prover(w, log_n, commit_helper=c2, fft_helper=h)
    try:
        pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
        for _ in range(2):
            got = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label], keep=True)
            assert pv._key_ffts is None
        want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, ch, threads=16)
        for key in ("wires_poly_comms", "split_quot_poly_comms"):
            for g, x in zip(got[key], want[key]):
                assert _same_point(g, x), key
        for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
            assert _same_point(got[key], want[key]), key
        assert np.array_equal(np.stack(got["wires_evals"]), np.stack(want["wires_evals"]))
        for key in ("perm_product", "perm_poly", "quot_poly", "lin_poly", "batch_poly"):
            assert np.array_equal(got["_debug"][key], want[key]), key
        bad = circ["wires"].copy()
        bad[4, 1] = oracle.rand_fr(cid, 4321, 1)[0]
        before = threading.active_count()
        with pytest.raises(WrongQuotientPolyDegree):
            pv.prove(bad, circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label])
        assert pv._key_ffts is None and threading.active_count() == before
    finally:
        pv.close()
        c2.close()
        h.close()
