"""The reference's only end-to-end test, on the device prover: prove a satisfied circuit, hand the proof to a verifier
(/root/reference/src/dispatcher2.rs:1273-1295 `test2`; dispatcher.rs:1118-1134).  jf-plonk's pairing verifier is not available;
oracle/verifier_ref.py checks the same equations in G1 for an SRS whose trapdoor the test knows.  The verifier is written from the
protocol in pure Python integers and derives every challenge itself from a fresh merlin transcript, so acceptance means: the 13
commitments, 10 evaluations and both opening proofs the GPU produced satisfy the TurboPlonk identities with the reference's
selector order, permutation argument, quotient split and Fiat-Shamir order — not merely "equal to the builder's restatement".

Also here: the device generators the full-size bench leg relies on (`plonk_synth_srs`, `plonk_synth_circuit`) against the oracle."""
import numpy as np
import pytest

from distributed_plonk_amd.prover import Prover, WrongQuotientPolyDegree
from distributed_plonk_amd.synthetic import SyntheticInstance
from distributed_plonk_amd.transcript import PlonkTranscript

pytestmark = pytest.mark.gpu

CURVES = [("bn254", 0), ("bls12_381", 1)]
TAU = 0x0123456789ABCDEF_FEDCBA9876543210_0F1E2D3C4B5A6978_1122334455667788 >> 3


def _blinders(oracle, cid, seed):
    return dict(wires=oracle.rand_fr(cid, seed, 10).reshape(5, 2, 4), perm=oracle.rand_fr(cid, seed + 1, 3))


@pytest.mark.parametrize("curve,cid", CURVES)
def test_synth_srs_is_the_trapdoor_key(gpu_workers, oracle, curve, cid):
    """plonk_synth_srs: d_out[i] = tau^i * G, against the C oracle's scalar multiplications (powers 0 .. 66 and, for a scalar with
    zero bytes and a large one, single entries)."""
    from oracle import bigint_ref as B
    from oracle import prover_ref as P
    from oracle import verifier_ref as V
    w = gpu_workers(curve)
    cv = B.CURVES[curve]
    q = w.q64
    n = 64
    ck, inf = P.make_ck_trapdoor(cid, n, TAU)                      # n + 3 = 67 powers, padded to 96
    buf = w.alloc(len(ck) * 2 * q * 8)
    try:
        w.memset_dev(buf.ptr, 0, buf.nbytes)
        w.synth_srs(V.fr_limbs(cv, TAU), n + 3, buf.ptr)
        got = buf.download((len(ck), 2 * q))
        assert np.array_equal(got, ck)
        for tau in (1, 256, (1 << 64) + 1, cv.fr.p - 1):           # tau = 1: every power is G; p - 1: alternates G, -G
            w.synth_srs(V.fr_limbs(cv, tau), 5, buf.ptr)
            got = buf.download((5, 2 * q))
            for i in range(5):
                assert V.point_int(cv, (got[i], False)) == B.scalar_mul(cv, pow(tau, i, cv.fr.p), (cv.gx, cv.gy)), (tau, i)
    finally:
        buf.free()


@pytest.mark.parametrize("curve,cid", CURVES)
@pytest.mark.parametrize("log_n", [1, 3, 6, 11])
def test_synth_circuit_is_satisfied(gpu_workers, oracle, curve, cid, log_n):
    """plonk_synth_circuit against the oracle's field arithmetic: the gate equation of dispatcher2.rs:465-477 holds on every gate,
    perm_idx is a permutation whose cycles carry equal wire values, id_perm = k_i * w^j, sigma = id_perm[perm_idx]; the
    coefficient-form key is the oracle's iNTT of the evaluations."""
    from oracle import bigint_ref as B
    w = gpu_workers(curve)
    n = 1 << log_n
    inst = SyntheticInstance(w, log_n, seed=77 + log_n, num_inputs=min(2, n))
    try:
        c = inst.download()
    finally:
        inst.close()
    f = B.CURVES[curve].fr
    op = lambda o, a, b=None: oracle.field_op(cid, 0, o, a, b)
    a, b, c_, d, e = c["wires"]
    s = c["selector_evals"]
    p5 = lambda x: op("mul", op("mul", op("mul", x, x), op("mul", x, x)), x)
    ab, cd = op("mul", a, b), op("mul", c_, d)
    acc = op("add", s[11], c["pub_input"])
    for t, x in ((0, a), (1, b), (2, c_), (3, d), (4, ab), (5, cd), (6, p5(a)), (7, p5(b)), (8, p5(c_)), (9, p5(d)), (12, op("mul", op("mul", ab, cd), e))):
        acc = op("add", acc, op("mul", s[t], x))
    acc = op("sub", acc, op("mul", s[10], e))
    assert not acc.any(), "gate equation violated"
    assert c["pub_input"][:inst.num_inputs].any() and not c["pub_input"][inst.num_inputs:].any()
    perm = c["perm_idx"].astype(np.int64)
    assert np.array_equal(np.sort(perm), np.arange(5 * n))
    flat = c["wires"].reshape(5 * n, 4)
    assert np.array_equal(flat[perm], flat), "copy constraints violated"
    if n >= 8:
        assert (perm // n != np.arange(5 * n) // n).all() and (perm % n != np.arange(5 * n) % n).any()     # hops columns and gates
        assert len(np.unique(flat.view([("l", np.uint64, 4)]))) > n // 2                                     # values are not degenerate
    wn = f.root_of_unity(n)
    k = [f.from_mont(B.from_limbs([int(x) for x in r])) for r in c["k"]]
    for i, j in ((0, 0), (1, 1), (4, n - 1), (2, n // 2)):
        want = f.to_mont(k[i] * pow(wn, j, f.p) % f.p)
        assert B.from_limbs([int(x) for x in c["id_perm"][i * n + j]]) == want
    assert np.array_equal(c["id_perm"][perm], c["sigma_evals"].reshape(5 * n, 4))
    for t in (0, 11, 12):
        assert np.array_equal(c["selectors"][t], oracle.ntt(cid, c["selector_evals"][t], True, False))
    assert np.array_equal(c["sigmas"][4], oracle.ntt(cid, c["sigma_evals"][4], True, False))
    # a different seed is a different circuit
    inst2 = SyntheticInstance(w, log_n, seed=1234, num_inputs=min(2, n))
    try:
        assert not np.array_equal(inst2.d_wires.download((5, n, 4)), c["wires"])
    finally:
        inst2.close()


def _verify(curve, vk, pub, proof, tau):
    from oracle import bigint_ref as B
    from oracle import verifier_ref as V
    return V.verify(B.CURVES[curve], vk, pub, proof, tau, transcript=PlonkTranscript(curve))


@pytest.mark.parametrize("curve,cid", CURVES)
@pytest.mark.parametrize("log_n,mode", [(4, "coset8n"), (5, "classes6"), (9, "coset8n"), (12, "classes6"), (12, "coset8n")])
def test_verifier_accepts_device_proofs_of_synthetic_circuits(gpu_workers, oracle, curve, cid, log_n, mode):
    """Everything on the device: circuit, key, trapdoor SRS, the five rounds with the real transcript.  The verifier gets the
    proof, the GPU-computed verifying key and the public inputs; the vk commitments are ALSO re-derived as f(tau)*G from the
    downloaded polynomials (no MSM involved)."""
    from oracle import bigint_ref as B
    from oracle import verifier_ref as V
    w = gpu_workers(curve)
    cv = B.CURVES[curve]
    inst = SyntheticInstance(w, log_n, seed=10 * log_n + cid, num_inputs=3, tau=TAU)
    pv = Prover(w, log_n, quotient_mode=mode)
    try:
        pv.load_key_dev(inst.sel_ptrs, inst.sig_ptrs, inst.k)
        pub = inst.public_inputs()
        fs = pv.fiat_shamir(pub)
        proof = pv.prove_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, inst.d_pi.ptr, _blinders(oracle, cid, 50 + log_n), fs)
        vk = pv.verifying_key()
        out = _verify(curve, vk, pub, proof, TAU)
        for name in ("beta", "gamma", "alpha", "zeta", "v"):               # prover and verifier ran the same Fiat-Shamir
            assert np.array_equal(out["challenges"][name], fs.drawn[name]), name
        if log_n <= 9:
            host = inst.download()
            for j in (0, 5, 11, 12):
                want = V.commit_by_trapdoor(cv, [V.fr_int(cv, x) for x in host["selectors"][j]], TAU)
                assert V.point_int(cv, vk["selector_comms"][j]) == want
            want = V.commit_by_trapdoor(cv, [V.fr_int(cv, x) for x in host["sigmas"][2]], TAU)
            assert V.point_int(cv, vk["sigma_comms"][2]) == want
            # the proof's wire commitment 0 is f(tau)*G of the polynomial the prover holds
            ptr, ln = pv.last_polys["wire_polys"][0]
            poly = pv._download(ptr, ln)
            assert V.point_int(cv, proof["wires_poly_comms"][0]) == V.commit_by_trapdoor(cv, [V.fr_int(cv, x) for x in poly], TAU)
        # one flipped evaluation must reject
        bad = [x.copy() for x in proof["wires_evals"]]
        bad[2][0] ^= np.uint64(1)
        with pytest.raises(V.VerificationError):
            _verify(curve, vk, pub, dict(proof, wires_evals=bad), TAU)
        # a different public input must reject
        pub2 = pub.copy()
        pub2[0] = oracle.rand_fr(cid, 5, 1)[0]
        with pytest.raises(V.VerificationError):
            _verify(curve, vk, pub2, proof, TAU)
    finally:
        pv.close()
        inst.close()


@pytest.mark.parametrize("curve,cid", CURVES)
def test_device_and_oracle_provers_agree_on_a_synthetic_circuit(gpu_workers, oracle, curve, cid):
    """The device-generated instance downloaded and proved by the oracle's restatement: same proof, bit for bit."""
    from oracle import prover_ref as P
    log_n = 7
    n = 1 << log_n
    w = gpu_workers(curve)
    inst = SyntheticInstance(w, log_n, seed=4242, num_inputs=2, tau=TAU)
    pv = Prover(w, log_n)
    try:
        pv.load_key_dev(inst.sel_ptrs, inst.sig_ptrs, inst.k)
        bl = _blinders(oracle, cid, 9)
        fs = pv.fiat_shamir(inst.public_inputs())
        got = pv.prove_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, inst.d_pi.ptr, bl, fs)
        circ = inst.download()
        ck, inf = P.make_ck_trapdoor(cid, n, TAU)
        want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, fs.drawn, threads=8)
        for key in ("wires_poly_comms", "split_quot_poly_comms"):
            for g, x in zip(got[key], want[key]):
                assert g[1] == x[1] and np.array_equal(g[0], x[0]), key
        for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
            assert got[key][1] == want[key][1] and np.array_equal(got[key][0], want[key][0]), key
        for key in ("wires_evals", "wire_sigma_evals"):
            assert np.array_equal(np.stack(got[key]), np.stack(want[key])), key
        assert np.array_equal(got["perm_next_eval"], want["perm_next_eval"])
    finally:
        pv.close()
        inst.close()


def test_verifier_accepts_device_proof_of_an_oracle_circuit(gpu_workers, oracle):
    """The oracle's circuit generator (gates solved for their output wire, random copy structure) through the device prover under
    the trapdoor key, with a commit helper context (two streams)."""
    from oracle import bigint_ref as B
    from oracle import prover_ref as P
    from oracle import verifier_ref as V
    from distributed_plonk_amd.worker import PlonkWorker
    curve, cid, log_n = "bn254", 0, 8
    n = 1 << log_n
    circ = P.make_circuit(cid, log_n, seed=31, num_inputs=4)
    ck, inf = P.make_ck_trapdoor(cid, n, TAU)
    w = gpu_workers(curve)
    helper = PlonkWorker(me=0, device=0, curve=curve)
    w.init(ck, n, 8 * n)
    helper.init(ck, n, 8 * n)
    pv = Prover(w, log_n, commit_helper=helper)
    try:
        pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
        pub = circ["pub_input"][:4]
        proof = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], _blinders(oracle, cid, 3), pv.fiat_shamir(pub))
        _verify(curve, pv.verifying_key(), pub, proof, TAU)
        with pytest.raises(V.VerificationError):                               # the untouched proof under another trapdoor
            _verify(curve, pv.verifying_key(), pub, proof, TAU + 2)
    finally:
        pv.close()
        helper.close()


def test_unsatisfied_synthetic_witness_fails_the_degree_check(gpu_workers, oracle):
    """dispatcher2.rs:511-518 on the device-generated instance: one overwritten wire value."""
    curve, cid, log_n = "bn254", 0, 6
    w = gpu_workers(curve)
    inst = SyntheticInstance(w, log_n, seed=8, tau=TAU)
    pv = Prover(w, log_n)
    try:
        pv.load_key_dev(inst.sel_ptrs, inst.sig_ptrs, inst.k)
        w.write_bytes(inst.wev[1] + 5 * 32, oracle.rand_fr(cid, 99, 1)[0].view(np.int64))
        with pytest.raises(WrongQuotientPolyDegree):
            pv.prove_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, inst.d_pi.ptr, _blinders(oracle, cid, 1), pv.fiat_shamir(inst.public_inputs()))
    finally:
        pv.close()
        inst.close()


@pytest.mark.parametrize("G,sharded", [(2, False), (4, True), (8, True)])
def test_verifier_accepts_class_prover_proofs(oracle, G, sharded):
    """The multi-rank prover by coset classes (G ranks as threads on one GPU), sharded and replicated key, real transcript on every
    rank: every rank's proof is accepted."""
    from oracle import prover_ref as P
    from distributed_plonk_amd.class_prover import ClassProver, key_shard_range, run_local_ranks
    curve, cid, log_n = "bn254", 0, 9
    n = 1 << log_n
    circ = P.make_circuit(cid, log_n, seed=61, num_inputs=2)
    ck, inf = P.make_ck_trapdoor(cid, n, TAU)
    K = len(ck)
    pub = circ["pub_input"][:2]
    bl = _blinders(oracle, cid, 21)

    def rank_main(comm, w):
        klo, khi = key_shard_range(K, comm.rank, comm.size) if sharded else (0, K)
        w.init(ck[klo:khi], n, 8 * n)
        pv = ClassProver(w, log_n, comm, key_range=(klo, khi) if sharded else None)
        try:
            pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
            proof = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, pv.fiat_shamir(pub))
            return proof, pv.verifying_key()
        finally:
            pv.close()

    results = run_local_ranks(G, rank_main, curve=curve)
    for proof, vk in results[:2] + results[-1:]:
        _verify(curve, vk, pub, proof, TAU)
