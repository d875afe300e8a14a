"""The multi-rank prover by coset classes (distributed_plonk_amd/class_prover.py): G ranks run as threads sharing the one
GPU of the test box, exchanging device buffers through the same all-to-all / all-gather calls the torch.distributed
transport makes.  Every rank must end up with the proof the oracle's restatement of dispatcher2.rs:296-712 produces."""
import numpy as np
import pytest

from distributed_plonk_amd.class_prover import ClassProver, TorchComm, key_shard_range, run_local_ranks, shard_range

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


def _check(got, want):
    for key in ("wires_poly_comms", "split_quot_poly_comms"):
        assert len(got[key]) == 5
        for g, x in zip(got[key], want[key]):
            assert _same_point(g, x), key
    for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
        assert _same_point(got[key], want[key]), key
    for key in ("wires_evals", "wire_sigma_evals"):
        assert np.array_equal(np.stack(got[key]), np.stack(want[key])), key
    assert np.array_equal(got["perm_next_eval"], want["perm_next_eval"])
    for key in ("perm_product", "perm_poly", "quot_poly", "lin_poly", "batch_poly"):       # perm_product: the all-gathered slices of the sharded grand product
        assert np.array_equal(got["_debug"][key], want[key]), key


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n,G", [(4, 2), (7, 4), (10, 8)])
def test_class_prover_matches_oracle(oracle, curve, cid, log_n, G):
    P, circ, ck, inf, bl, ch = _instance(oracle, cid, log_n, 600 + log_n)
    n = 1 << log_n

    def rank_main(comm, w):
        w.init(ck, n, 8 * n)                               # whole commit key on every rank
        from distributed_plonk_amd.worker import PlonkWorker
        helper = PlonkWorker(me=comm.rank, device=0, curve=curve) if log_n == 10 else None     # one size with the key's class evaluations on a third context beside rounds 1-2
        if helper is not None:
            helper.init(ck, n, 8 * n)
        pv = ClassProver(w, log_n, comm, cache_key_cosets=(log_n == 7), fft_helper=helper)      # one size with them resident across the two proofs
        try:
            pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
            out = None
            assert not pv.replicated_r12                   # the size-n iFFTs by residue class, the grand product by gate range
            for _ in range(2):                             # second proof reuses the work buffers
                out = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label], keep=True)
            assert pv._key_ffts is None
            return out, dict(pv.timings)
        finally:
            pv.close()
            if helper is not None:
                helper.close()

    results = run_local_ranks(G, rank_main, curve=curve)
    want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, ch, threads=8)
    for got, timings in results:
        _check(got, want)
        assert "round3_exchange" in timings


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n,G", [(5, 2), (8, 4), (9, 8), (5, 4), (4, 8)])     # the last two: polynomials that END before a rank's key slice (empty shards)
def test_class_prover_sharded_commit_key(oracle, curve, cid, log_n, G):
    """The SRS sharded G ways like the reference's (dispatcher2.rs:260-266): rank s holds only bases [s*K/G, (s+1)*K/G) and commits
    the coefficients of every polynomial whose index falls in that slice — incl. polynomials shorter than the key (n, n + 2, n + 3
    coefficients against a key padded to a multiple of 32) and the five split-quotient polynomials.  Same proof as the oracle's."""
    P, circ, ck, inf, bl, ch = _instance(oracle, cid, log_n, 900 + log_n)
    n = 1 << log_n
    K = len(ck)

    def rank_main(comm, w):
        klo, khi = key_shard_range(K, comm.rank, comm.size)
        w.init(ck[klo:khi], n, 8 * n)                      # this rank's slice only
        pv = ClassProver(w, log_n, comm, key_range=(klo, khi))
        try:
            pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
            fs = pv.fiat_shamir(circ["pub_input"][:2])      # verifying-key commitments through the sharded key too
            return pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label], keep=True), fs.drawn
        finally:
            pv.close()

    results = run_local_ranks(G, rank_main, curve=curve)
    want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, ch, threads=8)
    for got, _ in results:
        _check(got, want)
    for _, drawn in results[1:]:
        for k in results[0][1]:
            assert np.array_equal(drawn[k], results[0][1][k])


def test_class_prover_with_transcript_and_bad_witness(oracle):
    """Fiat-Shamir on every rank (identical transcripts because identical commitments) and the degree check on all ranks."""
    from distributed_plonk_amd.prover import WrongQuotientPolyDegree
    log_n, G = 6, 4
    P, circ, ck, inf, bl, _ = _instance(oracle, 0, log_n, 700)
    n = 1 << log_n
    bad = circ["wires"].copy()
    bad[4, 3] = oracle.rand_fr(0, 5, 1)[0]

    def rank_main(comm, w):
        w.init(ck, n, 8 * n)
        pv = ClassProver(w, log_n, comm)
        try:
            pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
            fs = pv.fiat_shamir(circ["pub_input"][:2])
            proof = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, fs)
            raised = False
            try:
                pv.prove(bad, circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, pv.fiat_shamir(circ["pub_input"][:2]))
            except WrongQuotientPolyDegree:
                raised = True
            return proof, fs.drawn, raised
        finally:
            pv.close()

    results = run_local_ranks(G, rank_main)
    drawn0 = results[0][1]
    for proof, drawn, raised in results:
        assert raised
        for k in drawn0:
            assert np.array_equal(drawn[k], drawn0[k])
    want = P.prove_rounds(0, log_n, ck, inf, circ, bl, drawn0, threads=8)
    assert _same_point(results[0][0]["opening_proof"], want["opening_proof"])
    assert _same_point(results[G - 1][0]["shifted_opening_proof"], want["shifted_opening_proof"])


def test_a_failure_in_one_gate_range_is_raised_on_every_rank(oracle):
    """an out-of-range permutation index lives in ONE rank's gate slice.  That rank must not raise before the
    all-gather of the slice totals (its peers would wait in the collective for ever): the status travels with the totals and every rank
    raises after it — no rank is left with a broken barrier or a hang."""
    import threading
    log_n, G = 6, 4
    P, circ, ck, inf, bl, ch = _instance(oracle, 0, log_n, 750)
    n = 1 << log_n
    bad_idx = circ["perm_idx"].copy()
    bad_idx[2] = 5 * n + 9                                   # wire 0, gate 2: rank 0's range only

    def rank_main(comm, w):
        w.init(ck, n, 8 * n)
        pv = ClassProver(w, log_n, comm)
        try:
            pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
            try:
                pv.prove(circ["wires"], circ["id_perm"], bad_idx, circ["pub_input"], bl, lambda label, _: ch[label])
            except threading.BrokenBarrierError:
                raise
            except Exception as ex:     # noqa: BLE001 - what the test is about
                first = (type(ex).__name__, str(ex))
            else:
                first = None
            # and the communicator is still usable: the good circuit proves right after
            good = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label])
            return first, good
        finally:
            pv.close()

    results = run_local_ranks(G, rank_main)
    want = P.prove_rounds(0, log_n, ck, inf, circ, bl, ch, threads=8)
    for r, (first, good) in enumerate(results):
        assert first is not None, f"rank {r} did not see the failure of rank 0's gate range"
        assert _same_point(good["opening_proof"], want["opening_proof"])
    assert "rank(s) [0]" in results[1][0][1] and "rank(s) [0]" in results[G - 1][0][1]


def test_class_prover_over_rccl_single_rank(gpu_workers, oracle):
    """The torch.distributed transport (nccl = RCCL) on the library's stream, world size 1: G = 1 is the degenerate class
    decomposition (one class = the whole coset), exercising all_to_all_single / all_gather_into_tensor / all_gather_object."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    from conftest import free_port
    os.environ["MASTER_PORT"] = str(free_port())
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        log_n = 8
        P, circ, ck, inf, bl, ch = _instance(oracle, 0, log_n, 800)
        n = 1 << log_n
        w = gpu_workers("bn254")
        w.init(ck, n, 8 * n)
        pv = ClassProver(w, log_n, TorchComm(w, torch.device("cuda", 0)))
        try:
            pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
            got = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label], keep=True)
        finally:
            pv.close()
        _check(got, P.prove_rounds(0, log_n, ck, inf, circ, bl, ch, threads=8))
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n,G", [(3, 2), (6, 4), (9, 8), (13, 8)])
def test_size_n_ifft_by_residue_class(gpu_workers, oracle, curve, cid, log_n, G):
    """The building blocks of ClassProver._interpolate_many on one context: for every class s, plonk_coset_eval_dev of the n EVALUATIONS
    (read as coefficients, folded G-fold onto n/G points) at shift w_n^-s, then plonk_class_interleave_dev(reverse, 1/n) over the G
    class vectors == domain.ifft (dispatcher2.rs:300-309) of the oracle, bit for bit; two polynomials side by side exercise the class stride."""
    from distributed_plonk_amd import fr as _fr
    f = _fr.FIELDS[curve]
    w = gpu_workers(curve)
    n = 1 << log_n
    L, K = n // G, 2
    ev = [oracle.rand_fr(cid, 40 + log_n + k, n) for k in range(K)]
    d_ev = [w.alloc(n * 32).upload(e) for e in ev]
    d_all, d_out = w.alloc(G * K * L * 32), w.alloc(n * 32)
    for s in range(G):
        shift = f.to_limbs(pow(f.root_of_unity(n), (n - s) % n, f.p))
        for k in range(K):
            w.coset_eval_dev(d_ev[k].ptr, n, L, shift, d_all.ptr + ((s * K + k) * L) * 32)
    for k in range(K):
        w.class_interleave_dev(d_all.ptr + k * L * 32, G, L, True, f.to_limbs(f.inv(n)), d_out.ptr, in_stride=K * L)
        assert np.array_equal(d_out.download((n, 4)), oracle.ntt(cid, ev[k], True, False)), (k, "residue-class iFFT")
    # without reversal / scale the call is a plain transpose of the class-major matrix
    w.class_interleave_dev(d_all.ptr, G, L, False, None, d_out.ptr, in_stride=K * L)
    cm = d_all.download((G, K, L, 4))[:, 0]
    assert np.array_equal(d_out.download((L, G, 4)), cm.transpose(1, 0, 2))
    for b in d_ev + [d_all, d_out]:
        b.free()


def test_class_interleave_rejects_bad_arguments(gpu_workers):
    from distributed_plonk_amd._ffi import PlonkError
    w = gpu_workers("bn254")
    a, b = w.alloc(64 * 32), w.alloc(64 * 32)
    for kw in (dict(classes=3, size=8), dict(classes=16, size=4), dict(classes=4, size=0), dict(classes=4, size=8, in_stride=4)):
        with pytest.raises(PlonkError):
            w.class_interleave_dev(a.ptr, kw["classes"], kw["size"], False, None, b.ptr, in_stride=kw.get("in_stride", 0))
    with pytest.raises(PlonkError):
        w.class_interleave_dev(a.ptr, 4, 8, False, None, a.ptr)
    with pytest.raises(PlonkError):                                   # nine-fold: beyond NTT_MAX_FOLD
        w.coset_eval_dev(a.ptr, 36, 4, np.array([1, 0, 0, 0], dtype=np.uint64), b.ptr)
    a.free(); b.free()


def test_shard_range_covers_everything():
    for length in (1, 7, 4099, 1 << 20):
        for G in (1, 2, 4, 8):
            edges = [shard_range(length, r, G) for r in range(G)]
            assert edges[0][0] == 0 and edges[-1][1] == length
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))


def test_class_prover_over_in_library_rccl_single_rank(oracle):
    """The same degenerate G = 1 proof through LibComm: the library's own RCCL communicator carries the all-to-all, the all-gather
    of quotient coefficients and the all-gather of partial commitment points — no torch.distributed anywhere."""
    from distributed_plonk_amd.class_prover import LibComm
    from distributed_plonk_amd.worker import PlonkWorker
    log_n = 8
    P, circ, ck, inf, bl, ch = _instance(oracle, 0, log_n, 801)
    n = 1 << log_n
    lanes = [PlonkWorker(me=0, device=0, curve="bn254") for _ in range(2)]
    try:
        lanes[0].comm_init(PlonkWorker.comm_unique_id(), 0, 1)
        for wk in lanes:
            wk.init(ck, n, 8 * n)
        pv = ClassProver(lanes[0], log_n, LibComm(lanes[0]), commit_helper=lanes[1])
        try:
            pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
            got = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label], keep=True)
        finally:
            pv.close()
        _check(got, P.prove_rounds(0, log_n, ck, inf, circ, bl, ch, threads=8))
    finally:
        for wk in lanes:
            wk.close()
