"""The N>1 path on CPU: separate processes over the `gloo` backend, world sizes 2, 4 and 8 (8 = the size of the driver's SCALE run).

What runs here is the product's rank-level plumbing (distributed_plonk_amd.dispatcher: the workload
partition, ONE all-to-all for the NTT exchange, the all-gather + add for MSM partials); the field / curve
arithmetic of each rank is supplied by the oracle — there is no GPU in this container and the product has
no CPU compute path.  The same functions run over RCCL on MI355X (tests/test_gpu_distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, log_n, cid, result_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from distributed_plonk_amd.dispatcher import (all_to_all_blocks, gather_points, make_fft_workloads,
                                                       make_msm_workloads, split_rc)
        from oracle import oracle as O
        N = 1 << log_n
        r, c = split_rc(N)
        wl = make_fft_workloads(N, world)
        me = wl[rank]
        coeffs = O.rand_fr(cid, 2024, N)                                  # every rank derives the same input
        rows = coeffs.reshape(c, r, 4).transpose(1, 0, 2)[me.row_start:me.row_end].copy()   # my decimated rows
        ok = True
        for is_inv, is_coset in [(False, False), (True, True), (False, True)]:
            # fft1 on my rows (worker.rs:235-278)
            mine = np.stack([O.fft1_helper(cid, rows[j], j + me.row_start, log_n, is_inv, is_coset) for j in range(me.num_rows())])
            # pack per peer (worker.rs:327-330) -> ONE all-to-all -> scatter-transpose (worker.rs:432-435)
            send = np.concatenate([O.exchange_pack(mine, w.col_start, w.col_end) for w in wl])
            send_t = torch.from_numpy(send.view(np.int64).reshape(-1).copy())
            recv_t = torch.empty_like(send_t)
            all_to_all_blocks(send_t, recv_t)
            recv = recv_t.numpy().view(np.uint64).reshape(world, -1, 4)
            cols = np.zeros((me.num_cols(), r, 4), dtype=np.uint64)
            for src in range(world):
                O.exchange_scatter(cols, wl[src].row_start, recv[src])
            # fft2 on my columns (worker.rs:347-381)
            out = np.stack([O.fft2_helper(cid, cols[i], i + me.col_start, log_n, is_inv, is_coset) for i in range(me.num_cols())])
            want = O.ntt(cid, coeffs, is_inv, is_coset).reshape(r, c, 4)   # natural order X[j*c + i]
            ok &= bool(np.array_equal(out, want[:, me.col_start:me.col_end].transpose(1, 0, 2)))
        # sharded MSM: my index range + gather + reduce (dispatcher.rs:218-238)
        n = 1 << 8
        bases = O.gen_bases(cid, 11, 32, n)
        sc = O.from_mont(cid, O.rand_fr(cid, 12, n))
        mw = make_msm_workloads(n, world)[rank]
        part = O.msm(cid, bases[mw.start:mw.end], sc[mw.start:mw.end])
        acc = None
        for p in gather_points(part):
            acc = p if acc is None else O.jac_add(cid, acc, p)
        got, gi = O.jac_to_affine(cid, acc)
        exp, ei = O.jac_to_affine(cid, O.msm(cid, bases, sc))
        ok &= bool(gi == ei and np.array_equal(got, exp))
        result_q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,cid", [(2, 7, 0), (2, 8, 1), (8, 8, 0)])          # 8 ranks: the SCALE run's world size (r = c = 16: 2 rows per rank)
def test_two_rank_exchange_and_msm_reduce(world, log_n, cid):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(rk, world, port, log_n, cid, q)) for rk in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(rk, True) for rk in range(world)]


# --------------------------------------------------------------------------------------------- coset-class decomposition (class_prover.py)
class _HostMem:
    """Stands in for PlonkWorker's device memory on a box without a GPU: "pointers" are byte offsets into one numpy arena."""

    def __init__(self, nbytes):
        self.arena = np.zeros(nbytes // 8, dtype=np.int64)

    def read_bytes(self, src, nbytes):
        return self.arena[src // 8:(src + nbytes) // 8].copy()

    def write_bytes(self, dst, arr):
        a = np.ascontiguousarray(arr).view(np.int64).reshape(-1)
        self.arena[dst // 8:dst // 8 + a.size] = a


def _class_rank_main(rank, world, port, log_m, cid, result_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from distributed_plonk_amd.class_prover import TorchComm, shard_range
        from oracle import oracle as O
        from oracle import prover_ref as P
        f = P.CURVE_OBJ[cid].fr
        m, G, s = 1 << log_m, world, rank
        mL = m // G
        evals = O.rand_fr(cid, 4242, m)                                   # the quotient's coset evaluations (same on every rank)
        mine = np.ascontiguousarray(evals[s::G])                          # my class
        # my additive contribution to every coefficient: (1/G) h_s^-i * iNTT_mL(mine)[i mod mL], h_s = g w_m^s
        E = O.ntt(cid, mine, True, False)
        w_m = P.fr_from_limbs(f, O.field_const(cid, 0, 4, log_m))
        h_inv = pow(f.generator * pow(w_m, s, f.p), -1, f.p)
        pw = np.zeros((m, 4), dtype=np.uint64)
        pw[0] = P.fr_to_limbs(f, pow(G, -1, f.p))
        filled = 1
        while filled < m:
            step = np.broadcast_to(P.fr_to_limbs(f, pow(h_inv, filled, f.p)), (filled, 4)).copy()
            pw[filled:2 * filled] = O.field_op(cid, 0, "mul", pw[:filled], step)
            filled *= 2
        contrib = O.field_op(cid, 0, "mul", np.tile(E, (G, 1)), pw)
        # the product's transport: ONE all-to-all (block r -> rank r), local sum, ONE all-gather
        mem = _HostMem(4 * m * 32)
        comm = TorchComm(mem, device=None)
        d_contrib, d_recv, d_mine, d_quot = 0, m * 32, 2 * m * 32, 3 * m * 32
        mem.write_bytes(d_contrib, contrib)
        comm.all_to_all_dev(d_contrib, d_recv, mL * 32)
        recv = mem.read_bytes(d_recv, m * 32).view(np.uint64).reshape(G, mL, 4)
        acc = recv[0]
        for p in range(1, G):
            acc = O.field_op(cid, 0, "add", acc, recv[p])
        mem.write_bytes(d_mine, acc)
        comm.all_gather_dev(d_mine, d_quot, mL * 32)
        quot = mem.read_bytes(d_quot, m * 32).view(np.uint64).reshape(m, 4)
        ok = bool(np.array_equal(quot, O.ntt(cid, evals, True, True)))    # quot_domain.coset_ifft, dispatcher2.rs:507
        # partial commitments: my coefficient shard, 96/144-byte all-gather, host add
        n = 1 << 7
        bases = O.gen_bases(cid, 11, 32, n)
        coeffs = O.rand_fr(cid, 13, n - 3)
        lo, hi = shard_range(coeffs.shape[0], rank, world)
        part = O.commit_polynomial(cid, bases[lo:hi], coeffs[lo:hi])
        acc = None
        for pt in comm.all_gather_host(part):
            acc = pt if acc is None else O.jac_add(cid, acc, pt)
        got, gi = O.jac_to_affine(cid, acc)
        exp, ei = O.jac_to_affine(cid, O.commit_polynomial(cid, bases, coeffs))
        ok &= bool(gi == ei and np.array_equal(got, exp))
        result_q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_m,cid", [(2, 8, 0), (2, 9, 1), (8, 9, 0)])
def test_two_rank_coset_class_exchange(world, log_m, cid):
    """class_prover.TorchComm over gloo, world size 2: per-class interpolation contributions -> all-to-all -> sum -> all-gather
    reproduces the whole-domain coset iFFT; sharded commitments add up."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_class_rank_main, args=(rk, world, port, log_m, cid, q)) for rk in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(rk, True) for rk in range(world)]


# --------------------------------------------------------------------------------------------- the whole class prover, SPMD over gloo
def _prover_rank_main(rank, world, port, log_n, cid, result_q, shard_key=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_worker import CpuWorker
        from distributed_plonk_amd.class_prover import ClassProver, TorchComm, key_shard_range
        from oracle import oracle as O
        from oracle import prover_ref as P
        curve = {0: "bn254", 1: "bls12_381"}[cid]
        n = 1 << log_n
        circ = P.make_circuit(cid, log_n, seed=77)                      # identical on every rank
        ck, inf = P.make_ck(cid, n, seed=78, unique=8)
        bl = dict(wires=O.rand_fr(cid, 1, 10).reshape(5, 2, 4), perm=O.rand_fr(cid, 2, 3))
        key_range = key_shard_range(len(ck), rank, world) if shard_key else None      # SRS sharded like dispatcher2.rs:260-266
        my_ck = ck[key_range[0]:key_range[1]] if shard_key else ck
        w = CpuWorker(curve, me=rank)
        w.init(my_ck, n, 8 * n)
        helper = None
        if rank % 2 == 0:                                               # half of the ranks use two commitment lanes
            helper = CpuWorker(curve, me=rank, share=w)
            helper.init(my_ck, n, 8 * n)
        pv = ClassProver(w, log_n, TorchComm(w, device=None), commit_helper=helper, key_range=key_range)
        pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
        fs = pv.fiat_shamir(circ["pub_input"][:2])                      # every rank runs its own transcript
        got = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, fs, keep=True)
        want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, fs.drawn)
        same = lambda a, b: a[1] == b[1] and np.array_equal(a[0], b[0])
        ok = all(same(g, x) for g, x in zip(got["wires_poly_comms"], want["wires_poly_comms"]))
        ok &= all(same(g, x) for g, x in zip(got["split_quot_poly_comms"], want["split_quot_poly_comms"]))
        ok &= same(got["prod_perm_poly_comm"], want["prod_perm_poly_comm"])
        ok &= same(got["opening_proof"], want["opening_proof"]) and same(got["shifted_opening_proof"], want["shifted_opening_proof"])
        ok &= bool(np.array_equal(np.stack(got["wires_evals"]), np.stack(want["wires_evals"])))
        ok &= bool(np.array_equal(got["_debug"]["quot_poly"], want["quot_poly"]))
        result_q.put((rank, bool(ok), fs.drawn["v"].tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,cid,shard_key", [(2, 4, 0, False), (4, 5, 1, True), (2, 5, 0, True), (8, 5, 0, True)])     # 8: the SCALE run's world size
def test_class_prover_spmd_over_gloo(world, log_n, cid, shard_key):
    """distributed_plonk_amd.class_prover.ClassProver end to end as `world` processes over gloo: the product's SPMD orchestration
    and torch.distributed calls run for real (host-staged tensors); each rank's device work is done by the oracle through the
    test-only stand-in tests/cpu_worker.py.  Every rank must produce the oracle prover's proof and draw the same challenges."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prover_rank_main, args=(rk, world, port, log_n, cid, q, shard_key)) for rk in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    assert all(r[1] for r in results)
    assert len({r[2] for r in results}) == 1            # identical transcripts on all ranks
