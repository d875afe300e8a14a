"""The N>1 path on CPU: two processes, `gloo` backend, world_size 2.

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


@pytest.mark.parametrize("log_n,cid", [(7, 0), (8, 1)])
def test_two_rank_exchange_and_msm_reduce(log_n, cid):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(rk, 2, port, log_n, cid, q)) for rk in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
