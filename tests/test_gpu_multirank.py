"""The N > 1 path on REAL GPUs: one process per GPU over the in-library RCCL transport (csrc/comm_rccl.hip: grouped ncclSend/ncclRecv
on the context's stream — what replaces the worker<->worker TCP + Cap'n Proto links of /root/reference/src/worker.rs:280-345,412-438
and the `result: Data` replies the dispatcher adds up, dispatcher.rs:218-238).

Self-activating: every case is parametrised on the number of GPUs the box offers.  World size 1 always runs (the same rank
program over a one-rank communicator — it keeps the program itself tested on the 1-GPU boxes gpurun grants); world sizes 2, 4 and 8
run whenever `torch.cuda.device_count()` allows and are skipped, cleanly, otherwise.  Each rank checks ITS share against the CPU
oracle on the same seeded input; rank 0 collects the verdicts.

  * `RankProver.fft_dev`: all four modes of the reference's 2-D transform (dispatcher2.rs:732-787) and the zero-padding-aware row
    pass (`plonk_fft1_dev_compact`), two contexts / two communicators per rank used alternately like bench.py's two lanes;
  * sharded `commit_many_dev` + one point all-gather (dispatcher2.rs:870-892);
  * `ClassProver` with the commit key sharded over the ranks (dispatcher2.rs:260-266), real transcript, proof == oracle prover's and
    accepted by the trapdoor verifier.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAU = 0x1234567_89ABCDEF_0FEDCBA9_87654321_13579BDF_2468ACE0_0F0F0F0F


def _emulated():
    """tests/test_hostemu.py runs this module's rank programs on CPU: the kernel sources compiled for the host (tests/hostemu), every rank a
    process with its own emulated device, shared memory under the library's comm_* interface in place of RCCL."""
    return os.environ.get("PLONK_ALLOW_HOSTEMU") == "1"


def _world_sizes():
    if _emulated():
        have = int(os.environ.get("HIPEMU_DEVICES", "1"))
    else:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return [pytest.param(w, marks=pytest.mark.skipif(have < w, reason=f"needs {w} GPUs, this box has {have}")) for w in (1, 2, 4, 8)]


def _spawn(target, world, *args, timeout=600):
    from conftest import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=target, args=(rk, world, port, q) + args) for rk in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in procs:
            results.append(q.get(timeout=timeout))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0, f"rank process exit code {p.exitcode}"
    results.sort(key=lambda x: x[0])
    for rk, ok, msg in results:
        assert ok, f"rank {rk}: {msg}"
    return results


def _setup(rank, world, port):
    """torch.distributed (gloo) only hands the 128-byte RCCL ids around and gathers small host objects; data moves through the
    library's own communicators."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    if not _emulated():
        torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _join(dist, workers, rank, world):
    from distributed_plonk_amd.worker import PlonkWorker
    ids = [PlonkWorker.comm_unique_id() for _ in workers] if rank == 0 else [None] * len(workers)
    dist.broadcast_object_list(ids, src=0)
    for w, uid in zip(workers, ids):                      # same creation order on every rank
        w.comm_init(uid, rank, world)
        assert w.comm_info()[:2] == (rank, world)


# ------------------------------------------------------------------------------------------------ distributed transform + sharded MSM
def _fft_rank(rank, world, port, q, curve, cid, log_n):
    try:
        dist = _setup(rank, world, port)
        from distributed_plonk_amd.dispatcher import RankProver, make_fft_workloads, split_rc
        from distributed_plonk_amd.worker import PlonkWorker
        from oracle import oracle as O
        n, m = 1 << log_n, 8 << log_n
        workers = [PlonkWorker(me=rank, device=rank, curve=curve) for _ in range(2)]
        try:
            _join(dist, workers, rank, world)
            bases_all = O.gen_bases(cid, 31, min(64, n), n)
            lo, hi = rank * n // world, (rank + 1) * n // world
            for w in workers:
                w.init(bases_all[lo:hi], n, m)                      # this rank's SRS shard (dispatcher2.rs:260-266)
            lanes = [RankProver(w, rank, world, transport="rccl") for w in workers]
            msgs = []
            call = 0
            for N, is_quot in ((n, False), (m, True)):
                r, c = split_rc(N)
                wl = make_fft_workloads(N, world)[rank]
                v = O.rand_fr(cid, 900 + log_n + int(is_quot), N)
                rows = np.ascontiguousarray(v.reshape(c, r, 4).transpose(1, 0, 2)[wl.row_start:wl.row_end])
                for is_inv, is_coset in ((False, False), (True, False), (False, True), (True, True)):
                    w = workers[call % 2]                           # alternate the two contexts / communicators like bench.py's lanes
                    d_rows = w.alloc(rows.nbytes).upload(rows)
                    d_out = w.alloc(r * wl.num_cols() * 32)
                    lanes[call % 2].fft_dev(d_rows.ptr, d_out.ptr, N, is_quot, is_inv, is_coset, out_layout=1)
                    w.sync()
                    got = d_out.download((r, wl.num_cols(), 4))
                    want = O.ntt(cid, v, is_inv, is_coset, threads=4).reshape(r, c, 4)[:, wl.col_start:wl.col_end]
                    if not np.array_equal(got, want):
                        msgs.append(f"fft N=2^{N.bit_length() - 1} inv={is_inv} coset={is_coset}")
                    d_rows.free(); d_out.free()
                    call += 1
            # zero-padded 8n coset FFT from compact rows
            r, c = split_rc(m)
            wl = make_fft_workloads(m, world)[rank]
            length = n + 3
            L = (length + r - 1) // r
            coeffs = O.rand_fr(cid, 77, length)
            v = np.zeros((m, 4), dtype=np.uint64)
            v[:length] = coeffs
            rows = np.ascontiguousarray(v.reshape(c, r, 4).transpose(1, 0, 2)[wl.row_start:wl.row_end, :L])
            w = workers[0]
            d_rows = w.alloc(rows.nbytes).upload(rows)
            d_out = w.alloc(r * wl.num_cols() * 32)
            lanes[0].fft_dev(d_rows.ptr, d_out.ptr, m, True, False, True, out_layout=1, row_len=L)
            w.sync()
            want = O.ntt(cid, v, False, True, threads=4).reshape(r, c, 4)[:, wl.col_start:wl.col_end]
            if not np.array_equal(d_out.download((r, wl.num_cols(), 4)), want):
                msgs.append("zero-padded row pass")
            d_rows.free(); d_out.free()
            # a round of three sharded commitments of ragged length: partial points through ONE all-gather, reduce on the host
            polys = [O.rand_fr(cid, 300 + j, ln) for j, ln in enumerate((n, n - 5, max(n // 2, 1)))]
            items, bufs = [], []
            for pl in polys:
                mine = pl[lo:min(hi, len(pl))] if len(pl) > lo else pl[:0]
                b = w.alloc(max(mine.nbytes, 32))
                if len(mine):
                    b.upload(mine)
                bufs.append(b)
                items.append((b.ptr, len(mine)))
            parts = w.commit_many_dev(items)
            gathered = w.comm_allgather_host(parts, world)                      # (world, 3, 3Q)
            for j, pl in enumerate(polys):
                acc = None
                for rk in range(world):
                    acc = gathered[rk][j] if acc is None else w.g1_add(acc, gathered[rk][j])
                got = w.g1_to_affine(acc)
                want = O.jac_to_affine(cid, O.commit_polynomial(cid, bases_all, pl, threads=4))
                if not (got[1] == want[1] and np.array_equal(got[0], want[0])):
                    msgs.append(f"sharded commitment {j}")
            for b in bufs:
                b.free()
            q.put((rank, not msgs, "; ".join(msgs)))
        finally:
            for w in workers:
                w.comm_destroy()
                w.close()
            dist.destroy_process_group()
    except BaseException as ex:      # noqa: BLE001 - reported through the queue
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:]))
        raise


@pytest.mark.parametrize("world", _world_sizes())
@pytest.mark.parametrize("curve,cid,log_n", [("bn254", 0, 9), ("bls12_381", 1, 8)])
def test_distributed_transform_and_sharded_commit_over_rccl(world, curve, cid, log_n):
    _spawn(_fft_rank, world, curve, cid, log_n)


# ------------------------------------------------------------------------------------------------ the whole class prover
def _class_rank(rank, world, port, q, curve, cid, log_n):
    try:
        dist = _setup(rank, world, port)
        from distributed_plonk_amd.class_prover import ClassProver, LibComm, key_shard_range
        from distributed_plonk_amd.transcript import PlonkTranscript
        from distributed_plonk_amd.worker import PlonkWorker
        from oracle import bigint_ref as B
        from oracle import oracle as O
        from oracle import prover_ref as P
        from oracle import verifier_ref as V
        n = 1 << log_n
        workers = [PlonkWorker(me=rank, device=rank, curve=curve) for _ in range(2)]
        try:
            _join(dist, workers, rank, world)
            circ = P.make_circuit(cid, log_n, seed=71, num_inputs=2)           # same instance on every rank
            ck, inf = P.make_ck_trapdoor(cid, n, TAU)
            klo, khi = key_shard_range(len(ck), rank, world)
            for w in workers:
                w.init(ck[klo:khi], n, 8 * n)

            def boot(obj):
                out = [None] * world
                dist.all_gather_object(out, obj)
                return out

            pv = ClassProver(workers[0], log_n, LibComm(workers[0], bootstrap=boot), commit_helper=workers[1], key_range=(klo, khi))
            try:
                pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
                pub = circ["pub_input"][:2]
                bl = dict(wires=O.rand_fr(cid, 5, 10).reshape(5, 2, 4), perm=O.rand_fr(cid, 6, 3))
                fs = pv.fiat_shamir(pub)
                got = pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, fs)
                vk = pv.verifying_key()
            finally:
                pv.close()
            msgs = []
            want = P.prove_rounds(cid, log_n, ck, inf, circ, bl, fs.drawn, threads=4)
            same = lambda a, b: a[1] == b[1] and np.array_equal(a[0], b[0])
            for key in ("wires_poly_comms", "split_quot_poly_comms"):
                if not all(same(g, x) for g, x in zip(got[key], want[key])):
                    msgs.append(key)
            for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
                if not same(got[key], want[key]):
                    msgs.append(key)
            for key in ("wires_evals", "wire_sigma_evals"):
                if not np.array_equal(np.stack(got[key]), np.stack(want[key])):
                    msgs.append(key)
            if rank == 0:
                try:
                    V.verify(B.CURVES[curve], vk, pub, got, TAU, transcript=PlonkTranscript(curve))
                except V.VerificationError as ex:
                    msgs.append(f"verifier: {ex}")
            q.put((rank, not msgs, "; ".join(msgs)))
        finally:
            for w in workers:
                w.comm_destroy()
                w.close()
            dist.destroy_process_group()
    except BaseException as ex:      # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:]))
        raise


@pytest.mark.parametrize("world", _world_sizes())
@pytest.mark.parametrize("curve,cid,log_n", [("bn254", 0, 8), ("bls12_381", 1, 6)])
def test_class_prover_with_sharded_key_over_rccl(world, curve, cid, log_n):
    _spawn(_class_rank, world, curve, cid, log_n)


# ------------------------------------------------------------------------------------------------ collectives entered in different orders
def _order_rank(rank, world, port, q):
    """Two contexts (two communicators, created in the same order everywhere) per rank.  With the order check on, a collective that the ranks
    enter on DIFFERENT communicators — the cross-communicator deadlock of comm_rccl.hip's header — is PLONK_ERR_STATE on every rank."""
    try:
        dist = _setup(rank, world, port)
        from distributed_plonk_amd._ffi import PlonkError
        from distributed_plonk_amd.worker import PlonkWorker
        a, b = (PlonkWorker(me=rank, device=rank, curve="bn254") for _ in range(2))
        try:
            _join(dist, [a, b], rank, world)
            a.set_option("comm_check_order", 1)                       # process-wide; every rank sets it
            msgs = []
            x = np.arange(12, dtype=np.uint64) + 100 * rank

            def gathered_ok(w):
                got = w.comm_allgather_host(x, world)
                return all(np.array_equal(got[r], np.arange(12, dtype=np.uint64) + 100 * r) for r in range(world))

            if not (gathered_ok(a) and gathered_ok(b)):               # same order everywhere: the check is transparent
                msgs.append("ordered collectives")
            if world > 1:
                first = a if rank % 2 == 0 else b                     # even ranks enter A's collective, odd ranks B's
                try:
                    first.comm_allgather_host(x, world)
                    msgs.append("collectives in different orders were not refused")
                except PlonkError as ex:
                    if ex.code != -4 or "different orders" not in str(ex):
                        msgs.append(f"wrong error: {ex}")
                # a device all-gather against a host one on the SAME communicator is a mismatch too (kind and bytes are part of the tag)
                d_s, d_r = a.alloc(64), a.alloc(64 * world)
                try:
                    if rank == 0:
                        a.comm_allgather_dev(d_s.ptr, d_r.ptr, 64)
                    else:
                        a.comm_allgather_host(x, world)
                    msgs.append("different collectives on one communicator were not refused")
                except PlonkError as ex:
                    if ex.code != -4:
                        msgs.append(f"wrong error: {ex}")
                d_s.free(); d_r.free()
                dist.barrier()
                if not (gathered_ok(a) and gathered_ok(b)):           # nothing was left half-issued: the communicators still work
                    msgs.append("collectives after a refusal")
            q.put((rank, not msgs, "; ".join(msgs)))
        finally:
            for w in (a, b):
                w.comm_destroy()
                w.close()
            dist.destroy_process_group()
    except BaseException:      # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:]))
        raise


@pytest.mark.parametrize("world", _world_sizes())
def test_collectives_entered_in_different_orders_are_refused_not_deadlocked(world):
    """the first run on more than one GPU must diagnose itself.  PLONK_COMM_CHECK_ORDER=1 / option "comm_check_order":
    before every collective the ranks compare (communicator ordinal, kind, bytes, count) over the device's first communicator."""
    _spawn(_order_rank, world, timeout=300)
