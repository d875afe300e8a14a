"""The reference's distributed call sequence (fftInit / fft1 / fft2Prepare+fftExchange / fft2, varMsm)
on S in-process workers sharing cuda:0 — the shape of dispatcher.rs:246-350 (`test_fft`: domains 2^11 and
2^13, all four modes) and dispatcher2.rs:1088-1216 (128 / 1024), checked bit-for-bit against the oracle."""
import numpy as np
import pytest

from distributed_plonk_amd.dispatcher import Dispatcher, make_fft_workloads, split_rc

pytestmark = pytest.mark.gpu

MODES = [(False, False), (True, False), (False, True), (True, True)]


@pytest.fixture(scope="module")
def workers4():
    from distributed_plonk_amd.worker import PlonkWorker
    ws = {c: [PlonkWorker(me=i, device=0, curve=c) for i in range(4)] for c in ("bn254", "bls12_381")}
    yield ws
    for l in ws.values():
        for w in l:
            w.close()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("S", [1, 2, 4])
def test_fft_like_reference_test_fft(workers4, oracle, curve, cid, S):
    d = Dispatcher(workers4[curve][:S])
    d.init(None, 1 << 11, 1 << 13)
    for is_quot in (False, True):
        N = d.quot_domain_size if is_quot else d.domain_size
        coeffs = oracle.rand_fr(cid, 31 + is_quot, N)
        for is_inv, is_coset in MODES:
            got = d.fft(coeffs, is_quot, is_inv, is_coset)
            want = oracle.ntt(cid, coeffs, is_inv, is_coset, threads=4)
            assert np.array_equal(got, want), f"{curve} S={S} quot={is_quot} inv={is_inv} coset={is_coset}"


@pytest.mark.parametrize("sizes", [(128, 1024), (1 << 16, 1 << 19)])
def test_fft_other_sizes_and_short_input(workers4, oracle, sizes):
    """dispatcher2.rs:1088-1216 sizes, a 2-pass row/column case, and the zero-padding of :746."""
    d = Dispatcher(workers4["bn254"][:2])
    d.init(None, sizes[0], sizes[1])
    for is_quot in (False, True):
        N = sizes[1] if is_quot else sizes[0]
        coeffs = oracle.rand_fr(0, 5, N // 2 + 3)                 # shorter than the domain
        padded = np.zeros((N, 4), dtype=np.uint64)
        padded[:len(coeffs)] = coeffs
        for is_inv, is_coset in [(False, True), (True, False), (True, True)]:
            assert np.array_equal(d.fft(coeffs, is_quot, is_inv, is_coset), oracle.ntt(0, padded, is_inv, is_coset, threads=8))


def test_per_step_parity_fft1_exchange_fft2(workers4, oracle):
    """Each step against its own restatement: fft1_helper (worker.rs:66-94), the pack/scatter index maps
    (worker.rs:327-330,432-435) and fft2_helper (worker.rs:96-115)."""
    S, log_n, cid = 2, 9, 0
    N = 1 << log_n
    r, c = split_rc(N)
    ws = workers4["bn254"][:S]
    d = Dispatcher(ws)
    d.init(None, N, 0)
    wl = make_fft_workloads(N, S)
    coeffs = oracle.rand_fr(cid, 99, N)
    rows = coeffs.reshape(c, r, 4).transpose(1, 0, 2).copy()
    for is_inv, is_coset in MODES:
        # oracle pipeline
        o_rows = np.stack([oracle.fft1_helper(cid, rows[b], b, log_n, is_inv, is_coset) for b in range(r)])
        o_cols = [np.zeros((wl[s].num_cols(), r, 4), dtype=np.uint64) for s in range(S)]
        for src in range(S):
            for dst in range(S):
                blk = oracle.exchange_pack(o_rows[wl[src].row_start:wl[src].row_end].reshape(-1, c, 4).reshape(wl[src].num_rows(), c * 4).view(np.uint64).reshape(wl[src].num_rows(), c, 4),
                                           wl[dst].col_start, wl[dst].col_end)
                oracle.exchange_scatter(o_cols[dst], wl[src].row_start, blk)
        want = [np.stack([oracle.fft2_helper(cid, o_cols[s][i], i + wl[s].col_start, log_n, is_inv, is_coset)
                          for i in range(wl[s].num_cols())]) for s in range(S)]
        # device pipeline
        id = 1234 + 2 * is_inv + is_coset
        for w in ws:
            w.fft_init(id, wl, False, is_inv, is_coset)
        for s, w in enumerate(ws):
            for j in range(wl[s].num_rows()):
                w.fft1(id, j, rows[wl[s].row_start + j])
        d._fft2_prepare_all(id)
        for s, w in enumerate(ws):
            assert np.array_equal(w.fft2(id, r), want[s]), f"rank {s} inv={is_inv} coset={is_coset}"


def test_call_order_errors(workers4):
    from distributed_plonk_amd._ffi import PlonkError
    w = workers4["bn254"][0]
    w.me = 0
    w.init(None, 1 << 6, 0)
    wl = make_fft_workloads(1 << 6, 1)
    with pytest.raises(PlonkError) as e:
        w.fft1(42, 0, np.zeros((8, 4), dtype=np.uint64))         # unknown id (reference: unwrap panic)
    assert e.value.code == -4
    w.fft_init(43, wl, False, False, False)
    with pytest.raises(PlonkError):
        w.fft1(43, 0, np.zeros((5, 4), dtype=np.uint64))         # wrong row length
    with pytest.raises(PlonkError) as e:
        w.fft2_prepare(43, None)                                 # rows missing
    assert e.value.code == -4
    with pytest.raises(PlonkError):
        w.fft_init(44, wl, True, False, False)                   # quot domain not initialised


def test_commit_polynomial_sharded(workers4, oracle):
    cid = 1
    ws = workers4["bls12_381"][:2]
    d = Dispatcher(ws)
    n = 1 << 9
    bases = oracle.gen_bases(cid, 4, 64, n + 32)
    bases[7] = 0                                                 # infinity in the SRS padding style
    inf = np.zeros(len(bases), dtype=np.uint8); inf[7] = 1
    d.init(bases, n, 0)
    poly = oracle.rand_fr(cid, 6, n + 2)
    xy, isinf = d.commit_polynomial(poly)
    oxy, oinf = oracle.jac_to_affine(cid, oracle.commit_polynomial(cid, bases, poly, inf, threads=4))
    assert isinf == oinf and np.array_equal(xy, oxy)


def test_rank_prover_rccl_single_rank(oracle):
    """The one-process-per-GPU path (RankProver) with the RCCL transport forced on a world of 1:
    HBM-resident rows -> row pass -> torch.distributed.all_to_all_single (backend nccl = RCCL) ->
    column pass, output in the [r][c/S] layout; plus the MSM all-gather + host add."""
    import os
    import torch
    import torch.distributed as dist
    from distributed_plonk_amd.dispatcher import RankProver, gather_points
    from distributed_plonk_amd.worker import PlonkWorker
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    from conftest import free_port
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    w = PlonkWorker(me=0, device=0, curve="bn254")
    try:
        log_n = 13
        N = 1 << log_n
        r, c = split_rc(N)
        w.init(None, N, 0)
        rp = RankProver(w, 0, 1, force_exchange=True)
        coeffs = oracle.rand_fr(0, 17, N)
        rows = np.ascontiguousarray(coeffs.reshape(c, r, 4).transpose(1, 0, 2))      # [r][c]
        for is_inv, is_coset in MODES:
            d_rows = w.alloc(N * 32).upload(rows)
            d_out = w.alloc(N * 32)
            rp.fft_dev(d_rows.ptr, d_out.ptr, N, False, is_inv, is_coset, out_layout=1)
            got = d_out.download((N, 4))                  # [r][c] with (j, i) = X[i + j*c]  == natural order
            assert np.array_equal(got, oracle.ntt(0, coeffs, is_inv, is_coset, threads=4))
            d_rows.free(); d_out.free()
        pts = gather_points(np.arange(12, dtype=np.uint64), None, torch.device("cuda", 0))
        assert len(pts) == 1 and np.array_equal(pts[0], np.arange(12, dtype=np.uint64))
    finally:
        w.close()
        dist.destroy_process_group()


def test_two_contexts_pipelined_transforms(oracle):
    """bench.py's N>1 schedule on one rank: two contexts (two HIP streams) alternate transforms whose exchange goes
    through the RCCL transport, nothing synchronises with the host in between; every result must still be exact."""
    import os
    import torch
    import torch.distributed as dist
    from distributed_plonk_amd.dispatcher import RankProver
    from distributed_plonk_amd.worker import PlonkWorker
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    from conftest import free_port
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    lanes = [PlonkWorker(me=0, device=0, curve="bn254") for _ in range(2)]
    try:
        log_n = 16
        N = 1 << log_n
        r, c = split_rc(N)
        provers = []
        for wk in lanes:
            wk.init(None, N, 0)
            provers.append(RankProver(wk, 0, 1, force_exchange=True))
        K = 6
        coeffs = [oracle.rand_fr(0, 300 + i, N) for i in range(K)]
        ins, outs = [], []
        for i in range(K):
            wk = lanes[i % 2]
            rows = np.ascontiguousarray(coeffs[i].reshape(c, r, 4).transpose(1, 0, 2))
            ins.append(wk.alloc(N * 32).upload(rows))
            outs.append(wk.alloc(N * 32))
        for i in range(K):                      # enqueue everything, no host sync in between
            provers[i % 2].fft_dev(ins[i].ptr, outs[i].ptr, N, False, bool(i & 2), True, out_layout=1)
        for wk in lanes:
            wk.sync()
        for i in range(K):
            want = oracle.ntt(0, coeffs[i], bool(i & 2), True, threads=8)
            assert np.array_equal(outs[i].download((N, 4)), want), i
    finally:
        for wk in lanes:
            wk.close()
        dist.destroy_process_group()


def test_in_library_rccl_transport_single_rank(oracle):
    """The transport INSIDE libplonk_hip.so (plonk_comm_init + grouped ncclSend/ncclRecv on the context's stream, comm_rccl.hip) on a
    world of 1 — no torch, no callback: plonk_fft2_prepare(ctx, id, NULL, NULL) exchanges through RCCL; all four modes, two
    contexts with two communicators pipelined like bench.py's lanes; plus the point all-gather and the raw device collectives."""
    from distributed_plonk_amd._ffi import PlonkError
    from distributed_plonk_amd.dispatcher import RankProver
    from distributed_plonk_amd.worker import PlonkWorker
    lanes = [PlonkWorker(me=0, device=0, curve="bn254") for _ in range(2)]
    try:
        with pytest.raises(PlonkError) as e:
            lanes[0].comm_info()                                   # no communicator yet
        assert e.value.code == -4
        for wk in lanes:                                           # one communicator per context, created in the same order everywhere
            wk.comm_init(PlonkWorker.comm_unique_id(), 0, 1)
        rank, world, ver = lanes[0].comm_info()
        assert (rank, world) == (0, 1) and ver > 0
        with pytest.raises(PlonkError):
            lanes[0].comm_init(PlonkWorker.comm_unique_id(), 0, 1)  # already has one
        log_n = 14
        N = 1 << log_n
        r, c = split_rc(N)
        provers = []
        for wk in lanes:
            wk.init(None, N, 0)
            provers.append(RankProver(wk, 0, 1, transport="rccl"))
        K = 8
        coeffs = [oracle.rand_fr(0, 700 + i, N) for i in range(K)]
        ins, outs = [], []
        for i in range(K):
            wk = lanes[i % 2]
            ins.append(wk.alloc(N * 32).upload(np.ascontiguousarray(coeffs[i].reshape(c, r, 4).transpose(1, 0, 2))))
            outs.append(wk.alloc(N * 32))
        for i in range(K):                                          # enqueue everything, no host sync in between
            is_inv, is_coset = MODES[i % 4]
            provers[i % 2].fft_dev(ins[i].ptr, outs[i].ptr, N, False, is_inv, is_coset, out_layout=1)
        for wk in lanes:
            wk.sync()
        for i in range(K):
            is_inv, is_coset = MODES[i % 4]
            assert np.array_equal(outs[i].download((N, 4)), oracle.ntt(0, coeffs[i], is_inv, is_coset, threads=8)), i
        # varMsm replies: all-gather of host points
        pts = lanes[0].comm_allgather_host(np.arange(36, dtype=np.uint64).reshape(3, 12), 1)
        assert pts.shape == (1, 3, 12) and np.array_equal(pts[0], np.arange(36, dtype=np.uint64).reshape(3, 12))
        # raw device collectives (the class prover's two data-path exchanges)
        a, b = lanes[0].alloc(4096), lanes[0].alloc(4096)
        a.upload(np.arange(512, dtype=np.uint64))
        lanes[0].comm_alltoall_dev(a.ptr, b.ptr, 4096)
        lanes[0].sync()
        assert np.array_equal(b.download((512,)), np.arange(512, dtype=np.uint64))
        lanes[0].memset_dev(b.ptr, 0, 4096)
        lanes[0].comm_allgather_dev(a.ptr, b.ptr, 4096)
        lanes[0].sync()
        assert np.array_equal(b.download((512,)), np.arange(512, dtype=np.uint64))
        # a workload list that disagrees with the communicator is an error, not a hang
        wl = make_fft_workloads(N, 2)
        lanes[0].fft_init(99, wl, False, False, False)
        d = lanes[0].alloc(N * 16)
        lanes[0].fft1_dev(99, d.ptr)
        with pytest.raises(PlonkError):
            lanes[0].fft2_prepare(99, None)
        lanes[0].comm_destroy()
        with pytest.raises(PlonkError):
            lanes[0].comm_info()
    finally:
        for wk in lanes:
            wk.close()


def test_collectives_of_a_device_come_from_one_host_thread():
    """comm_rccl.hip: the per-device total order of collectives only prevents the cross-communicator deadlock if every rank issues them in the
    same order, which two racing host threads cannot promise — the first thread to issue a collective on a device owns its collectives, another
    thread is refused with PLONK_ERR_STATE instead of deadlocking with the peers; the owner is forgotten with the device's last
    communicator."""
    import os
    import threading
    from distributed_plonk_amd._ffi import PlonkError
    from distributed_plonk_amd.worker import PlonkWorker
    if os.environ.get("PLONK_ALLOW_HOSTEMU") == "1":
        pytest.skip("a property of comm_rccl.hip; the emulation's shared-memory communicator (tests/hostemu/comm_local.cpp) has no stream order to protect")
    wk = PlonkWorker(me=0, device=0, curve="bn254")
    try:
        wk.comm_init(PlonkWorker.comm_unique_id(), 0, 1)
        a, b = wk.alloc(4096), wk.alloc(4096)
        a.upload(np.arange(512, dtype=np.uint64))
        wk.comm_alltoall_dev(a.ptr, b.ptr, 4096)                   # this thread now owns device 0's collectives
        wk.sync()
        seen = []

        def other():
            try:
                wk.comm_allgather_dev(a.ptr, b.ptr, 4096)
                seen.append("issued")
            except PlonkError as ex:
                seen.append(ex.code)

        t = threading.Thread(target=other)
        t.start(); t.join()
        assert seen == [-4], seen                                  # PLONK_ERR_STATE, and nothing was enqueued
        wk.comm_allgather_dev(a.ptr, b.ptr, 4096)                  # the owner carries on
        wk.sync()
        assert np.array_equal(b.download((512,)), np.arange(512, dtype=np.uint64))
        wk.comm_destroy()                                          # last communicator of the device: ownership is released ...
        seen.clear()

        def fresh():
            try:
                wk.comm_init(PlonkWorker.comm_unique_id(), 0, 1)
                wk.comm_allgather_dev(a.ptr, b.ptr, 4096)          # ... so another thread may own the next communicator's collectives
                wk.sync()
                seen.append("issued")
            except PlonkError as ex:
                seen.append(ex.code)

        t = threading.Thread(target=fresh)
        t.start(); t.join()
        assert seen == ["issued"], seen
    finally:
        wk.close()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_N,S,length_of", [
    (11, 1, lambda N: N // 8 + 3), (11, 2, lambda N: N // 8 + 3),      # r = 32, c = 64: 9 leading coefficients per row, 8 classes of 8
    (13, 4, lambda N: N // 8 + 3),                                      # odd log: c = 2r
    (16, 2, lambda N: N // 8 + 2),                                      # c = 256: classes of 32 points
    (19, 4, lambda N: N // 8 + 3),                                      # c = 1024, classes of 128
    (22, 2, lambda N: N // 8 + 3),                                      # c = 2048: classes of 256 (single pass each)
    (24, 4, lambda N: N // 8 + 3),                                      # c = 4096: classes of 512
    (27 - 2, 2, lambda N: N // 8 + 3),                                  # c = 8192: two-pass classes (2^10)
    (16, 1, lambda N: N // 2 + 1),                                      # half full: two classes
    (16, 2, lambda N: N - 5),                                           # nearly dense: one class, zero tail only
    (13, 2, lambda N: 7),                                               # almost everything zero: the class cap
])
def test_zero_padded_row_pass_matches_oracle(gpu_workers, oracle, curve, cid, log_N, S, length_of):
    """plonk_fft1_dev_compact: the distributed forward transform of a ZERO-PADDED vector from only the leading coefficients of every
    decimated row (what the reference pads and ships in full, dispatcher2.rs:746-766) == the oracle's transform of the explicitly
    padded vector, plain and coset, for every class count, S in-process workers with the device-to-device block exchange."""
    from distributed_plonk_amd.worker import PlonkWorker
    N = 1 << log_N
    length = length_of(N)
    r, c = split_rc(N)
    row_len = min(c, (length + r - 1) // r)
    coeffs = oracle.rand_fr(cid, 8100 + log_N, length)
    v = np.zeros((N, 4), dtype=np.uint64)
    v[:length] = coeffs
    t = np.ascontiguousarray(v.reshape(c, r, 4).transpose(1, 0, 2))                  # t[b][a] = v[a*r + b]  (dispatcher2.rs:754)
    assert not t[:, row_len:].any()
    workers = [gpu_workers(curve)] + [PlonkWorker(me=i, device=0, curve=curve) for i in range(1, S)]
    try:
        d = Dispatcher(workers)
        d.init(None, N, 0)
        wl = make_fft_workloads(N, S)
        for is_coset in (False, True):
            id = 4242 + int(is_coset)
            keep = []
            for s, w in enumerate(workers):
                w.fft_init(id, wl, False, False, is_coset)
                rows = np.ascontiguousarray(t[wl[s].row_start:wl[s].row_end, :row_len])
                buf = w.alloc(rows.nbytes).upload(rows)
                keep.append((buf, rows))
                w.fft1_dev_compact(id, buf.ptr, row_len)
            d._fft2_prepare_all(id)
            u = np.empty((c, r, 4), dtype=np.uint64)
            for s, w in enumerate(workers):
                u[wl[s].col_start:wl[s].col_end] = w.fft2(id, r)
            got = np.ascontiguousarray(u.transpose(1, 0, 2)).reshape(-1, 4)          # :786
            want = oracle.ntt(cid, v, False, is_coset, threads=32)
            assert np.array_equal(got, want), (is_coset, S)
            for (buf, rows), w in zip(keep, workers):
                assert np.array_equal(buf.download(rows.shape), rows)                # the compact rows are not modified
                buf.free()
        with pytest.raises(Exception):                                               # inverse transforms take dense rows
            workers[0].fft_init(777, wl, False, True, False)
            b = workers[0].alloc(64)
            workers[0].fft1_dev_compact(777, b.ptr, 1)
    finally:
        for w in workers[1:]:
            w.close()


def test_exchange_standin_and_async_copy(gpu_workers):
    """The diagnostic exchange of bench.py --simulate-ranks (plonk_exchange_standin: blocks 1 .. S-1 of `send` -> the same blocks of `recv`, device to
    device on the given stream — the bytes that would cross the fabric; block 0, the rank's own, stays) and plonk_memcpy_d2d_async, both ordered on
    the context's stream."""
    import ctypes as C
    from distributed_plonk_amd import _ffi
    w = gpu_workers("bn254")
    S, b = 4, 4096
    src = np.arange(S * b // 8, dtype=np.uint64)
    old = np.full(S * b // 8, 0xABCD, dtype=np.uint64)
    d_s, d_r = w.alloc(S * b).upload(src), w.alloc(S * b).upload(old)
    _ffi.check(w.lib.plonk_exchange_standin(None, d_s.ptr, d_r.ptr, b, S, w.stream_ptr()))
    w.sync()
    got = d_r.download((S * b // 8,))
    assert np.array_equal(got[b // 8:], src[b // 8:]) and np.array_equal(got[:b // 8], old[:b // 8])
    fn = C.cast(w.lib.plonk_exchange_standin, _ffi.EXCHANGE_FN)             # usable wherever a plonk_exchange_fn is taken (worker.fft2_prepare passes it through)
    assert isinstance(fn, _ffi.EXCHANGE_FN) and fn(None, d_s.ptr, d_r.ptr, b, 1, w.stream_ptr()) == 0       # one rank: nothing moves
    w.memcpy_d2d_async(d_r.ptr, d_s.ptr + b, b)
    w.sync()
    assert np.array_equal(d_r.download((b // 8,)), src[b // 8:2 * b // 8])
    d_s.free(); d_r.free()


def test_trim_releases_the_caches_and_everything_still_works(gpu_workers, oracle):
    """plonk_trim: the pooled exchange buffers of finished tasks, the NTT factor planes / class tables, the MSM workspace and the scratch of a
    context go back to the device; the next calls rebuild what they need and give the same bits (a worker's State outlives a circuit,
    worker.rs:42-59)."""
    from distributed_plonk_amd.dispatcher import Dispatcher
    from distributed_plonk_amd._ffi import MsmWorkload
    w = gpu_workers("bn254")
    n = 1 << 12
    v = oracle.rand_fr(0, 91, n)
    bases = oracle.gen_bases(0, 92, 64, n)
    sc = oracle.from_mont(0, oracle.rand_fr(0, 93, n))
    d = Dispatcher([w])
    d.init(bases, n, 8 * n)
    want_fft = oracle.ntt(0, v, True, True)
    want_big = oracle.ntt(0, oracle.rand_fr(0, 94, 8 * n), False, True)
    want_msm = oracle.jac_to_affine(0, oracle.msm(0, bases, sc, threads=4))
    for _ in range(2):
        assert np.array_equal(d.fft(v, is_quot=False, is_inv=True, is_coset=True), want_fft)
        assert np.array_equal(w.ntt(oracle.rand_fr(0, 94, 8 * n), False, True), want_big)
        got = w.g1_to_affine(w.var_msm(MsmWorkload(0, n), sc))
        assert got[1] == want_msm[1] and np.array_equal(got[0], want_msm[0])
        w.trim()
        w.trim()                                    # idempotent
