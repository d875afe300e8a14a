"""Host-side mirror of the reference dispatcher for the hot path: the RPC client wrappers
(`connect/init/msm/fft_init/fft1/fft2_prepare/fft2`, /root/reference/src/dispatcher.rs:29-175,
src/dispatcher2.rs:945-1086), the distributed-NTT orchestration `Prover::fft`
(dispatcher2.rs:732-787) and the distributed MSM / `commit_polynomial` (dispatcher.rs:218-238,
dispatcher2.rs:835-893).

Two deployments of the same worker API:
  * `Dispatcher`      — S workers inside one process (S contexts, on one or several GPUs); the
                        worker<->worker `fftExchange` is a device-to-device block copy.  This is the
                        shape of the reference's own tests (dispatcher.rs:177-350) without TCP.
  * `RankProver`      — one process per GPU (torchrun), data resident in HBM, the exchange is ONE
                        RCCL all-to-all over xGMI (`torch.distributed.all_to_all_single`); MSM
                        partials are all-gathered (96/144 bytes per rank) and added on the host.
"""
from __future__ import annotations

import random
from typing import List, Optional, Sequence

import numpy as np

from ._ffi import FftWorkload, MsmWorkload
from .worker import PlonkWorker


def split_rc(domain_size: int):
    """r = 1 << (log N >> 1), c = N / r  (worker.rs:143-144, dispatcher2.rs:744-745)."""
    log_n = domain_size.bit_length() - 1
    assert 1 << log_n == domain_size, "domain size must be a power of two"
    r = 1 << (log_n >> 1)
    return r, domain_size // r


def make_fft_workloads(domain_size: int, num_slaves: int) -> List[FftWorkload]:
    """dispatcher2.rs:272-291 / dispatcher.rs:278-285."""
    r, c = split_rc(domain_size)
    return [FftWorkload(i * r // num_slaves, (i + 1) * r // num_slaves, i * c // num_slaves, (i + 1) * c // num_slaves)
            for i in range(num_slaves)]


def make_msm_workloads(n: int, num_slaves: int) -> List[MsmWorkload]:
    """dispatcher.rs:223-226 / dispatcher2.rs:875-878."""
    return [MsmWorkload(i * n // num_slaves, (i + 1) * n // num_slaves) for i in range(num_slaves)]


def decimate_rows(coeffs: np.ndarray, r: int) -> np.ndarray:
    """The host transpose of dispatcher2.rs:754: t[b][a] = coeffs[a*r + b]  ->  (r, c, 4)."""
    n = coeffs.shape[0]
    return np.ascontiguousarray(coeffs.reshape(n // r, r, 4).transpose(1, 0, 2))


def undecimate_cols(u: np.ndarray) -> np.ndarray:
    """dispatcher2.rs:786: out[j*c + i] = u[i][j] for u of shape (c, r, 4)."""
    return np.ascontiguousarray(u.transpose(1, 0, 2)).reshape(-1, 4)


class Dispatcher:
    """S in-process workers behind the reference's dispatcher call sequence."""

    def __init__(self, workers: Sequence[PlonkWorker], seed: int = 0xD15EA5E):
        self.workers = list(workers)
        self.num_slaves = len(self.workers)
        for i, w in enumerate(self.workers):
            w.me = i
        self.domain_size = 0
        self.quot_domain_size = 0
        self.n_bases = 0
        self._rng = random.Random(seed)          # task ids (the reference uses thread_rng, dispatcher2.rs:743)

    # ---- init (dispatcher.rs:50-68,213-216: the full SRS goes to every worker)
    def init(self, bases: Optional[np.ndarray], domain_size: int, quot_domain_size: int):
        for w in self.workers:
            w.init(bases, domain_size, quot_domain_size)
        self.domain_size, self.quot_domain_size = domain_size, quot_domain_size
        self.n_bases = 0 if bases is None else len(bases)

    # ---- distributed MSM (dispatcher.rs:218-238)
    def msm(self, scalars: np.ndarray) -> np.ndarray:
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        acc = None
        for w, wl in zip(self.workers, make_msm_workloads(len(scalars), self.num_slaves)):
            part = w.var_msm(wl, scalars[wl.start:wl.end])
            acc = part if acc is None else w.g1_add(acc, part)          # .reduce(|a, b| a + b)
        return acc

    def commit_polynomial(self, poly_mont: np.ndarray):
        """dispatcher2.rs:835-893 -> (affine xy limbs, is_infinity).  `into_repr` runs on the workers'
        GPUs; the zero padding up to bases.len() is implicit (zero scalars add nothing)."""
        poly_mont = np.ascontiguousarray(poly_mont, dtype=np.uint64)[:self.n_bases]
        w0 = self.workers[0]
        acc = None
        for w, wl in zip(self.workers, make_msm_workloads(self.n_bases, self.num_slaves)):
            chunk = poly_mont[wl.start:min(wl.end, len(poly_mont))]
            if len(chunk) == 0:
                continue
            canon = w.field_op(0, 4, chunk)                                   # s.into_repr()
            part = w.var_msm(MsmWorkload(wl.start, wl.start + len(chunk)), canon)
            acc = part if acc is None else w0.g1_add(acc, part)
        if acc is None:
            acc = w0.var_msm(MsmWorkload(0, 0), np.zeros((0, 4), dtype=np.uint64))
        return w0.g1_to_affine(acc)                                           # Commitment(commitment.into())

    # ---- distributed NTT (dispatcher2.rs:732-787)
    def fft(self, coeffs: np.ndarray, is_quot: bool, is_inv: bool, is_coset: bool) -> np.ndarray:
        N = self.quot_domain_size if is_quot else self.domain_size
        S = self.num_slaves
        r, c = split_rc(N)
        id = self._rng.getrandbits(64)
        workloads = make_fft_workloads(N, S)
        v = np.zeros((N, 4), dtype=np.uint64)                 # coeffs.resize(domain.size())
        v[:len(coeffs)] = coeffs
        for w in self.workers:
            w.fft_init(id, workloads, is_quot, is_inv, is_coset)
        t = decimate_rows(v, r)                               # :754
        for s, w in enumerate(self.workers):                  # one fft1 per row (:756-766)
            for j in range(workloads[s].num_rows()):
                w.fft1(id, j, t[workloads[s].row_start + j])
        self._fft2_prepare_all(id)
        u = np.empty((c, r, 4), dtype=np.uint64)
        for s, w in enumerate(self.workers):                  # :774-784
            u[workloads[s].col_start:workloads[s].col_end] = w.fft2(id, r)
        return undecimate_cols(u)                             # :786

    def _fft2_prepare_all(self, id: int):
        """fft2_prepare on all workers (:767-772).  In-process exchange: every worker records its send /
        recv buffers, then block (s -> d) is copied device-to-device (the fftExchange of worker.rs:412-438)."""
        S = self.num_slaves
        if S == 1:
            self.workers[0].fft2_prepare(id, None)
            return
        bufs = {}

        def recorder(rank):
            def cb(send, recv, nbytes, n_ranks, stream):
                bufs[rank] = (send, recv, nbytes)
                return 0
            return cb

        for s, w in enumerate(self.workers):
            w.fft2_prepare(id, recorder(s))
        for w in self.workers:
            w.sync()
        for d, wd in enumerate(self.workers):
            for s in range(S):
                send_s, _, nbytes = bufs[s]
                _, recv_d, _ = bufs[d]
                wd.memcpy_d2d(recv_d + s * nbytes, send_s + d * nbytes, nbytes)


# =====================================================================================================
# one process per GPU
# =====================================================================================================
class _DevPtr:
    """Expose a raw HBM pointer to torch without copying (`torch.as_tensor(obj, device='cuda')`)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def all_to_all_blocks(send, recv, group=None):
    """ONE all-to-all: block p of `send` goes to rank p; block p of `recv` came from rank p.  Tensors
    live wherever the backend wants them (HBM for nccl = RCCL over xGMI, host for gloo in CPU tests)."""
    import torch.distributed as dist
    dist.all_to_all_single(recv, send, group=group)


def make_torch_exchange(group=None):
    """plonk_exchange_fn on top of torch.distributed (backend "nccl" is RCCL on ROCm).  Replaces the S^2
    per-exchange TCP connections of worker.rs:303-338."""
    import torch

    def exchange(send_ptr, recv_ptr, bytes_per_peer, n_ranks, stream_ptr):
        total = bytes_per_peer * n_ranks
        dev = torch.device("cuda", torch.cuda.current_device())
        ext = torch.cuda.ExternalStream(stream_ptr, device=dev)
        with torch.cuda.stream(ext):
            send = torch.as_tensor(_DevPtr(send_ptr, total), device=dev)
            recv = torch.as_tensor(_DevPtr(recv_ptr, total), device=dev)
            all_to_all_blocks(send, recv, group)
        return 0

    return exchange


def gather_points(point: np.ndarray, group=None, device=None) -> List[np.ndarray]:
    """All-gather one Jacobian point per rank (the `result: Data` replies of varMsm, 96/144 bytes)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.from_numpy(point.astype(np.int64, copy=True).view(np.int64))
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    return [o.cpu().numpy().view(np.uint64) for o in outs]


class RankProver:
    """One rank of the multi-GPU path: this process owns one GPU, `rank` of `world` workers."""

    def __init__(self, worker: PlonkWorker, rank: int, world: int, group=None, seed: int = 0xD15EA5E, force_exchange: bool = False,
                 exchange=None, transport: str = "torch"):
        """transport "rccl": the worker carries its own RCCL communicator (PlonkWorker.comm_init) and the library performs the
        all-to-all and the point gather itself — no Python in the data path; "torch": torch.distributed (nccl = RCCL, or gloo in
        the CPU tests) through the exchange callback.  `exchange` overrides both (bench.py --simulate-ranks: a no-op)."""
        self.w, self.rank, self.world, self.group = worker, rank, world, group
        worker.me = rank
        self.transport = transport
        self._rng = random.Random(seed)          # same seed on every rank -> same task ids
        if exchange is not None:
            self._exchange = exchange
        elif transport == "rccl":
            self._exchange = None                # plonk_fft2_prepare(ctx, id, NULL, NULL): the context's communicator
        else:
            self._exchange = make_torch_exchange(group) if (world > 1 or force_exchange) else None

    def fft_dev(self, d_rows_ptr: int, d_out_ptr: int, domain_size: int, is_quot: bool, is_inv: bool, is_coset: bool,
                out_layout: int = 1, row_len: int = 0):
        """HBM-resident distributed NTT.  In: this rank's decimated rows [r/S][c] (row b = elements with
        index mod r == b) — CONSUMED (the row pass uses the buffer as its workspace, like plonk_ntt_dev's d_in).
        Out (layout 1): [r][c/S], element (j, i) = X[(i + col_start) + j*c].
        row_len > 0: the rows belong to a zero-padded vector and hold only their leading row_len coefficients
        ([r/S][row_len]; plonk_fft1_dev_compact) — the zero-padding-aware row pass, forward transforms only."""
        id = self._rng.getrandbits(64)
        wl = make_fft_workloads(domain_size, self.world)
        self.w.fft_init(id, wl, is_quot, is_inv, is_coset)
        if row_len:                                   # zero-padded input: [r/S][row_len] leading coefficients per row, not consumed
            self.w.fft1_dev_compact(id, d_rows_ptr, row_len)
        else:
            self.w.fft1_dev(id, d_rows_ptr)
        self.w.fft2_prepare(id, self._exchange)
        self.w.fft2_dev(id, d_out_ptr, out_layout)

    def msm_dev(self, start: int, end: int, d_scalars_ptr: int, device=None) -> np.ndarray:
        """This rank's share (bases[start..end) resident locally) + reduce across ranks."""
        part = self.w.msm_dev(start, end, d_scalars_ptr)
        if self.world == 1:
            return part
        acc = None
        parts = self.w.comm_allgather_host(part, self.world) if self.transport == "rccl" else gather_points(part, self.group, device)
        for p in parts:
            acc = p if acc is None else self.w.g1_add(acc, p)
        return acc
