"""Multi-rank prover by COSET CLASSES: the five rounds of `Prover::prove` (/root/reference/src/dispatcher2.rs:296-712) on G ranks
(one per GPU) with two data-path collectives per proof instead of one all-to-all per NTT.

The reference distributes every one of its 33 NTTs over the workers (row pass, all-to-all, column pass — `Prover::fft`,
dispatcher2.rs:732-787) and reassembles each result on the dispatcher.  On MI355X the per-proof vectors fit every GPU, so the
parallelism is moved to where the work is: the 8n-point quotient domain  {x_j = g * w_m^j}  splits into G classes
j = s (mod G), and class s is itself a coset  {(g * w_m^s) * w_(m/G)^k}.  Rank s

  * evaluates all 25 round-3 polynomials on ITS class with a local (m/G)-point NTT (`plonk_coset_eval_dev`) — no exchange;
  * runs the quotient kernel on its class (`plonk_quotient_evals_class_dev`) — z(w x) is point j + m/n, same class for G | 8;
  * inverts locally (`plonk_coset_interp_dev`): its additive contribution to every quotient coefficient;
  * ONE all-to-all sums the contributions (rank r ends up owning coefficients [r*m/G, (r+1)*m/G)), ONE all-gather replicates
    the quotient polynomial;
  * every commitment is an index-sharded MSM (dispatcher2.rs:870-890): rank s covers coefficients [s*L/G, (s+1)*L/G); the
    partial points of a round (96/144 bytes each) travel in ONE all-gather and are added on the host.

Of the O(n) rounds, the evaluations at zeta, the linearisation / batch polynomial, the two synthetic divisions and the quotient's
degree check are sharded by coefficient index (round 3: `_evaluate_many`, `_openings`, `_degree` — 32-byte partials travel).
Round 5 of the build distributed what rounds 1 and 2 still replicated (the reference distributes every FFT, dispatcher2.rs:294-351):

  * the seven size-n iFFTs, by RESIDUE CLASS of the coefficient index (`_interpolate_many`): coefficient s + G t of the interpolant is
    1/n * E(w_n^-(s + G t)) with E the polynomial whose coefficients are the n evaluations, so rank s needs E on the n/G points
    w_n^-s * <w_(n/G)> — `plonk_coset_eval_dev(evaluations, n, n/G, shift = w_n^-s)`, one fold of n values onto n/G (7/8 n products)
    and an (n/G)-point transform instead of an n-point one; ONE all-gather per round (the round's polynomials together,
    n/G * 32 bytes per rank and polynomial, every pair over its own xGMI link) and `plonk_class_interleave_dev` (reverse, * 1/n)
    give every rank the natural-order coefficient vector its class evaluations of round 3 read;
  * the permutation grand product, by gate range (`_perm_product`): rank s runs `plonk_perm_product_range_dev` over its n/G gates
    (+ 1: the last value is the slice's total), the 32-byte totals travel, rank s multiplies its slice by the totals before it and
    ONE all-gather (n/G * 32 bytes per rank) replicates the product vector for the iFFT above.

PLONK_CLASS_REPLICATED_R12=1 keeps rounds 1-2 replicated (A/B runs).
Results are bit-identical to the single-GPU `Prover` (and the oracle): tests/test_gpu_class_prover.py runs G = 2, 4, 8 ranks as
threads sharing one GPU; tests/test_gloo_multirank.py covers the torch.distributed transport on CPU tensors.
"""
from __future__ import annotations

import os
import threading
import time
from typing import List, Optional

import numpy as np

from .prover import Prover
from .worker import PlonkWorker


# --------------------------------------------------------------------------------------------- transports
class LocalComm:
    """G ranks as threads of one process (tests): device buffers are exchanged with plonk_memcpy_d2d, host objects through a
    shared board.  Every rank owns its PlonkWorker (context + stream) — possibly on the same GPU."""

    class Board:
        def __init__(self, size: int):
            self.size = size
            self.barrier = threading.Barrier(size)
            self.slots: List[object] = [None] * size

    def __init__(self, board: "LocalComm.Board", rank: int, worker: PlonkWorker):
        self.board, self.rank, self.size, self.w = board, rank, board.size, worker

    def all_gather_host(self, obj):
        b = self.board
        b.slots[self.rank] = obj
        b.barrier.wait()
        out = list(b.slots)
        b.barrier.wait()
        return out

    def all_to_all_dev(self, d_send: int, d_recv: int, nbytes: int):
        """block p of d_send (nbytes each) -> block `rank` of rank p's d_recv."""
        self.w.sync()
        peers = self.all_gather_host((d_recv, None))
        for p in range(self.size):
            self.w.memcpy_d2d(peers[p][0] + self.rank * nbytes, d_send + p * nbytes, nbytes)
        self.w.sync()
        self.board.barrier.wait()

    def all_gather_dev(self, d_send: int, d_recv: int, nbytes: int):
        """d_send (nbytes) -> block `rank` of every rank's d_recv."""
        self.w.sync()
        peers = self.all_gather_host(d_recv)
        for p in range(self.size):
            self.w.memcpy_d2d(peers[p] + self.rank * nbytes, d_send, nbytes)
        self.w.sync()
        self.board.barrier.wait()


class TorchComm:
    """One process per GPU under torchrun: torch.distributed (backend nccl = RCCL over xGMI) on the library's stream.
    `device` = None uses CPU tensors through host staging (gloo; tests)."""

    def __init__(self, worker: PlonkWorker, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.w, self.device = torch, dist, worker, device
        self.rank, self.size = dist.get_rank(), dist.get_world_size()

    def all_gather_host(self, obj):
        out = [None] * self.size
        self.dist.all_gather_object(out, obj)
        return out

    def _tensor(self, ptr: int, nbytes: int):
        class _Buf:
            __cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<i8", "data": (ptr, False), "version": 2}
        return self.torch.as_tensor(_Buf(), device=self.device)

    def _staged(self, ptr: int, nbytes: int):
        return self.torch.from_numpy(self.w.read_bytes(ptr, nbytes))

    def _unstage(self, t, ptr: int):
        self.w.write_bytes(ptr, t.numpy())

    def all_to_all_dev(self, d_send: int, d_recv: int, nbytes: int):
        total = nbytes * self.size
        if self.device is None:
            src, dst = self._staged(d_send, total), self.torch.empty(total // 8, dtype=self.torch.int64)
            self.dist.all_to_all_single(dst, src)
            self._unstage(dst, d_recv)
            return
        stream = self.torch.cuda.ExternalStream(self.w.stream_ptr(), device=self.device)
        with self.torch.cuda.stream(stream):
            self.dist.all_to_all_single(self._tensor(d_recv, total), self._tensor(d_send, total))

    def all_gather_dev(self, d_send: int, d_recv: int, nbytes: int):
        total = nbytes * self.size
        if self.device is None:
            src, dst = self._staged(d_send, nbytes), self.torch.empty(total // 8, dtype=self.torch.int64)
            self.dist.all_gather_into_tensor(dst, src)
            self._unstage(dst, d_recv)
            return
        stream = self.torch.cuda.ExternalStream(self.w.stream_ptr(), device=self.device)
        with self.torch.cuda.stream(stream):
            self.dist.all_gather_into_tensor(self._tensor(d_recv, total), self._tensor(d_send, nbytes))


class LibComm:
    """One process per GPU: the library's own RCCL communicator (plonk_comm_init) — grouped ncclSend/ncclRecv and ncclAllGather on
    the context's stream, no torch and no Python callback in the data path.  `bootstrap(obj) -> [obj of every rank]` is only used
    for the handful of small host objects the prover exchanges (partial commitment points travel through
    plonk_comm_allgather_host)."""

    def __init__(self, worker: PlonkWorker, bootstrap=None):
        self.w = worker
        self.rank, self.size, self.rccl_version = worker.comm_info()
        self._bootstrap = bootstrap

    def all_gather_host(self, obj):
        if isinstance(obj, (list, tuple)) and obj and all(isinstance(x, np.ndarray) for x in obj):
            obj = np.stack(obj)                   # the partial points of a round: one ncclAllGather
        if isinstance(obj, np.ndarray):
            return list(self.w.comm_allgather_host(obj, self.size))
        if self.size == 1:
            return [obj]
        if self._bootstrap is None:
            raise TypeError("LibComm.all_gather_host: only numpy arrays travel through RCCL; pass `bootstrap` for other objects")
        return self._bootstrap(obj)

    def all_to_all_dev(self, d_send: int, d_recv: int, nbytes: int):
        self.w.comm_alltoall_dev(d_send, d_recv, nbytes)

    def all_gather_dev(self, d_send: int, d_recv: int, nbytes: int):
        self.w.comm_allgather_dev(d_send, d_recv, nbytes)


# --------------------------------------------------------------------------------------------- the SPMD prover
def shard_range(length: int, rank: int, size: int):
    """coefficients [i*L/S, (i+1)*L/S) — the MSM sharding of dispatcher2.rs:875-878."""
    return rank * length // size, (rank + 1) * length // size


def key_shard_range(key_len: int, rank: int, size: int):
    """The slice of the commit key rank `rank` keeps resident: [rank*K/S, (rank+1)*K/S) of the K (padded) bases — the SRS sharding
    of dispatcher2.rs:260-266.  Every polynomial is committed against bases [0, len), so a rank's share of ANY polynomial is the
    coefficients whose index falls inside its key slice."""
    return rank * key_len // size, (rank + 1) * key_len // size


class ClassProver(Prover):
    """Rank `comm.rank` of `comm.size` (a power of two <= 8).  Same calls, same results as `Prover`; every rank must make them.
    key_range = None: `worker.init(ck, n, 8n)` was given the WHOLE commit key (replicated), each polynomial's coefficients are split
    evenly.  key_range = (lo, hi): the worker (and the commit_helper) hold only bases [lo, hi) — `key_shard_range(len(ck), rank, G)` —
    and this rank commits, for every polynomial, the coefficients with index in [lo, hi): the key is sharded G ways like the
    reference's (dispatcher2.rs:260-266), 64 / 96 B per point resident."""

    def __init__(self, worker: PlonkWorker, log_n: int, comm, commit_helper: Optional[PlonkWorker] = None, key_range=None, cache_key_cosets: bool = False,
                 fft_helper: Optional[PlonkWorker] = None):
        """cache_key_cosets: keep this rank's class evaluations of the 18 proving-key polynomials resident across proofs (18 * 8n/G * 32 B:
        9.7 GB per rank at 2^24 gates and 8 ranks) — 18 of the 25 class evaluations of round 3 depend on nothing a proof draws.  Same proof;
        NOT the reference's work (it re-transforms the key every proof, dispatcher2.rs:387-404): a labelled variant, like Prover's.
        fft_helper: another context on this rank's GPU (`init`-ed for the same domains): those 18 class evaluations are issued on ITS stream by a
        thread of their own at the start of the proof, beside rounds 1 and 2 (no collective involved), and round 3 joins them — Prover's
        fft_helper for the class prover; the reference's work, same proof."""
        super().__init__(worker, log_n, cache_key_cosets=False, commit_helper=commit_helper, fft_helper=fft_helper)
        self.cache_class_key = bool(cache_key_cosets)
        self._class_key = None
        self.key_range = None if key_range is None else (int(key_range[0]), int(key_range[1]))
        G = comm.size
        if G & (G - 1) or G > self.m // self.n:
            raise ValueError(f"{G} ranks: coset classes need a power of two <= m/n = {self.m // self.n}")
        self.comm = comm
        self.G, self.s = G, comm.rank
        f = self.f
        w_m = f.root_of_unity(self.m)
        self.shift = f.to_limbs(f.generator * pow(w_m, self.s, f.p))     # g * w_m^s : this rank's class is shift * <w_(m/G)>
        self.inv_G = f.to_limbs(f.inv(G))
        # A/B knob (bench.py --simulate-ranks): 1 = rounds 4 / 5 and the degree check computed redundantly on every rank as in round 2
        self.replicated_rounds = os.environ.get("PLONK_CLASS_REPLICATED_ROUNDS") == "1"
        # A/B knob: 1 = the size-n iFFTs and the grand product of rounds 1-3 on every rank, as before round 5
        self.replicated_r12 = os.environ.get("PLONK_CLASS_REPLICATED_R12") == "1" or G == 1 or self.n < 8 * G
        self.shift_n = f.to_limbs(pow(f.root_of_unity(self.n), (self.n - self.s) % self.n, f.p))      # w_n^-s
        self.inv_n = f.to_limbs(f.inv(self.n))

    # ---- the proving key's class evaluations beside rounds 1-2 (fft_helper)
    def _key_ffts_start(self, alloc):
        if self.fft_helper is None or self.cache_class_key:
            return
        mL, key, h, shift, n = self.m // self.G, self._key, self.fft_helper, self.shift, self.n
        d_kc = self._work("class_key_cosets_helper", 18 * mL)
        kc = [d_kc.ptr + j * mL * 32 for j in range(18)]
        self.w.sync()                                   # the helper's stream reads the key polynomials this context's stream wrote
        errs = []

        def run():
            try:
                for j, src in enumerate(key["sel"] + key["sig"]):
                    h.coset_eval_dev(src, n, mL, shift, kc[j])
                h.sync()
            except BaseException as ex:     # noqa: BLE001 - re-raised by _key_ffts_join
                errs.append(ex)

        th = threading.Thread(target=run)
        th.start()
        self._key_ffts = (th, kc, errs)

    # ---- rounds 1-3: the size-n iFFTs by residue class, the grand product by gate range (module docstring)
    def _interpolate_many(self, alloc, pairs):
        if self.replicated_r12:
            return super()._interpolate_many(alloc, pairs)
        w, n, G, s = self.w, self.n, self.G, self.s
        L, K = n // G, len(pairs)
        # named work buffers shared by the three calls of a proof (K = 5, 1, 1 polynomials: the first call sizes them), not numbered ones per call
        d_mine = self._work("interp_mine", K * L)    # [polynomial][L]: this class's values of every polynomial of the round
        d_all = self._work("interp_all", G * K * L)  # [class][polynomial][L] after the all-gather
        for k, (src, _) in enumerate(pairs):
            w.coset_eval_dev(src, n, L, self.shift_n, d_mine.ptr + k * L * 32)
        self.comm.all_gather_dev(d_mine.ptr, d_all.ptr, K * L * 32)
        for k, (_, dst) in enumerate(pairs):
            w.class_interleave_dev(d_all.ptr + k * L * 32, G, L, True, self.inv_n, dst, in_stride=K * L)

    def _perm_product(self, alloc, wev, d_id: int, d_idx: int, beta, gamma) -> int:
        if self.replicated_r12:
            return super()._perm_product(alloc, wev, d_id, d_idx, beta, gamma)
        w, f, n, G, s = self.w, self.f, self.n, self.G, self.s
        lo, hi = shard_range(n, s, G)
        cnt = hi - lo
        last = hi == n                               # nobody multiplies by the last slice's total
        d_loc = alloc(cnt + 1)
        # A zero denominator or an out-of-range permutation index shows up only on the rank whose gate slice holds it (the reference panics,
        # dispatcher2.rs:338-343).  Raising here, BEFORE the collective, would leave the other ranks waiting in it for ever: the failure travels
        # as a status word beside the 32-byte slice total, and every rank raises the same error after the all-gather.
        failure = None
        try:
            w.perm_product_range_dev(wev, d_id, d_idx, beta, gamma, n, lo, cnt + (0 if last else 1), d_loc.ptr)
            total = self.f.to_limbs(1) if last else w.read_bytes(d_loc.ptr + cnt * 32, 32).view(np.uint64).copy()
        except Exception as ex:     # noqa: BLE001 - re-raised below, on every rank
            failure, total = ex, self.f.to_limbs(1)
        word = np.zeros((1, 8), dtype=np.uint64)
        word[0, :4] = np.asarray(total, dtype=np.uint64).reshape(-1)[:4]
        word[0, 4] = 0 if failure is None else 1
        gathered = [np.asarray(part, dtype=np.uint64).reshape(-1) for part in self.comm.all_gather_host(word)]
        failed = [r for r, part in enumerate(gathered) if int(part[4]) != 0]
        if failed:
            if failure is not None:
                raise failure
            raise ZeroDivisionError(f"permutation product: rank(s) {failed} reported a zero denominator or an out-of-range permutation index "
                                    f"in their gate range (the reference panics, dispatcher2.rs:338-343)")
        pre = 1
        for part in gathered[:s]:
            pre = pre * f.from_limbs(part[:4]) % f.p
        d_send = alloc(cnt)
        w.poly_lincomb_dev([(d_loc.ptr, cnt)], f.vec_to_limbs([pre]), d_send.ptr, cnt)
        d_prod = alloc(n)
        self.comm.all_gather_dev(d_send.ptr, d_prod.ptr, cnt * 32)
        return d_prod.ptr

    # ---- commitments: index-sharded MSMs, ONE all-gather of the round's partial points (dispatcher2.rs:870-892)
    def _commit(self, d_poly: int, length: int):
        return self._commit_many([(d_poly, length)])[0]

    def _commit_many(self, items):
        """The shard [s*L/G, (s+1)*L/G) of every polynomial of the round (two contexts / host threads when a commit_helper
        is set), then one collective for all partial points and the host reduce(a + b) per commitment."""
        shards = []
        for ptr, ln in items:
            if self.key_range is None:
                lo, hi = shard_range(ln, self.s, self.G)
                shards.append((ptr + lo * 32, lo, hi - lo))
            else:                                                # sharded key: coefficient i pairs with LOCAL base i - key_lo
                klo, khi = self.key_range
                lo, hi = min(klo, ln), min(khi, ln)
                if hi <= lo:                                     # the polynomial ends before this rank's key slice: an empty shard
                    shards.append((ptr, 0, 0))                   # (lo - klo would be negative, i.e. ~2^64 through c_size_t)
                else:
                    shards.append((ptr + lo * 32, lo - klo, hi - lo))
        parts = [None] * len(shards)
        lanes = [self.w] + ([self.commit_helper] if self.commit_helper is not None and len(shards) > 1 else [])

        def commit_shards(worker, idx):
            """The shards idx of the round on one context: those that start at the same key index go through plonk_commit_many_dev as
            ONE Pippenger problem (at 2^21 points per rank the fixed costs of an MSM exceed its bucket accumulation)."""
            by_start = {}
            for i in idx:
                by_start.setdefault(shards[i][1], []).append(i)
            for start, group in by_start.items():
                if len(group) == 1:
                    parts[group[0]] = worker.commit_range_dev(*shards[group[0]])
                else:
                    for i, j in zip(group, worker.commit_many_dev([(shards[i][0], shards[i][2]) for i in group], start=start)):
                        parts[i] = j

        if len(lanes) == 1:
            commit_shards(self.w, range(len(shards)))
        else:
            self.w.sync()
            errs = []

            def run(lane):
                try:
                    commit_shards(lanes[lane], range(lane, len(shards), 2))
                except BaseException as ex:     # noqa: BLE001 - re-raised below
                    errs.append(ex)

            th = [threading.Thread(target=run, args=(lane,)) for lane in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errs:
                raise errs[0]
        acc = [None] * len(shards)
        for rank_parts in self.comm.all_gather_host(parts):               # collectives only on this thread, in program order
            for i, p in enumerate(rank_parts):
                acc[i] = p if acc[i] is None else self.w.g1_add(acc[i], p)
        return [self.w.g1_to_affine(a) for a in acc]

    # ---- the O(n) rounds, sharded by coefficient index (round 2 of this file computed them redundantly on every rank: ~20 ms at
    # n = 2^24 that did not shrink with G).  Only 32-byte partials travel: one small all-gather per step.
    def _sum_gathered(self, rows: np.ndarray) -> List[np.ndarray]:
        """rows (k, 4): this rank's k partial values -> the k sums over all ranks (Fr limbs), identical on every rank."""
        f = self.f
        acc = [0] * rows.shape[0]
        for part in self.comm.all_gather_host(np.ascontiguousarray(rows, dtype=np.uint64)):
            for j in range(rows.shape[0]):
                acc[j] = (acc[j] + f.from_limbs(part[j])) % f.p
        return [f.to_limbs(x) for x in acc]

    def _evaluate_many(self, polys, points):
        """Round 4 (:545-555): rank s evaluates coefficients [s*L/G, (s+1)*L/G) of every polynomial — sum_i c_i z^i over the slice is
        z^lo times the evaluation of the slice — and the 10 partial values are added across ranks."""
        if self.replicated_rounds:
            return super()._evaluate_many(polys, points)
        f, w = self.f, self.w
        rows = np.zeros((len(polys), 4), dtype=np.uint64)
        for j, ((ptr, ln), pt) in enumerate(zip(polys, points)):
            lo, hi = shard_range(ln, self.s, self.G)
            if hi > lo:
                part = f.from_limbs(w.poly_eval_dev(ptr + lo * 32, hi - lo, pt))
                rows[j] = f.to_limbs(part * pow(f.from_limbs(pt), lo, f.p) % f.p)
        return self._sum_gathered(rows)

    def _degree(self, d_poly: int, length: int) -> int:
        """The quotient is replicated after the all-gather; each rank scans its slice, the highest non-zero index wins."""
        if self.replicated_rounds:
            return super()._degree(d_poly, length)
        lo, hi = shard_range(length, self.s, self.G)
        d = self.w.poly_degree_dev(d_poly + lo * 32, hi - lo) if hi > lo else -1
        mine = np.array([lo + d + 1 if d >= 0 else 0], dtype=np.uint64)         # 0 = all zero in my slice
        return int(max(int(x[0]) for x in self.comm.all_gather_host(mine))) - 1

    def _commit_range(self, length: int):
        """The coefficient indices [lo, hi) of a length-`length` polynomial this rank commits (what _commit_many picks)."""
        if self.key_range is None:
            return shard_range(length, self.s, self.G)
        lo, hi = min(self.key_range[0], length), min(self.key_range[1], length)
        return (lo, hi) if hi > lo else (0, 0)

    def _openings(self, alloc, lin_terms, lin_coeffs, batch_terms, batch_coeffs, perm_poly, zeta, zeta_w, PP: int, keep: bool) -> dict:
        """Round 5 (:566-697) with every vector restricted to the slice this rank commits.  The witness polynomial of an opening is
        q_t = sum_{k > t} c_k z^(k-t-1) (the synthetic division of :651-666).  For t in [lo, hi) that is the division of the local
        piece (c_(lo+1) .. c_hi) with ONE extra top coefficient T = q_hi = sum over the higher ranks' pieces, and
        T = sum_{r' above} z^(lo_r' - hi) * E_r' with E_r' the evaluation of rank r''s own piece at z: 32 bytes per rank and
        polynomial travel, the batch polynomial is only ever formed on the slice (1/G of the scalar * polynomial work)."""
        if self.replicated_rounds:
            return super()._openings(alloc, lin_terms, lin_coeffs, batch_terms, batch_coeffs, perm_poly, zeta, zeta_w, PP, keep)
        w, f, p = self.w, self.f, self.f.p
        Lq = PP - 1
        lo, hi = self._commit_range(Lq)
        cnt = hi - lo
        terms = list(lin_terms) + list(batch_terms)
        coeffs = list(lin_coeffs) + list(batch_coeffs)
        d_b = alloc(2 * (cnt + 2))                                   # two pieces: [x, c_(lo+1) .. c_hi, T]
        piece = [d_b.ptr, d_b.ptr + (cnt + 2) * 32]
        rows = np.zeros((2, 6), dtype=np.uint64)                     # (lo, hi, E limbs) per opening
        pts = (zeta, zeta_w)
        if cnt > 0:
            w.memset_dev(d_b.ptr, 0, 2 * (cnt + 2) * 32)
            sl = [(ptr + (lo + 1) * 32, min(ln, hi + 1) - (lo + 1)) for ptr, ln in terms]
            live = [(t, c) for t, c in zip(sl, coeffs) if t[1] > 0]
            w.poly_lincomb_dev([t for t, _ in live], f.vec_to_limbs([c for _, c in live]), piece[0] + 32, cnt)       # batch_poly on (lo, hi]
            have = min(perm_poly[1], hi + 1) - (lo + 1)
            if have > 0:
                w.memcpy_d2d(piece[1] + 32, perm_poly[0] + (lo + 1) * 32, have * 32)
            for j in range(2):
                rows[j, 0], rows[j, 1] = lo, hi
                rows[j, 2:6] = w.poly_eval_dev(piece[j] + 32, cnt, pts[j])
        gathered = self.comm.all_gather_host(rows)
        d_q = alloc(2 * (cnt + 2))
        virt = []
        for j in range(2):
            if cnt > 0:
                z = f.from_limbs(pts[j])
                T = 0
                for part in gathered:
                    plo, phi = int(part[j, 0]), int(part[j, 1])
                    if phi > plo and plo >= hi:
                        T = (T + f.from_limbs(part[j, 2:6]) * pow(z, plo - hi, p)) % p
                w.write_bytes(piece[j] + (cnt + 1) * 32, f.to_limbs(T).view(np.int64))
                w.poly_div_linear_dev(piece[j], cnt + 2, pts[j], d_q.ptr + j * (cnt + 2) * 32)      # out[t - lo] = q_t, t in [lo, hi)
            virt.append(d_q.ptr + j * (cnt + 2) * 32 - lo * 32)      # a pointer _commit_many offsets by lo * 32 again
        out = dict(comms=self._commit_many([(virt[0], Lq), (virt[1], Lq)]))
        if keep:                                                     # debug copies of the whole polynomials (tests): the proof above never used them
            d_lin, d_batch = alloc(PP), alloc(PP)
            w.poly_lincomb_dev(list(lin_terms), f.vec_to_limbs(list(lin_coeffs)), d_lin.ptr, PP)
            w.poly_lincomb_dev([(d_lin.ptr, PP)] + list(batch_terms), f.vec_to_limbs([1] + list(batch_coeffs)), d_batch.ptr, PP)
            out["lin_poly"], out["batch_poly"] = d_lin.download((PP, 4)), d_batch.download((PP, 4))
        return out

    # ---- the quotient's coset evaluations, class by class
    def _quotient_poly(self, alloc, tick, wire_polys, perm_poly, pi_poly, alpha, beta, gamma) -> int:
        w, n, m, key, G, s = self.w, self.n, self.m, self._key, self.G, self.s
        mL = m // G
        t0 = time.perf_counter()
        d_cls = alloc(25 * mL)
        cls = [d_cls.ptr + j * mL * 32 for j in range(25)]
        srcs = [(p, n) for p in key["sel"] + key["sig"]] + list(wire_polys) + [perm_poly, pi_poly]
        if self.cache_class_key:
            if self._class_key is None or self._class_key[0] is not key:
                d_key = self._work("class_key_cosets", 18 * mL)
                for j in range(18):
                    w.coset_eval_dev(srcs[j][0], srcs[j][1], mL, self.shift, d_key.ptr + j * mL * 32)
                self._class_key = (key, d_key)
            cls[:18] = [self._class_key[1].ptr + j * mL * 32 for j in range(18)]
        joined = self._key_ffts is not None
        if joined:
            cls[:18] = self._key_ffts_join()
        for j, (ptr, ln) in enumerate(srcs):
            if (self.cache_class_key or joined) and j < 18:
                continue
            w.coset_eval_dev(ptr, ln, mL, self.shift, cls[j])             # this class's slice of the coset FFT of :387-429
        tick("round3_coset_ffts", t0)
        t0 = time.perf_counter()
        d_qev = alloc(mL)
        w.quotient_evals_dev(cls[0:13], cls[13:18], cls[18:23], cls[23], cls[24], alpha, beta, gamma, key["k"], d_qev.ptr,
                             class_stride=G, class_offset=s)
        d_contrib = alloc(m)                                              # this class's share of every coefficient
        w.coset_interp_dev(d_qev.ptr, mL, self.shift, self.inv_G, 0, m, d_contrib.ptr)
        tick("round3_quotient", t0)
        t0 = time.perf_counter()
        d_recv = alloc(m)
        self.comm.all_to_all_dev(d_contrib.ptr, d_recv.ptr, mL * 32)     # block r: coefficients [r*mL, (r+1)*mL)
        d_mine = alloc(mL)
        ones = np.tile(self.f.to_limbs(1), (G, 1))
        w.poly_lincomb_dev([(d_recv.ptr + p * mL * 32, mL) for p in range(G)], ones, d_mine.ptr, mL)
        d_quot = alloc(m)
        self.comm.all_gather_dev(d_mine.ptr, d_quot.ptr, mL * 32)
        tick("round3_exchange", t0)
        return d_quot.ptr


def run_local_ranks(G: int, fn, device: int = 0, curve: str = "bn254"):
    """Run fn(comm, worker) on G threads sharing one GPU (tests / single-GPU dry runs of the multi-rank path).
    Returns the list of results; re-raises the first exception."""
    board = LocalComm.Board(G)
    results: List[object] = [None] * G
    errors: List[Optional[BaseException]] = [None] * G

    def body(r):
        w = PlonkWorker(me=r, device=device, curve=curve)
        try:
            results[r] = fn(LocalComm(board, r, w), w)
        except BaseException as e:          # noqa: BLE001 - reported to the caller below
            errors[r] = e
            board.barrier.abort()
        finally:
            w.close()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errors:
        if e is not None:
            raise e
    return results
