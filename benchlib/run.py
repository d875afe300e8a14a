"""The state of one bench.py run — process group, contexts, HBM-resident synthetic inputs — and the proof-equivalent step it times.

One "step" = one pass of the hot path over one batch of synthetic input: 7 (i)NTT(n), 25 coset-NTT(8n) from n + 3 coefficients, 1 coset-iNTT(8n)
and 13 KZG commitments (/root/reference/src/dispatcher2.rs:294-691).  Everything here runs on every rank."""
import os
import threading
import time

from .cli import plan
from .common import N_MSM, N_NTT_BIG, N_NTT_SMALL

ROUNDS = (5, 1, 5, 2)      # commitments per prover round: dispatcher2.rs:313-321, 352-358, 519-531, 690-697


class Bench:
    def __init__(self, args):
        import numpy as np
        import torch
        import torch.distributed as dist
        self.args, self.np, self.torch, self.dist = args, np, torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus and self.world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        # tests/test_hostemu.py only: a DRY RUN of this program's control flow (the N > 1 legs above all) against the host emulation of the
        # library — rank processes under gloo, no GPU, tiny sizes.  Its JSON line says `emulated` and carries no value.
        self.emulated = os.environ.get("PLONK_ALLOW_HOSTEMU") == "1"
        if self.emulated:
            args.no_cpu_baseline = args.no_other_configs = True
            if args.transport != "rccl":
                raise SystemExit("bench.py under the host emulation: only the default in-library transport has a stand-in (tests/hostemu/comm_local.cpp)")
        elif not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product path has no CPU fallback)")
        if not self.emulated:
            torch.cuda.set_device(self.local_rank)
        self.pg_kwargs = dict(backend="gloo") if self.emulated else dict(backend="nccl", device_id=torch.device("cuda", self.local_rank))
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(**self.pg_kwargs)
        self.n = 1 << args.log_n
        self.m = 8 * self.n
        self.nbig = 0 if args.n_domain_only else N_NTT_BIG          # size-8n transforms per step
        if args.n_domain_only:
            args.scheme, args.no_class_prover, args.next_rows, args.no_other_configs, args.no_cpu_baseline = "reference2d", True, "none", True, True
        self.S = self.world
        self.sim = args.simulate_ranks if (self.world == 1 and args.simulate_ranks > 1) else 0
        if self.sim:
            self.S = self.sim
        pl = plan(args, self.S)
        if not pl["ok"]:
            raise SystemExit("bench.py: " + "; ".join(pl["problems"]))
        self.dev = torch.device("cpu") if self.emulated else torch.device("cuda", self.local_rank)
        # N > 1: two contexts (two HIP streams) per rank, so that the all-to-all of one transform overlaps the
        # row / column passes of the next (the 26 size-8n transforms of a proof are independent polynomials)
        self.multi = self.S > 1 or args.multi_path
        self.n_lanes = 2 if self.multi else 1
        self.phase = {"ntt": 0.0, "msm": 0.0}              # host-clock split of the step at its one internal sync point (this rank)
        # N = 1 only: the transforms on a context of their own, issued while the commitment threads run (--overlap-phases; cli.py has the numbers)
        self.overlap = (not self.multi and self.S == 1 and self.nbig > 0 and
                        (args.overlap_phases == "on" or (args.overlap_phases == "auto" and args.log_n <= 22)))
        # N > 1 (and its one-GPU diagnostics): on unless 'off' (profiles/r05_opening_measurements.txt: rank 0 of 8 simulated, 101.6 -> 96.2 ms per
        # step in one lease).  The two transform lanes get contexts (and communicators) of their own; the commitment threads run on the two
        # commitment contexts while the main thread issues the distributed transforms, so the 33 all-to-alls are also covered by the MSMs of the
        # step.  Every collective still comes from the main thread, in the same order on every rank (all-to-alls, then the point all-gather).
        self.overlap_multi = (self.multi and self.nbig > 0 and args.overlap_phases != "off" and (args.transport == "rccl" or bool(self.sim)))
        self._contexts()
        self._inputs()
        self._class_scheme_inputs()
        self.scheme = args.scheme if self.multi else "single"
        self.step = self.step_classes if self.scheme == "classes" else self.step_ref2d

    # ------------------------------------------------------------------------------------------------ set-up
    def _contexts(self):
        from distributed_plonk_amd.dispatcher import RankProver
        from distributed_plonk_amd.worker import PlonkWorker
        args, dist = self.args, self.dist
        # always two contexts for the commitments: the 13 MSMs of a proof are independent, and a second stream fills the sort /
        # reduction phases and the wave tail of one MSM with the bucket accumulation of the next (measured: 29.3 -> 26.8 ms per
        # 2^24-point commit, 4.9 -> 4.0 ms at the 2^21 points of an 8-rank shard; tools/msm_overlap.py)
        self.n_commit_lanes = max(2, int(os.environ.get("PLONK_BENCH_COMMIT_LANES", "2")))      # experiment knob; 2 is the measured choice
        self.workers = [PlonkWorker(me=self.rank, device=self.local_rank, curve=args.curve) for _ in range(self.n_commit_lanes)]
        self.w = self.workers[0]
        self.wt = PlonkWorker(me=self.rank, device=self.local_rank, curve=args.curve) if self.overlap else self.w      # the transforms' context
        self.step_workers = [PlonkWorker(me=self.rank, device=self.local_rank, curve=args.curve) for _ in range(self.n_lanes)] if self.overlap_multi else []
        self.q64 = self.w.q64
        if os.environ.get("PLONK_BENCH_MSM_WINDOW"):                 # experiment knob: force the Pippenger window (0 / unset: the library's cost model)
            for x in self.workers:
                x.set_option("msm_window", int(os.environ["PLONK_BENCH_MSM_WINDOW"]))
        if os.environ.get("PLONK_BENCH_ACC_PERSIST") is not None:    # experiment knob: workgroups per CU of the persistent accumulation (0 = plain grid)
            for x in self.workers:
                x.set_option("msm_acc_persist", int(os.environ["PLONK_BENCH_ACC_PERSIST"]))
        self.experiment_opts = {}
        for kv in filter(None, os.environ.get("PLONK_BENCH_OPTS", "").split(",")):   # experiment knob: "key=value,..." through plonk_set_option on every context
            k_, v_ = kv.split("=")                                                    # (e.g. msm_reduce_grid=0); recorded in config.experiment_opts
            self.experiment_opts[k_.strip()] = int(v_)
            for x in self.workers:
                x.set_option(k_.strip(), int(v_))
        # --simulate-ranks: the exchange of a distributed transform is the library's stand-in (device-to-device copies of the foreign blocks on the
        # transform's stream, no Python in the call) or, with --sim-exchange none, nothing
        self.sim_standin = bool(self.sim) and args.sim_exchange == "standin"
        self.sim_bytes = {"alltoall": 0, "allgather": 0, "calls": 0}       # what rank 0 would send per call, summed (sim_exchange_report)
        self.noop_exchange = None
        if self.sim:
            import ctypes as C_
            from distributed_plonk_amd import _ffi
            self.noop_exchange = (C_.cast(_ffi.lib().plonk_exchange_standin, _ffi.EXCHANGE_FN) if self.sim_standin else
                                  (lambda send, recv, nbytes, n_ranks, stream: 0))
        self.transport = args.transport if (self.world > 1 or args.multi_path) else "torch"
        self.rccl_info = None
        if args.multi_path and self.world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29655")
            dist.init_process_group(rank=0, world_size=1, **self.pg_kwargs)
        if (self.world > 1 or args.multi_path) and self.transport == "rccl" and not self.sim:
            # one RCCL communicator per context (two streams -> two communicators), created in the same order on every rank; the
            # 128-byte ids travel once through the launcher's rendezvous — nothing else of the data path touches torch
            with_comm = self.workers + self.step_workers
            ids = [PlonkWorker.comm_unique_id() for _ in with_comm] if self.rank == 0 else [None] * len(with_comm)
            dist.broadcast_object_list(ids, src=0)
            for x, uid in zip(with_comm, ids):
                x.comm_init(uid, self.rank, self.world)
            r_, w_, v_ = self.w.comm_info()
            # what EVERY rank's communicator reports (plonk_comm_info = ncclCommUserRank / ncclCommCount), gathered once through the launcher: the first
            # run on more than one GPU shows at a glance whether each process got its own device and the communicator spans them all
            seen = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(seen, (self.rank, r_, w_, self.local_rank))
            else:
                seen = [(self.rank, r_, w_, self.local_rank)]
            self.rccl_info = {"rank0_reports_world": w_, "rccl_version": v_, "communicators_per_rank": len(with_comm),
                              "ranks_seen": sorted(s_[1] for s_ in seen), "world_seen_by_every_rank": sorted({s_[2] for s_ in seen}),
                              "devices": [s_[3] for s_ in sorted(seen)],
                              "order_check": os.environ.get("PLONK_COMM_CHECK_ORDER", "0") not in ("", "0")}
        self.provers = [RankProver(x, self.rank, self.S, exchange=self.noop_exchange, transport=self.transport) for x in self.workers[:self.n_lanes]]
        # the provers of the STEP: their own contexts under --overlap-phases on, else the two above (the legs after the headline always use those)
        self.step_provers = ([RankProver(x, self.rank, self.S, exchange=self.noop_exchange, transport=self.transport) for x in self.step_workers]
                             if self.overlap_multi else self.provers)
        self.torch_comm = None
        if self.multi and self.transport == "torch" and not self.sim:
            from distributed_plonk_amd.class_prover import TorchComm
            self.torch_comm = TorchComm(self.w, None if self.emulated else self.dev)

    def _inputs(self):
        """Resident synthetic inputs (seeded; the reference uses thread_rng)."""
        from distributed_plonk_amd.dispatcher import split_rc
        args, w, n, m, S, rank, nbig = self.args, self.w, self.n, self.m, self.S, self.rank, self.nbig
        self.n_loc, self.m_loc = n // S, (m // S if nbig else 8)
        n_loc, m_loc, n_lanes = self.n_loc, self.m_loc, self.n_lanes
        # Every operation of a step has its OWN input (committing one scalar vector 13 times and transforming one polynomial
        # 25 times cannot show a data-dependent defect): 7 vectors for the size-n iNTTs, 25 coefficient vectors for the forward coset
        # FFTs, 13 scalar vectors for the commitments — 26 GiB at n = 2^24.  Above 2^26 (configs[4]: 8 GiB per vector) they are shared.
        self.distinct_inputs = args.log_n <= 26
        self.n_small_bufs = N_NTT_SMALL if self.distinct_inputs else n_lanes
        self.buf_n = [[w.alloc(n_loc * 32), w.alloc(n_loc * 32)] for _ in range(max(self.n_small_bufs, n_lanes))]
        self.buf_m = [[w.alloc(m_loc * 32), w.alloc(m_loc * 32)] for _ in range(n_lanes)]
        for i, pair in enumerate(self.buf_n):
            w.synth_fr(0xD15EA5E + 64 * rank + i, pair[0].ptr, n_loc)
        for lane in range(n_lanes):
            w.synth_fr(0xBADC0DE + 16 * rank + lane, self.buf_m[lane][0].ptr, m_loc)
        n_scal = N_MSM if self.distinct_inputs else 2
        self.scal = [w.alloc(n_loc * 32) for _ in range(n_scal)]          # commit_polynomial takes Montgomery coefficients (into_repr inside)
        for i, b in enumerate(self.scal):
            w.synth_fr(0x5CA1A5 + 64 * rank + i, b.ptr, n_loc)
        # the coefficient vectors the 25 forward coset transforms start from: n + 3 coefficients (the blinded permutation polynomial's
        # length; wires have n + 2, selectors n), which the reference zero-pads to 8n (dispatcher2.rs:746)
        self.padded = (S == 1) and not self.multi and not args.dense_coset and nbig > 0
        self.poly_len = n + 3
        self.gen_limbs = None
        self.polys = []
        self.n_polys = (N_NTT_BIG - 1) if self.distinct_inputs else 1
        if self.padded:
            from distributed_plonk_amd import fr as _fr
            self.gen_limbs = _fr.FIELDS[args.curve].to_limbs(_fr.FIELDS[args.curve].generator)
            self.polys = [w.alloc(self.poly_len * 32) for _ in range(self.n_polys)]
            for i, b in enumerate(self.polys):
                w.synth_fr(0xC0EFF + i, b.ptr, self.poly_len)
        self.bases = w.alloc(n_loc * 16 * self.q64)
        # SRS shard of this rank: pairwise-distinct points (or 2^11 random points tiled, dispatcher.rs:190-196)
        w.synth_bases(0x5EED + rank, 0 if args.bases == "distinct" else min(n_loc, 1 << 11), n_loc, self.bases.ptr)
        for x in self.workers + self._extra_contexts():
            x.init_dev(self.bases.ptr, n_loc, n, m if nbig else 0)      # both contexts hold the SRS shard in the resident limb form
            x.sync()
        # reference2d on N > 1 ranks: the zero-padded polynomial arrives as this rank's decimated rows, of which only the leading
        # c/8 + 1 coefficients can be non-zero (dispatcher2.rs:746, 754) — plonk_fft1_dev_compact
        self.rows_compact, self.row_len_m = [], 0
        if self.multi and not args.dense_coset and nbig:
            r_m, c_m = split_rc(m)
            self.row_len_m = (self.poly_len + r_m - 1) // r_m
            self.rows_compact = [w.alloc((r_m // S) * self.row_len_m * 32) for _ in range(self.n_polys)]
            for i, b in enumerate(self.rows_compact):
                w.synth_fr(0xC0EFF + 64 * rank + i, b.ptr, (r_m // S) * self.row_len_m)
        # The 13 commitments of a proof come in rounds; the commitments of one round are independent and go through
        # plonk_commit_many_dev as ONE Pippenger problem, split over the commit lanes.  PLONK_BENCH_COMMIT_BATCH=0: one MSM at a time.
        self.commit_batch = os.environ.get("PLONK_BENCH_COMMIT_BATCH", "1") != "0"
        self.use_lanes = min(self.n_commit_lanes, max(1, int(os.environ.get("PLONK_BENCH_COMMIT_USE_LANES", str(self.n_commit_lanes)))))

    def _class_scheme_inputs(self):
        """N > 1, scheme "classes": the step with the coset-class decomposition.  Every rank holds the coefficient
        vectors (the size-n iNTTs that produce them run by residue class: an n/N-point class transform, one all-gather, one interleave), evaluates all 25
        polynomials on its OWN class of the 8n-point coset with a local zero-padding-aware (8n/N)-point transform, and the quotient's
        coset iFFT is the class-local inverse + one all-to-all (sum) + one all-gather.  Same work as the reference's 33 distributed
        transforms, two data-path collectives instead of 33."""
        self.cls = None
        if not (self.multi and self.nbig):
            return
        from distributed_plonk_amd import fr as _fr
        np, w, n, m, rank = self.np, self.w, self.n, self.m, self.rank
        f_ = _fr.FIELDS[self.args.curve]
        G = self.S
        mL = m // G
        self.cls = dict(
            bn=[[w.alloc(n * 32), w.alloc(n * 32)] for _ in range(self.n_small_bufs)], polys=[w.alloc(self.poly_len * 32) for _ in range(self.n_polys)],
            out=w.alloc(mL * 32), contrib=w.alloc(m * 32), recv=w.alloc(m * 32), mine=w.alloc(mL * 32), quot=w.alloc(m * 32),
            shift=f_.to_limbs(f_.generator * pow(f_.root_of_unity(m), rank, f_.p) % f_.p), inv_g=f_.to_limbs(pow(G, -1, f_.p)),
            cls_mine=w.alloc((n // G) * 32), cls_all=w.alloc(n * 32), inv_n=f_.to_limbs(pow(n, -1, f_.p)),
            shift_n=f_.to_limbs(pow(f_.root_of_unity(n), (n - rank) % n, f_.p)),
            ones=np.tile(f_.to_limbs(1), (G, 1)))
        for i, pair in enumerate(self.cls["bn"]):                     # the same vectors on every rank
            w.synth_fr(0xD15EA5E + i, pair[0].ptr, n)
        for i, b in enumerate(self.cls["polys"]):
            w.synth_fr(0xC0EFF + i, b.ptr, self.poly_len)
        w.synth_fr(0x5EC7, self.cls["recv"].ptr, m)                  # (--simulate-ranks skips the exchange: keep the operands valid)

    # ------------------------------------------------------------------------------------------------ --simulate-ranks: the exchange stand-in
    @property
    def sim_exchange_note(self):
        return ("collectives replaced by device-to-device copies of the bytes they would move" if self.sim_standin else "no exchange")

    def sim_alltoall(self, w, d_send, d_recv, nbytes, force=False):
        """blocks 1 .. S-1 of d_send -> d_recv on w's stream (what leaves for / arrives from the S - 1 peers)"""
        if (self.sim_standin or force) and self.S > 1:
            w.memcpy_d2d_async(d_recv + nbytes, d_send + nbytes, (self.S - 1) * nbytes)

    def sim_allgather(self, w, d_send, d_recv, nbytes, force=False):
        """d_send -> the S - 1 foreign blocks of d_recv (the bytes that would arrive), and the rank's own block"""
        if self.sim_standin or force:
            for p_ in range(self.S):
                w.memcpy_d2d_async(d_recv + p_ * nbytes, d_send, nbytes)

    def sim_exchange_report(self):
        """what the stand-in moved per step and what those bytes cost on the fabric: MI355X xGMI is point-to-point, 7 links per GPU at
        ~153 GB/s peak per direction (MI355X_MICROARCH.md); in an all-to-all / all-gather over 8 GPUs every pair has its own link, so the
        collective's floor is bytes_per_peer / link rate — quoted at 100 % and at 60 % of the link peak.  A MODEL beside a measurement."""
        S, n, m = self.S, self.n, self.m
        per_peer = {"ntt_n": (n // S // S) * 32, "ntt_8n": (m // S // S) * 32 if self.nbig else 0}
        colls = {"ntt_n": 7, "ntt_8n": 26 if self.nbig else 0}
        floor = lambda rate: sum(colls[k_] * per_peer[k_] / rate for k_ in colls) * 1e3
        return {"mode": "standin" if self.sim_standin else "none",
                "what": self.sim_exchange_note,
                "reference2d_bytes_per_peer_and_collective": per_peer, "collectives_per_step": colls,
                "bytes_out_per_rank_and_step": sum(colls[k_] * per_peer[k_] * (S - 1) for k_ in colls),
                "xgmi_model_ms_per_step": {"at_153_GBps_per_link": round(floor(153e9), 2), "at_92_GBps_per_link": round(floor(92e9), 2),
                                           "note": "serial sum of the 33 all-to-alls at a per-link rate, every pair on its own link; with two lanes (and "
                                                   "--overlap-phases) they overlap the other lane's passes and the commitments, so this is an upper "
                                                   "bound of what can be exposed, not an addend"}}

    def _extra_contexts(self):
        """the contexts --overlap-phases adds to the two commitment contexts"""
        return ([self.wt] if self.wt is not self.w else []) + self.step_workers

    # ------------------------------------------------------------------------------------------------ the operations of a step
    def ntt(self, lane, bufs, size, inv, coset, is_quot):
        """one whole-vector / distributed transform; the pair ping-pongs (the next step transforms this step's output)"""
        if self.S == 1:
            self.wt.ntt_dev(bufs[0].ptr, bufs[1].ptr, size, inv, coset)
        else:
            self.step_provers[lane].fft_dev(bufs[0].ptr, bufs[1].ptr, size, is_quot, inv, coset, out_layout=1)
        bufs[0], bufs[1] = bufs[1], bufs[0]

    def coset_fft_8n(self, lane, i):
        """quot_domain.coset_fft of polynomial i of the step (dispatcher2.rs:387-424)."""
        buf_m = self.buf_m
        if self.padded:
            self.wt.coset_eval_dev(self.polys[i % len(self.polys)].ptr, self.poly_len, self.m, self.gen_limbs, buf_m[lane][0].ptr)
        elif self.rows_compact:
            self.step_provers[lane].fft_dev(self.rows_compact[i % len(self.rows_compact)].ptr, buf_m[lane][1].ptr, self.m, True, False, True, out_layout=1,
                                       row_len=self.row_len_m)
            buf_m[lane][0], buf_m[lane][1] = buf_m[lane][1], buf_m[lane][0]
        else:
            self.ntt(lane, buf_m[lane], self.m, False, True, True)

    @staticmethod
    def commit_groups(count):
        groups, at = [], 0
        while at < count:
            for r in ROUNDS:
                r = min(r, count - at)
                if r:
                    groups.append((at, r))
                    at += r
        return groups

    def commits_start(self, count):
        """commitment i of the step takes scalar vector i"""
        src = [self.scal[i % len(self.scal)].ptr for i in range(count)]
        parts = [None] * count
        errs = []
        use_lanes, n_loc, cworkers = self.use_lanes, self.n_loc, self.workers

        def run(lane):
            try:
                if self.commit_batch:
                    for gi, (at, r) in enumerate(self.commit_groups(count)):
                        # the odd polynomial of a round goes to another context every round (5 -> 3 + 2, then 1 -> 0 + 1, 5 -> 2 + 3, ...):
                        # 7 + 6 commitments per step instead of 8 + 5, so neither context runs a long tail alone
                        mine = list(range(at + (lane + gi) % use_lanes, at + r, use_lanes))
                        if mine:
                            pts = cworkers[lane].commit_many_dev([(src[i], n_loc) for i in mine])
                            for j, i in enumerate(mine):
                                parts[i] = pts[j]
                    return
                for i in range(lane, count, use_lanes):
                    parts[i] = cworkers[lane].commit_dev(src[i], n_loc)
            except BaseException as ex:     # noqa: BLE001 - re-raised on the main thread
                errs.append(ex)

        th = [threading.Thread(target=run, args=(lane,)) for lane in range(use_lanes)]
        for t_ in th:
            t_.start()
        return th, parts, errs

    def commits_finish(self, handle, all_parts=False):
        from distributed_plonk_amd.dispatcher import gather_points
        th, parts, errs = handle
        count = len(parts)
        for t_ in th:
            t_.join()
        if errs:
            raise errs[0]
        if not self.multi or self.sim:
            return parts if all_parts else parts[-1]
        w, q64 = self.w, self.q64
        flat = self.np.concatenate(parts)                                         # one collective for all partial points
        gathered = list(w.comm_allgather_host(flat, self.world)) if self.transport == "rccl" else gather_points(flat, None, self.dev)
        acc = [None] * count
        for p in gathered:                                                   # reduce(a + b) per commitment, on the host
            for i in range(count):
                pt = p[i * 3 * q64:(i + 1) * 3 * q64]
                acc[i] = pt if acc[i] is None else w.g1_add(acc[i], pt)
        return acc[-1]

    def _commit_phase(self, t_in):
        for x in set(self.workers) | set(self._extra_contexts()):       # the transforms may have run on contexts of their own (headline.unoverlapped_roofline)
            x.sync()
        # (running the commitments concurrently with the transforms instead was measured: 977 vs 987 ms per step, not worth
        #  distorting the per-launch NTT timings the roofline is computed from)
        t_mid = time.perf_counter()
        res = self.commits_finish(self.commits_start(N_MSM))
        self.phase["ntt"] += t_mid - t_in
        self.phase["msm"] += time.perf_counter() - t_mid
        return res

    def _overlapped_finish(self, t_in, handle):
        """--overlap-phases: the commitment threads were started before the first transform was issued"""
        for x in (self.step_workers or [self.wt]):
            x.sync()
        t_mid = time.perf_counter()
        res = self.commits_finish(handle)
        self.phase["ntt"] += t_mid - t_in                    # the transforms, with commitments running beside them
        self.phase["msm"] += time.perf_counter() - t_mid     # what was left of the commitments after the last transform
        return res

    def step_ref2d(self):
        t_in = time.perf_counter()
        overlap = self.overlap or self.overlap_multi
        handle = self.commits_start(N_MSM) if overlap else None
        for i in range(N_NTT_SMALL):
            self.ntt(i % self.n_lanes, self.buf_n[i % len(self.buf_n)], self.n, True, False, False)
        for i in range(self.nbig - 1):
            self.coset_fft_8n(i % self.n_lanes, i)
        if self.nbig:
            self.ntt(0, self.buf_m[0], self.m, True, True, True)
        if overlap:
            return self._overlapped_finish(t_in, handle)
        return self._commit_phase(t_in)

    def step_classes(self):
        t_in = time.perf_counter()
        c, w, n, m, sim, transport = self.cls, self.w, self.n, self.m, self.sim, self.transport
        G = self.S
        mL = m // G
        L = n // G
        for i in range(N_NTT_SMALL):       # size-n iFFT by residue class (class_prover.py): n values folded onto n/G points, all-gather, interleave
            pair = c["bn"][i % len(c["bn"])]
            w.coset_eval_dev(pair[0].ptr, n, L, c["shift_n"], c["cls_mine"].ptr)
            self._allgather(w, c["cls_mine"].ptr, c["cls_all"].ptr, L * 32)
            w.class_interleave_dev(c["cls_all"].ptr, G, L, True, c["inv_n"], pair[1].ptr)
            pair[0], pair[1] = pair[1], pair[0]
        for i in range(N_NTT_BIG - 1):
            w.coset_eval_dev(c["polys"][i % len(c["polys"])].ptr, self.poly_len, mL, c["shift"], c["out"].ptr)
        # quotient coefficients: this class's additive share of every coefficient, summed across ranks, then replicated
        w.coset_interp_dev(c["out"].ptr, mL, c["shift"], c["inv_g"], 0, m, c["contrib"].ptr)
        if sim:
            self.sim_alltoall(w, c["contrib"].ptr, c["recv"].ptr, mL * 32)
        elif transport == "rccl":
            w.comm_alltoall_dev(c["contrib"].ptr, c["recv"].ptr, mL * 32)
        else:
            self.torch_comm.all_to_all_dev(c["contrib"].ptr, c["recv"].ptr, mL * 32)
        w.poly_lincomb_dev([(c["recv"].ptr + p_ * mL * 32, mL) for p_ in range(G)], c["ones"], c["mine"].ptr, mL)
        self._allgather(w, c["mine"].ptr, c["quot"].ptr, mL * 32)
        return self._commit_phase(t_in)

    def _allgather(self, w, d_send, d_recv, nbytes):
        if self.sim:
            self.sim_allgather(w, d_send, d_recv, nbytes)
        elif self.transport == "rccl":
            w.comm_allgather_dev(d_send, d_recv, nbytes)
        else:
            self.torch_comm.all_gather_dev(d_send, d_recv, nbytes)

    # ------------------------------------------------------------------------------------------------ helpers of the legs
    def dev_sync(self):
        if not self.emulated:
            self.torch.cuda.synchronize()

    def full_sync(self):
        for x in set(self.workers) | set(self._extra_contexts()):
            x.sync()
        self.dev_sync()
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def every_rank(self, flag):
        if self.world == 1:
            return bool(flag)
        t_ = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device=self.dev)
        self.dist.all_reduce(t_, op=self.dist.ReduceOp.MIN)
        return bool(t_.item())

    def release_step_buffers(self):
        """the op-mix step's vectors (tens of GiB at 2^24) are not needed after the legs that use them; the class prover wants the room"""
        for pair in self.buf_n + self.buf_m:
            for b in pair:
                b.free()
        for b in self.polys + self.rows_compact + self.scal:
            b.free()
        cls = self.cls
        if cls is not None:
            for k_ in ("out", "contrib", "recv", "mine", "quot", "cls_mine", "cls_all"):
                if cls[k_] is not None:
                    cls[k_].free()
                    cls[k_] = None
            for b in cls["polys"] + [x for pair in cls["bn"] for x in pair]:
                b.free()
            cls["polys"], cls["bn"] = [], []
        del self.buf_n[:], self.buf_m[:], self.polys[:], self.rows_compact[:], self.scal[:]

    def close(self):
        if self.buf_n or self.buf_m:
            self.release_step_buffers()
        self.bases.free()
        for x in self.workers + self._extra_contexts():
            x.close()
        if self.dist.is_initialized():
            self.dist.destroy_process_group()
