"""The CPU baseline leg: the oracle (a C restatement of the reference's arkworks path; `kind: "port"`) timed on the GPU box's host cores on a
bounded sample of the same workload.  A reported baseline, not the optimisation target.  The reference builds ark-poly WITHOUT its "parallel"
feature and ark-ec WITH it (Cargo.toml:31-34): its NTTs are single-threaded, its MSM runs its windows on the rayon pool.  `value` is that
configuration; the all-threads OpenMP NTT of the oracle is reported beside it."""
import os
import time

from .common import N_MSM, N_NTT_BIG, N_NTT_SMALL


def cpu_baseline(b):
    import ctypes as C
    from distributed_plonk_amd._ffi import check
    from oracle import oracle as O
    args, np, w, n = b.args, b.np, b.w, b.n
    cid = O.CURVE_IDS[args.curve]
    ls = min(args.cpu_sample_log_n, args.log_n)
    ns = 1 << ls
    thr = O.max_threads()
    v = O.rand_fr(cid, 1, ns)
    vb = O.rand_fr(cid, 2, 8 * ns)
    hb = np.empty((ns, 2 * b.q64), dtype=np.uint64)
    check(w.lib.plonk_memcpy_d2h(w.ctx, hb.ctypes.data_as(C.c_void_p), b.bases.ptr, hb.nbytes))

    def timed(fn):
        t = time.perf_counter()
        fn()
        return time.perf_counter() - t

    t_ntt_par = timed(lambda: O.ntt(cid, v, True, False, threads=thr))
    t_ntt8_par = timed(lambda: O.ntt(cid, vb, False, True, threads=thr))
    t_ntt_1 = timed(lambda: O.ntt(cid, v, True, False, threads=1))
    t_ntt8_1 = timed(lambda: O.ntt(cid, vb, False, True, threads=1))
    t_msm = timed(lambda: O.commit_polynomial(cid, hb, v, threads=thr))
    t_step = N_NTT_SMALL * t_ntt_1 + N_NTT_BIG * t_ntt8_1 + N_MSM * t_msm
    t_step_par = N_NTT_SMALL * t_ntt_par + N_NTT_BIG * t_ntt8_par + N_MSM * t_msm
    cpu = {"value": round(ns / t_step, 1), "unit": "constraints/s", "cores": thr, "kind": "port",
           "sample": f"oracle (C restatement of ark-poly/ark-ec 0.3.0) at n=2^{ls}, each op of the step run ONCE in full and combined "
                     f"with the per-proof op mix 7/26/13: iNTT(n) {t_ntt_1*1e3:.0f} ms and coset-NTT(8n) {t_ntt8_1*1e3:.0f} ms on 1 thread "
                     f"(the reference's ark-poly has no `parallel` feature, Cargo.toml:31), commit(n) {t_msm*1e3:.0f} ms on {thr} threads "
                     f"(ark-ec `parallel`: windows on the rayon pool).  `value` is this 2^{ls} measurement; the estimate for the GPU line's "
                     f"size is in extrapolated_to_bench_size",
           "all_threads_ntt": {"value": round(ns / t_step_par, 1), "iNTT_n_ms": round(t_ntt_par * 1e3, 1), "coset_NTT_8n_ms": round(t_ntt8_par * 1e3, 1),
                               "note": "the oracle's OpenMP NTT on every host thread - faster than the reference's build would be"},
           "host_cores_online": os.cpu_count()}
    if args.log_n > ls:
        # labelled extrapolation to the GPU line's size (BASELINE.md §3 allows it): radix-2 NTT cost per element grows with log2 of
        # the size ((log n + 3) / (ls + 3) for the 8n transforms, log n / ls for the n ones); Pippenger's cost per point is taken as
        # constant (it falls slightly with n: larger windows).  An estimate, not a measurement.
        up = 1 << (args.log_n - ls)
        t_ext = up * (N_NTT_SMALL * t_ntt_1 * args.log_n / ls + N_NTT_BIG * t_ntt8_1 * (args.log_n + 3) / (ls + 3) + N_MSM * t_msm)
        cpu["extrapolated_to_bench_size"] = {"log_n": args.log_n, "value": round(n / t_ext, 1), "unit": "constraints/s", "s_per_step": round(t_ext, 1),
                                             "note": f"EXTRAPOLATED from the 2^{ls} sample above with the operation counts of radix-2 NTT (n log n) and "
                                                     f"Pippenger (linear in n at a fixed window): not measured at 2^{args.log_n}"}
    return cpu
