"""The CPU baseline leg: the oracle (a C restatement of the reference's arkworks path; `kind: "port"`) timed on the GPU box's host cores on
bounded samples of the same workload.  A reported baseline, not the optimisation target.  The reference builds ark-poly WITHOUT its "parallel"
feature and ark-ec WITH it (Cargo.toml:31-34): its NTTs are single-threaded, its MSM runs its windows on the rayon pool.  `value` is that
configuration at the LARGEST measured sample; `value_at_bench_size` carries it to the GPU line's size with the exponent fitted through the
two samples (a 2^20 figure beside a 2^24 line invites the wrong comparison)."""
import math
import os
import time

from .common import N_MSM, N_NTT_BIG, N_NTT_SMALL


def _sample(b, O, cid, ls, thr, with_parallel_ntt):
    """every op of the step ONCE at n = 2^ls -> seconds per op (single-threaded transforms, window-parallel commitment)"""
    import ctypes as C
    from distributed_plonk_amd._ffi import check
    np, w = b.np, b.w
    ns = 1 << ls
    v = O.rand_fr(cid, 1, ns)
    vb = O.rand_fr(cid, 2, 8 * ns)
    hb = np.empty((ns, 2 * b.q64), dtype=np.uint64)
    check(w.lib.plonk_memcpy_d2h(w.ctx, hb.ctypes.data_as(C.c_void_p), b.bases.ptr, hb.nbytes))

    def timed(fn):
        t = time.perf_counter()
        fn()
        return time.perf_counter() - t

    t = {}
    if with_parallel_ntt:
        t["ntt_par"] = timed(lambda: O.ntt(cid, v, True, False, threads=thr))
        t["ntt8_par"] = timed(lambda: O.ntt(cid, vb, False, True, threads=thr))
    t["ntt"] = timed(lambda: O.ntt(cid, v, True, False, threads=1))
    t["ntt8"] = timed(lambda: O.ntt(cid, vb, False, True, threads=1))
    t["msm"] = timed(lambda: O.commit_polynomial(cid, hb, v, threads=thr))
    t["step"] = N_NTT_SMALL * t["ntt"] + N_NTT_BIG * t["ntt8"] + N_MSM * t["msm"]
    return t


def cpu_baseline(b):
    from oracle import oracle as O
    args, n = b.args, b.n
    cid = O.CURVE_IDS[args.curve]
    thr = O.max_threads()
    ls1 = min(args.cpu_sample_log_n, args.log_n)
    ls2 = min(args.cpu_sample_log_n2, args.log_n) if args.cpu_sample_log_n2 else 0
    t1 = _sample(b, O, cid, ls1, thr, with_parallel_ntt=True)
    t2 = _sample(b, O, cid, ls2, thr, with_parallel_ntt=False) if ls2 > ls1 else None
    ls, t = (ls2, t2) if t2 else (ls1, t1)
    ns = 1 << ls
    desc = lambda l_, t_: {"log_n": l_, "constraints_per_s": round((1 << l_) / t_["step"], 1), "s_per_step": round(t_["step"], 2),
                           "iNTT_n_1_thread_ms": round(t_["ntt"] * 1e3, 1), "coset_NTT_8n_1_thread_ms": round(t_["ntt8"] * 1e3, 1),
                           f"commit_n_{thr}_threads_ms": round(t_["msm"] * 1e3, 1)}
    t_step_par = N_NTT_SMALL * t1["ntt_par"] + N_NTT_BIG * t1["ntt8_par"] + N_MSM * t1["msm"]
    cpu = {"value": round(ns / t["step"], 1), "unit": "constraints/s", "cores": thr, "kind": "port",
           "sample": f"oracle (C restatement of ark-poly/ark-ec 0.3.0) at n=2^{ls}, each op of the step run ONCE in full and combined "
                     f"with the per-proof op mix 7/26/13: iNTT(n) {t['ntt']*1e3:.0f} ms and coset-NTT(8n) {t['ntt8']*1e3:.0f} ms on 1 thread "
                     f"(the reference's ark-poly has no `parallel` feature, Cargo.toml:31), commit(n) {t['msm']*1e3:.0f} ms on {thr} threads "
                     f"(ark-ec `parallel`: windows on the rayon pool).  `value` is this 2^{ls} measurement, NOT the GPU line's size: see value_at_bench_size",
           "samples": [desc(ls1, t1)] + ([desc(ls2, t2)] if t2 else []),
           "all_threads_ntt": {"log_n": ls1, "value": round((1 << ls1) / t_step_par, 1), "iNTT_n_ms": round(t1["ntt_par"] * 1e3, 1),
                               "coset_NTT_8n_ms": round(t1["ntt8_par"] * 1e3, 1),
                               "note": "the oracle's OpenMP NTT on every host thread - faster than the reference's build would be"},
           "host_cores_online": os.cpu_count()}
    if args.log_n > ls:
        # the GPU line's size.  (a) with the exponent the two measured samples give (step time ~ n^e); (b) with the operation counts of
        # radix-2 NTT (n log n) and Pippenger (linear in n at a fixed window) from the larger sample.  BASELINE.md §3 allows a labelled
        # extrapolation; both are estimates, not measurements.
        up = 1 << (args.log_n - ls)
        t_ops = up * (N_NTT_SMALL * t["ntt"] * args.log_n / ls + N_NTT_BIG * t["ntt8"] * (args.log_n + 3) / (ls + 3) + N_MSM * t["msm"])
        ext = {"log_n": args.log_n, "unit": "constraints/s", "by_operation_counts": {"value": round(n / t_ops, 1), "s_per_step": round(t_ops, 1)}}
        if t2:
            e = math.log(t2["step"] / t1["step"]) / math.log(2.0 ** (ls2 - ls1))
            t_fit = t2["step"] * (2.0 ** (args.log_n - ls2)) ** e
            ext["fitted_exponent"] = round(e, 4)
            ext["by_fitted_exponent"] = {"value": round(n / t_fit, 1), "s_per_step": round(t_fit, 1)}
            cpu["fitted_exponent"] = round(e, 4)
            cpu["value_at_bench_size"] = round(n / t_fit, 1)
        else:
            cpu["value_at_bench_size"] = round(n / t_ops, 1)
        cpu["extrapolated"] = True
        ext["value"] = cpu["value_at_bench_size"]
        ext["note"] = (f"EXTRAPOLATED, not measured at 2^{args.log_n}: step time ~ n^e with e fitted through the measured 2^{ls1} and 2^{ls2} samples"
                       if t2 else f"EXTRAPOLATED from the 2^{ls} sample with the operation counts of radix-2 NTT and Pippenger: not measured at 2^{args.log_n}")
        cpu["extrapolated_to_bench_size"] = ext
    else:
        cpu["value_at_bench_size"], cpu["extrapolated"] = cpu["value"], False
    return cpu
