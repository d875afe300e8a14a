"""The CPU baseline leg: the oracle (a C restatement of the reference's arkworks path; `kind: "port"`) timed on the GPU box's host cores.  A reported
baseline, not the optimisation target.  The reference builds ark-poly WITHOUT its "parallel" feature and ark-ec WITH it (Cargo.toml:31-34): its
NTTs are single-threaded, its MSM runs its windows on the rayon pool.

Round 6: `value` is MEASURED at the GPU line's own size — every op of the step run ONCE in full at n = 2^log_n (one iNTT(n),
one coset-NTT(8n), one commit(n); ~90 s of host time and ~10 GiB of host memory at 2^24) and combined with the per-proof op mix 7 / 26 / 13 —
`extrapolated: false`.  The 2^20 / 2^22 samples of rounds 4-5 stay beside it (`samples`, `fitted_exponent`): they are what the full-size figure
can be checked against, and what `value` falls back to (labelled, `extrapolated: true`) when the host lacks the memory or `--cpu-full-size off`.
The leg runs LAST in bench.py, after the GPU has been released, under the line's watchdog: a slow host costs this field, never the line."""
import math
import os
import time

from .common import N_MSM, N_NTT_BIG, N_NTT_SMALL


def fetch_bases(b):
    """the SRS the run committed against, copied to the host while the contexts still exist (the leg itself runs after they are gone)"""
    import ctypes as C
    from distributed_plonk_amd._ffi import check
    np, w = b.np, b.w
    hb = np.empty((b.n, 2 * b.q64), dtype=np.uint64)
    check(w.lib.plonk_memcpy_d2h(w.ctx, hb.ctypes.data_as(C.c_void_p), b.bases.ptr, hb.nbytes))
    return hb


def _sample(O, cid, ls, thr, hb, with_parallel_ntt, clock):
    """every op of the step ONCE at n = 2^ls -> seconds per op (single-threaded transforms, window-parallel commitment)"""
    ns = 1 << ls
    v = O.rand_fr(cid, 1, ns)
    vb = O.rand_fr(cid, 2, 8 * ns)

    def timed(fn):
        t = clock()
        fn()
        return clock() - t

    t = {}
    if with_parallel_ntt:
        t["ntt_par"] = timed(lambda: O.ntt(cid, v, True, False, threads=thr))
        t["ntt8_par"] = timed(lambda: O.ntt(cid, vb, False, True, threads=thr))
    t["ntt"] = timed(lambda: O.ntt(cid, v, True, False, threads=1))
    t["ntt8"] = timed(lambda: O.ntt(cid, vb, False, True, threads=1))
    del vb
    t["msm"] = timed(lambda: O.commit_polynomial(cid, hb[:ns], v, threads=thr))
    t["step"] = N_NTT_SMALL * t["ntt"] + N_NTT_BIG * t["ntt8"] + N_MSM * t["msm"]
    return t


def _host_can_hold(log_n):
    """the full-size pass holds the 8n-point vector twice (the oracle transforms a copy) plus the SRS and the n-point vectors"""
    need = (2 * 8 + 6) * (32 << log_n)
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        return True
    return avail > 1.25 * need


def cpu_baseline(args, cfg, host_bases, clock=time.perf_counter):
    """args: the run's arguments (curve, log_n, cpu_sample_log_n, cpu_sample_log_n2, cpu_full_size); cfg: unused keys tolerated; host_bases: (n, 2 * q64)
    u64 affine Montgomery bases on the host.  `clock` is injectable: tests/test_bench_guard.py drives the arithmetic below with a deterministic one."""
    from oracle import oracle as O
    n = 1 << args.log_n
    cid = O.CURVE_IDS[args.curve]
    thr = O.max_threads()
    ls1 = min(args.cpu_sample_log_n, args.log_n)
    ls2 = min(args.cpu_sample_log_n2, args.log_n) if args.cpu_sample_log_n2 else 0
    want_full = getattr(args, "cpu_full_size", "auto")
    t1 = _sample(O, cid, ls1, thr, host_bases, True, clock)
    t2 = _sample(O, cid, ls2, thr, host_bases, False, clock) if ls2 > ls1 else None
    lsm, tm = (ls2, t2) if t2 else (ls1, t1)               # the largest of the small samples
    full_note = None
    t3 = None
    if args.log_n > lsm:
        if want_full == "off":
            full_note = "--cpu-full-size off"
        elif want_full == "auto" and not _host_can_hold(args.log_n):
            full_note = f"the host lacks the memory for a 2^{args.log_n + 3}-point oracle transform"
        else:
            t3 = _sample(O, cid, args.log_n, thr, host_bases, False, clock)
    desc = lambda l_, t_: {"log_n": l_, "constraints_per_s": round((1 << l_) / t_["step"], 1), "s_per_step": round(t_["step"], 2),
                           "iNTT_n_1_thread_ms": round(t_["ntt"] * 1e3, 1), "coset_NTT_8n_1_thread_ms": round(t_["ntt8"] * 1e3, 1),
                           f"commit_n_{thr}_threads_ms": round(t_["msm"] * 1e3, 1)}
    ls, t = (args.log_n, t3) if t3 else (lsm, tm)           # what `value` is measured on
    t_step_par = N_NTT_SMALL * t1["ntt_par"] + N_NTT_BIG * t1["ntt8_par"] + N_MSM * t1["msm"]
    cpu = {"value": round((1 << ls) / t["step"], 1), "unit": "constraints/s", "cores": thr, "kind": "port",
           "sample": f"oracle (C restatement of ark-poly/ark-ec 0.3.0) at n=2^{ls}" + (" — the GPU line's own size" if ls == args.log_n else "") +
                     f", each op of the step run ONCE in full and combined "
                     f"with the per-proof op mix 7/26/13: iNTT(n) {t['ntt']*1e3:.0f} ms and coset-NTT(8n) {t['ntt8']*1e3:.0f} ms on 1 thread "
                     f"(the reference's ark-poly has no `parallel` feature, Cargo.toml:31), commit(n) {t['msm']*1e3:.0f} ms on {thr} threads "
                     f"(ark-ec `parallel`: windows on the rayon pool)" +
                     ("" if ls == args.log_n else f".  `value` is this 2^{ls} measurement, NOT the GPU line's size: see value_at_bench_size"),
           "samples": [desc(ls1, t1)] + ([desc(ls2, t2)] if t2 else []) + ([desc(args.log_n, t3)] if t3 else []),
           "all_threads_ntt": {"log_n": ls1, "value": round((1 << ls1) / t_step_par, 1), "iNTT_n_ms": round(t1["ntt_par"] * 1e3, 1),
                               "coset_NTT_8n_ms": round(t1["ntt8_par"] * 1e3, 1),
                               "note": "the oracle's OpenMP NTT on every host thread - faster than the reference's build would be"},
           "host_cores_online": os.cpu_count(),
           "compares_with": "op_mix (the same 7 / 26 / 13 operation mix on the GPU); a proof adds the quotient, grand product and O(n) rounds on both sides"}
    e = None
    if t2:
        e = math.log(t2["step"] / t1["step"]) / math.log(2.0 ** (ls2 - ls1))
        cpu["fitted_exponent"] = round(e, 4)
    if args.log_n > lsm:
        # what rounds 4-5 reported instead of a measurement, kept as a cross-check of it: (a) the exponent the two small samples give (step time ~ n^e);
        # (b) the operation counts of radix-2 NTT (n log n) and Pippenger (linear in n at a fixed window) from the larger small sample
        up = 1 << (args.log_n - lsm)
        t_ops = up * (N_NTT_SMALL * tm["ntt"] * args.log_n / lsm + N_NTT_BIG * tm["ntt8"] * (args.log_n + 3) / (lsm + 3) + N_MSM * tm["msm"])
        ext = {"log_n": args.log_n, "unit": "constraints/s", "by_operation_counts": {"value": round(n / t_ops, 1), "s_per_step": round(t_ops, 1)}}
        if t2:
            t_fit = t2["step"] * (2.0 ** (args.log_n - ls2)) ** e
            ext["fitted_exponent"] = round(e, 4)
            ext["by_fitted_exponent"] = {"value": round(n / t_fit, 1), "s_per_step": round(t_fit, 1)}
        ext["value"] = ext["by_fitted_exponent"]["value"] if t2 else ext["by_operation_counts"]["value"]
        ext["note"] = (f"from the 2^{ls1}" + (f" and 2^{ls2}" if t2 else "") + " samples: " +
                       ("a cross-check of the full-size measurement above, not the reported figure" if t3 else
                        f"EXTRAPOLATED, not measured at 2^{args.log_n} ({full_note})"))
        cpu["extrapolated_to_bench_size"] = ext
        if t3:
            cpu["value_at_bench_size"], cpu["extrapolated"] = cpu["value"], False
        else:
            cpu["value_at_bench_size"], cpu["extrapolated"] = ext["value"], True
            cpu["full_size_skipped"] = full_note
    else:
        cpu["value_at_bench_size"], cpu["extrapolated"] = cpu["value"], False
    return cpu
