"""A/B of fixed-base table shapes in ONE process (box clocks drift between processes): one context per shape on the same bases, commits
interleaved round-robin, per-shape host time and per-phase device time.   python tools/msm_table_ab.py LOGN "mode:c:sets" ...   (0:0:0 = plain)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_plonk_amd.worker import PlonkWorker

log_n = int(sys.argv[1])
shapes = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]] or [(0, 0, 0), (2, 20, 1)]
n = 1 << log_n
ws = []
w0 = PlonkWorker(curve=os.environ.get("CURVE", "bn254"))
bases = w0.alloc(n * 16 * w0.q64)
w0.synth_bases(0x5EED, 0, n, bases.ptr)
sc = w0.alloc(n * 32)
w0.synth_fr(7, sc.ptr, n)
for i, (mode, c, g) in enumerate(shapes):
    w = w0 if i == 0 else PlonkWorker(curve=os.environ.get("CURVE", "bn254"))
    w.set_option("msm_precompute", mode)
    w.set_option("msm_table_c", c)
    w.set_option("msm_table_sets", g)
    t = time.perf_counter(); w.init_dev(bases.ptr, n, 0, 0); w.sync()
    print(f"shape {mode}:{c}:{g} init {(time.perf_counter() - t) * 1e3:.0f} ms")
    w.commit_dev(sc.ptr, n)
    w.profile_enable(True); w.profile_reset()
    ws.append(w)
rounds, reps = int(os.environ.get("ROUNDS", "3")), int(os.environ.get("REPS", "3"))
KEYS = ("msm_digits_kernel", "msm_sort", "msm_bucket_order", "msm_accumulate_kernel", "msm_heavy", "msm_accumulate_redo_kernel", "msm_reduce")
host = [0.0] * len(ws)
dev = [[0.0] * len(KEYS) for _ in ws]
for _ in range(rounds):                       # blocks of `reps` commits per shape, shapes alternating; the profiler is process-wide, so it is read per block
    for i, w in enumerate(ws):
        w.profile_reset()
        t = time.perf_counter()
        for _ in range(reps):
            w.commit_dev(sc.ptr, n)
        w.sync()
        host[i] += time.perf_counter() - t
        for j, k in enumerate(KEYS):
            ms, cnt = w.profile_get(k)
            dev[i][j] += ms
tot = rounds * reps
for i, w in enumerate(ws):
    parts = "  ".join(f"{k.replace('msm_', '').replace('_kernel', '')} {dev[i][j] / tot:.2f}" for j, k in enumerate(KEYS))
    print(f"shape {shapes[i][0]}:{shapes[i][1]}:{shapes[i][2]}  commit {host[i] / tot * 1e3:7.2f} ms   " + parts)
