"""Same-process A/B of two formulations of the quotient kernel (option quotient_fuse), alternating: python tools/quotient_ab.py 24 6 8 [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributed_plonk_amd.worker import PlonkWorker
log_n, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
w = PlonkWorker(0, 0, os.environ.get("QUOT_CURVE", "bn254"))
n, m = 1 << log_n, 8 << log_n
w.init(None, n, m)
bufs = [w.alloc(m * 32) for _ in range(25)]
for j, b in enumerate(bufs):
    w.synth_fr(100 + j, b.ptr, m)
out = w.alloc(m * 32)
ch = np.arange(32, dtype=np.uint64).reshape(8, 4) + 5
ptr = [b.ptr for b in bufs]
w.profile_enable(True)
alg = 27 * 32 * m
res = {va: [], vb: []}
for rep in range(reps + 1):
    for v in (va, vb):
        w.set_option("quotient_fuse", v)
        w.profile_reset()
        w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr)
        w.sync()
        ms, _ = w.profile_get("quotient_evals_kernel")
        if rep:                                      # the first round warms tables and clocks
            res[v].append(ms)
for v in (va, vb):
    a = res[v]
    print(f"2^{log_n} points x 8, quotient_fuse={v}: " + " ".join(f"{x:.2f}" for x in a) + f"  | median {sorted(a)[len(a) // 2]:.2f} ms = {alg / sorted(a)[len(a) // 2] / 1e6 / 8000:.4f} of 8 TB/s")
