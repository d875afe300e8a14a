import sys, time, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from distributed_plonk_amd.worker import PlonkWorker
w = PlonkWorker(0, 0, "bn254")
for log_n in [int(x) for x in sys.argv[1:]]:
    n = 1 << log_n
    a = w.alloc(n * 32); b = w.alloc(n * 32)
    w.synth_fr(1, a.ptr, n)
    w.profile_enable(True)
    for it in range(3):
        w.profile_reset()
        w.ntt_dev(a.ptr, b.ptr, n, False, True)
        w.ntt_dev(b.ptr, a.ptr, n, False, True)
        w.sync()
    out = {k: w.profile_get(k) for k in ["ntt_pass_kernel"] + [f"ntt_pass_kernel<{i}>" for i in range(6, 10)]}
    tot = out["ntt_pass_kernel"][0] / 2
    print(log_n, "NTT ms", round(tot, 3), "alg GB/s", round(64 * n / tot / 1e6, 1), {k: round(v[0] / max(v[1], 1), 4) for k, v in out.items() if v[1]})
    a.free(); b.free()
