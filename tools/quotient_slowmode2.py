"""bisect: does a second context (second HIP stream + its tables) switch the variant quotient kernel into its slow mode?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributed_plonk_amd.worker import PlonkWorker
mode = sys.argv[1]
log_n = 24
n, m = 1 << log_n, 8 << log_n
w2 = PlonkWorker(0, 0, "bn254") if mode == "second_first" else None
w = PlonkWorker(0, 0, "bn254")
if mode.startswith("first_then_second"):
    w2 = PlonkWorker(0, 0, "bn254")
w.init(None, n, m)
seed0 = 0xABC if "seedabc" in mode else 100
bufs = [w.alloc(m * 32) for _ in range(25)]
for j, b in enumerate(bufs):
    w.synth_fr(seed0 + j, b.ptr, m)
out = w.alloc(m * 32)
ch = np.arange(32, dtype=np.uint64).reshape(8, 4) + (3 if "ch3" in mode else 5)
ptr = [b.ptr for b in bufs]
w.profile_enable(True)
def q(tag):
    ts = []
    for it in range(3):
        w.profile_reset()
        w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr)
        w.sync()
        ts.append(round(w.profile_get("quotient_evals_kernel")[0], 2))
    print(mode, tag, ts, flush=True)
q("first")
if mode == "second_after":
    w2 = PlonkWorker(0, 0, "bn254")
    q("after creating a second context")
    w2.close()
    q("after closing it")
