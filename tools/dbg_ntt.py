import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from distributed_plonk_amd.worker import PlonkWorker
from oracle import oracle as O
w = PlonkWorker(0, 0, "bn254")
for log_n in [int(a) for a in sys.argv[1:]]:
    v = O.rand_fr(0, 5, 1 << log_n)
    for inv, coset in [(False, False), (False, True), (True, True)]:
        got = w.ntt(v, inv, coset)
        t = time.time()
        want = O.ntt(0, v, inv, coset, threads=64)
        bad = np.nonzero((got != want).any(axis=1))[0]
        print(log_n, inv, coset, "OK" if len(bad) == 0 else f"BAD {len(bad)} first {bad[:8]} last {bad[-3:]}", round(time.time() - t, 1), "s oracle", flush=True)
