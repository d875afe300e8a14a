import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np
from distributed_plonk_amd.worker import PlonkWorker
from oracle import oracle as O
w = PlonkWorker(0, 0, "bn254")
log_n = int(sys.argv[1])
v = O.rand_fr(0, 5, 1 << log_n)
want = {}
for inv, coset in [(False, False), (True, True)]:
    want[(inv, coset)] = O.ntt(0, v, inv, coset, threads=64)
for rep in range(3):
    for inv, coset in [(False, False), (True, True)]:
        got = w.ntt(v, inv, coset)
        bad = np.nonzero((got != want[(inv, coset)]).any(axis=1))[0]
        print(os.environ.get("TAG", ""), log_n, inv, coset, "rep", rep, "OK" if len(bad) == 0 else f"BAD {len(bad)} first {bad[:6]}", flush=True)
