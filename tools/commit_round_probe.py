#!/usr/bin/env python3
"""Why does round 3's commitment phase (5 split-quotient chunks) take ~119 ms in the proof when round 1's (5 wire polynomials, same lengths) takes ~102?
After two proofs at 2^log_n, the prover's own two-lane `_commit_many` is timed on the polynomials the last proof left in HBM: the five wire polynomials, the five
quotient chunks, and the quotient chunks COPIED into five separate buffers; alternating, wall clock around synchronised calls.
    python tools/commit_round_probe.py [log_n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from distributed_plonk_amd.prover import Prover  # noqa: E402
from distributed_plonk_amd.synthetic import SyntheticInstance  # noqa: E402
from distributed_plonk_amd.worker import PlonkWorker  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log_n
w, w2 = PlonkWorker(curve="bn254"), PlonkWorker(curve="bn254")
inst = SyntheticInstance(w, log_n, seed=0xC1AC, num_inputs=3, tau=12345678901234567890, helpers=[w2])
pv = Prover(w, log_n, commit_helper=w2)
pv.load_key_dev(inst.sel_ptrs, inst.sig_ptrs, inst.k)
pub = inst.public_inputs()
consts = np.arange(64, dtype=np.uint64).reshape(16, 4) + 11
bl = {"wires": consts[5:15].reshape(5, 2, 4), "perm": consts[12:15]}
for _ in range(2):
    t0 = time.perf_counter()
    pv.prove_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, inst.d_pi.ptr, bl, pv.fiat_shamir(pub), check_degree=True)
    print("proof ms", round((time.perf_counter() - t0) * 1e3, 1), {k: round(v, 1) for k, v in pv.timings.items()}, flush=True)
lp = pv.last_polys
wires, split = list(lp["wire_polys"]), list(lp["split_quot_polys"])
copies = []
for ptr, ln in split:
    b = w.alloc(ln * 32)
    w.memcpy_d2d(b.ptr, ptr, ln * 32)
    copies.append((b.ptr, ln))
w.sync(); w2.sync()


def timed(items):
    w.sync(); w2.sync()
    t0 = time.perf_counter()
    pv._commit_many(items)
    w.sync(); w2.sync()
    return (time.perf_counter() - t0) * 1e3


for rep in range(3):
    print(f"rep {rep}: wires {timed(wires):7.2f} ms   quotient chunks in place {timed(split):7.2f} ms   quotient chunks copied out {timed(copies):7.2f} ms   "
          f"lens {[ln - n for _, ln in wires]} / {[ln - n for _, ln in split]}", flush=True)
