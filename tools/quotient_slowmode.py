"""Diagnosis of the quotient kernel's slow mode: the build of quotient.hip with the raised unroll budget ran 52 ms
standalone and 112 ms inside bench.py.  One process, the quotient kernel timed (HIP events) after each step of what bench.py does before
its quotient leg; run with PLONK_HIP_LIB pointing at the variant build and at the product build.
    python tools/quotient_slowmode.py [log_n]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributed_plonk_amd import fr as _fr
from distributed_plonk_amd.worker import PlonkWorker

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n, m = 1 << log_n, 8 << log_n
w = PlonkWorker(0, 0, "bn254")
w2 = PlonkWorker(0, 0, "bn254")
w.init(None, n, m)
vecs = [w.alloc(m * 32) for _ in range(25)]
for j, b in enumerate(vecs):
    w.synth_fr(0xABC + j, b.ptr, m)
out = w.alloc(m * 32)
ch = np.arange(32, dtype=np.uint64).reshape(8, 4) + 3
ptr = [b.ptr for b in vecs]


def q(tag):
    w.profile_enable(True)
    ts = []
    for it in range(3):
        w.profile_reset()
        t0 = time.perf_counter()
        w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr)
        w.sync()
        host = (time.perf_counter() - t0) * 1e3
        ts.append((round(w.profile_get("quotient_evals_kernel")[0], 2), round(host, 2)))
    w.profile_enable(False)
    print(f"{tag:58s} kernel/host ms: {ts}", flush=True)


q("fresh process (first call builds the 1/(x-1) plane)")
bases = w.alloc(n * 64)
w.synth_bases(0x5EED, 0, n, bases.ptr)
for x in (w, w2):
    x.init_dev(bases.ptr, n, n, m)
q("after init_dev (SRS -> limb form) on two contexts")
sc = w.alloc(n * 32)
w.synth_fr(5, sc.ptr, n)
w.commit_dev(sc.ptr, n)
q("after one 2^24 commitment (MSM kernels, incl. redo / heavy)")
w.commit_many_dev([(sc.ptr, n)] * 3)
w2.commit_many_dev([(sc.ptr, n)] * 2)
q("after batched commitments on both contexts")
p = w.alloc((n + 3) * 32)
w.synth_fr(7, p.ptr, n + 3)
g = _fr.FIELDS["bn254"].to_limbs(_fr.FIELDS["bn254"].generator)
w.coset_eval_dev(p.ptr, n + 3, m, g, out.ptr)
q("after a zero-padded 8n coset FFT")
a, b = w.alloc(n * 32), w.alloc(n * 32)
w.synth_fr(9, a.ptr, n)
w.ntt_dev(a.ptr, b.ptr, n, True, False)
w.ntt_dev(out.ptr, vecs[0].ptr, m, True, True)
w.synth_fr(0xABC, vecs[0].ptr, m)
q("after size-n and size-8n inverse transforms")
w.perm_product_dev(ptr[0:5], ptr[5], ptr[6], ch[0], ch[1], 16, out.ptr) if False else None
w.poly_eval_dev(p.ptr, n, ch[3])
q("after poly_eval")
import threading
th = [threading.Thread(target=lambda x=x: x.commit_many_dev([(sc.ptr, n)] * 2)) for x in (w, w2)]
[t.start() for t in th]; [t.join() for t in th]
q("after commitments from two host threads")
