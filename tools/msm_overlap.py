"""Do two independent commitments overlap when issued from two contexts (two HIP streams, two host threads)?
python tools/msm_overlap.py [LOGN]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_plonk_amd.worker import PlonkWorker

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log_n
K = 6
ws = [PlonkWorker(), PlonkWorker()]
q = ws[0].q64
bases = ws[0].alloc(n * 16 * q)
ws[0].synth_bases(0x5EED, 0, n, bases.ptr)
for w in ws:
    w.init_dev(bases.ptr, n, 0, 0)
sc = ws[0].alloc(n * 32)
ws[0].synth_fr(7, sc.ptr, n)
ws[0].sync()
for w in ws:
    w.commit_dev(sc.ptr, n)

t = time.perf_counter()
for _ in range(2 * K):
    ws[0].commit_dev(sc.ptr, n)
serial = (time.perf_counter() - t) * 1e3
print(f"serial   {2 * K} commits: {serial:8.1f} ms  ({serial / (2 * K):.2f} ms each)")


def run(w):
    for _ in range(K):
        w.commit_dev(sc.ptr, n)


t = time.perf_counter()
th = [threading.Thread(target=run, args=(w,)) for w in ws]
for x in th:
    x.start()
for x in th:
    x.join()
par = (time.perf_counter() - t) * 1e3
print(f"2 lanes  {2 * K} commits: {par:8.1f} ms  ({par / (2 * K):.2f} ms each)  speed-up {serial / par:.3f}")
