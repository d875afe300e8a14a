"""Do the transforms and the commitments of a step gain anything from running CONCURRENTLY instead of one phase after the other?

    python tools/overlap_probe.py [LOGN] [N_TRANSFORMS] [N_COMMITS_PER_LANE]

bench.py runs a step as two phases — every transform, one sync, then the 13 commitments on two contexts (benchlib/run.py: _commit_phase) — because
round 1 measured the concurrent form at -1 % (977 vs 987 ms per step).  The kernels have changed since (the NTT pass leaves 14-25 % of a SIMD's time
without a ready wave, profiles/r04_ntt_stall_attribution.txt; the bucket accumulation is persistent), so the question is asked again with the shipped
library: K zero-padded 8n coset FFTs on one context, 2 x M commitments on two more, timed (a) transforms alone, (b) commitments alone, (c) all three
host threads at once.  No torch (the import costs a fresh box a minute): ctypes and numpy only.  Prints one JSON line."""
import json
import os
import sys
import threading
import time

sys.modules.setdefault("torch", None)          # _ffi.lib() imports torch when it can, to order the HIP runtimes; not needed here
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_plonk_amd import fr as _fr                      # noqa: E402
from distributed_plonk_amd.worker import PlonkWorker             # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4
curve = os.environ.get("CURVE", "bn254")
n, m = 1 << log_n, 8 << log_n
t0 = time.perf_counter()
wt = PlonkWorker(curve=curve)                                    # the transforms' context
wc = [PlonkWorker(curve=curve) for _ in range(2)]                # the two commitment contexts of bench.py
bases = wt.alloc(n * 16 * wt.q64)
wt.synth_bases(0x5EED, 0, n, bases.ptr)
for x in [wt] + wc:
    x.init_dev(bases.ptr, n, n, m)
    x.sync()
gen = _fr.FIELDS[curve].to_limbs(_fr.FIELDS[curve].generator)
polys = [wt.alloc((n + 3) * 32) for _ in range(2)]
scal = [wt.alloc(n * 32) for _ in range(4)]
for i, b in enumerate(polys):
    wt.synth_fr(0xC0EFF + i, b.ptr, n + 3)
for i, b in enumerate(scal):
    wt.synth_fr(0x5CA1A5 + i, b.ptr, n)
out = wt.alloc(m * 32)
wt.sync()
setup_s = time.perf_counter() - t0


def transforms():
    for i in range(K):
        wt.coset_eval_dev(polys[i % 2].ptr, n + 3, m, gen, out.ptr)
    wt.sync()


def commits(lane):
    for i in range(M):
        wc[lane].commit_many_dev([(scal[(2 * i + lane) % 4].ptr, n), (scal[(2 * i + lane + 1) % 4].ptr, n)])
    wc[lane].sync()


def timed(fns):
    th = [threading.Thread(target=f, args=a) for f, a in fns]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    return (time.perf_counter() - t) * 1e3


both = [(commits, (0,)), (commits, (1,))]
timed([(transforms, ())] + both)                                 # warm-up: planes, workspaces, first launches
res = {"transforms_ms": [], "commits_ms": [], "concurrent_ms": []}
for _ in range(int(os.environ.get("REPS", "3"))):
    res["transforms_ms"].append(round(timed([(transforms, ())]), 2))
    res["commits_ms"].append(round(timed(both), 2))
    res["concurrent_ms"].append(round(timed([(transforms, ())] + both), 2))
best = {k: min(v) for k, v in res.items()}
res.update(log_n=log_n, curve=curve, transforms=K, commits=4 * M, setup_s=round(setup_s, 1),
           serial_ms=round(best["transforms_ms"] + best["commits_ms"], 2), concurrent_best_ms=best["concurrent_ms"],
           concurrent_over_serial=round(best["concurrent_ms"] / (best["transforms_ms"] + best["commits_ms"]), 4))
print(json.dumps(res))
