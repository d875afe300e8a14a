"""Time one MSM (commit) at n = 2^LOGN on the GPU with the per-kernel profiler: python tools/msm_only.py [LOGN]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_plonk_amd.worker import PlonkWorker

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log_n
w = PlonkWorker(curve=os.environ.get("CURVE", "bn254"))
q = w.q64
bases = w.alloc(n * 16 * q)
w.synth_bases(0x5EED, int(os.environ.get("MSM_TILE", "0")), n, bases.ptr)       # MSM_TILE=2048: 2^11 points tiled (gathers from L2)
if os.environ.get("MSM_PRECOMPUTE"):
    w.set_option("msm_precompute", int(os.environ["MSM_PRECOMPUTE"]))       # fixed-base window table (takes effect at init)
    w.set_option("msm_table_c", int(os.environ.get("MSM_TAB_C", "0")))
    w.set_option("msm_table_sets", int(os.environ.get("MSM_TAB_G", "0")))
t = time.perf_counter(); w.init_dev(bases.ptr, n, 0, 0); w.sync(); print("init ms", (time.perf_counter() - t) * 1e3)
if os.environ.get("MSM_FUSED"):
    w.set_option("msm_fused_y3", int(os.environ["MSM_FUSED"]))
if os.environ.get("MSM_PERSIST"):
    w.set_option("msm_acc_persist", int(os.environ["MSM_PERSIST"]))
if os.environ.get("MSM_REDUCE_GRID"):
    w.set_option("msm_reduce_grid", int(os.environ["MSM_REDUCE_GRID"]))     # window reduction as row / column tree sums + bit sums (experiment)
if os.environ.get("MSM_WINDOW"):
    w.set_option("msm_window", int(os.environ["MSM_WINDOW"]))
sc = w.alloc(n * 32)
w.synth_fr(7, sc.ptr, n)
w.commit_dev(sc.ptr, n)
w.profile_enable(True); w.profile_reset()
t = time.perf_counter()
for _ in range(3):
    w.commit_dev(sc.ptr, n)
w.sync()
print("commit ms", (time.perf_counter() - t) / 3 * 1e3)
for k in ("msm_digits_kernel", "msm_sort", "msm_bucket_order", "msm_accumulate_kernel", "msm_heavy", "msm_accumulate_redo_kernel", "msm_reduce"):
    ms, cnt = w.profile_get(k)
    print(f"  {k:28s} {ms / max(cnt, 1):8.3f} ms x {cnt}")
