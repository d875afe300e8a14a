import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from distributed_plonk_amd.worker import PlonkWorker
log_n = 26
n = 1 << log_n
w = PlonkWorker(curve="bn254")
bases = w.alloc(n * 64); w.synth_bases(0x5EED, 1 << 11, n, bases.ptr); w.init_dev(bases.ptr, n, 0, 0)
sc = w.alloc(n * 32); w.synth_fr(7, sc.ptr, n)
for sl in (26, 25, 24):
    w.set_option("msm_slice_log", sl)
    w.commit_dev(sc.ptr, n)
    w.profile_enable(True); w.profile_reset()
    t = time.perf_counter(); w.commit_dev(sc.ptr, n); w.sync(); dt = (time.perf_counter() - t) * 1e3
    print("slice_log", sl, "commit ms", round(dt, 1), {k: round(w.profile_get(k)[0], 2) for k in ("msm_sort", "msm_accumulate_kernel", "msm_reduce", "msm_digits_kernel")}, flush=True)
