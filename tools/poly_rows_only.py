"""The O(n) rows alone (for rocprofv3 --kernel-trace --stats): python tools/poly_rows_only.py 24 [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributed_plonk_amd.worker import PlonkWorker
log_n = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << log_n
w = PlonkWorker(0, 0, "bn254")
src, dst = w.alloc(n * 32), w.alloc(n * 32)
w.synth_fr(7, src.ptr, n)
z = np.array([0x1234567, 0x89abcdef, 0x13579bdf, 0x02468ace], dtype=np.uint64)
w.profile_enable(True)
if os.environ.get("ROWS_PERM", "1") == "1":
    from distributed_plonk_amd.synthetic import SyntheticInstance
    inst = SyntheticInstance(w, log_n, seed=0xC1AC, num_inputs=3, tau=12345, init_worker=False)
    be = np.array([5, 6, 7, 8], dtype=np.uint64)
    ga = np.array([9, 10, 11, 12], dtype=np.uint64)
    names = ["perm_terms_kernel", "perm_scan_num", "perm_scan_den_final"]
    for it in range(reps + 1):
        w.profile_reset()
        w.perm_product_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, be, ga, n, dst.ptr)
        w.sync()
        if it:
            print(f"2^{log_n} perm_product: " + "  ".join(f"{k} {w.profile_get(k)[0]:.3f}" for k in names) + " ms", flush=True)
    inst.close()
for name, fn in (("poly_div_kernels", lambda: w.poly_div_linear_dev(src.ptr, n, z, dst.ptr)), ("poly_eval_kernel", lambda: w.poly_eval_dev(src.ptr, n, z))):
    t = []
    for it in range(reps + 1):
        w.profile_reset()
        fn()
        w.sync()
        t.append(w.profile_get(name)[0])
    print(f"2^{log_n} {name}: " + " ".join(f"{x:.3f}" for x in t[1:]) + " ms", flush=True)
