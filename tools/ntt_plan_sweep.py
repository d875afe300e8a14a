import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from distributed_plonk_amd.worker import PlonkWorker
w = PlonkWorker(0, 0, "bn254")
for log_n in (20, 24, 27):
    n = 1 << log_n
    a = w.alloc(n * 32); b = w.alloc(n * 32)
    w.synth_fr(1, a.ptr, n)
    w.profile_enable(True)
    for mx in (9, 8, 7, 6):
        w.set_option("ntt_max_log_r", mx)
        for it in range(3):
            w.profile_reset()
            w.ntt_dev(a.ptr, b.ptr, n, False, True)
            w.ntt_dev(b.ptr, a.ptr, n, False, True)
            w.sync()
        ms, cnt = w.profile_get("ntt_pass_kernel")
        print(log_n, "max_log_r", mx, "passes", cnt // 2, "NTT ms", round(ms / 2, 3), "alg GB/s", round(64 * n / (ms / 2) / 1e6, 1), flush=True)
    a.free(); b.free()
