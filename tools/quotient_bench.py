"""Time the quotient-evaluation kernel on HBM-resident inputs (next row)."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from distributed_plonk_amd.worker import PlonkWorker
w = PlonkWorker(0, 0, "bn254")
for log_n in [int(a) for a in sys.argv[1:]]:
    n, m = 1 << log_n, 8 << log_n
    w.init(None, n, m)
    bufs = [w.alloc(m * 32) for _ in range(25)]
    for j, b in enumerate(bufs):
        w.synth_fr(100 + j, b.ptr, m)
    out = w.alloc(m * 32)
    ch = np.arange(32, dtype=np.uint64).reshape(8, 4) + 5
    ptr = [b.ptr for b in bufs]
    if __import__('os').environ.get('QUOT_ALIAS'):      # diagnostic: all 25 input vectors are ONE buffer (1 stream instead of 26: compute + latency only)
        ptr = [bufs[0].ptr] * 25
    w.profile_enable(True)
    alg = 27 * 32 * m          # 26 input reads (z twice) + 1 write of 32 B per point
    names = {0: "unlifted, 4 waves (default)", 4: "unlifted, uncapped registers (3 waves)", 1: "lifted wires, 1 product / reduction",
             2: "lifted, 2 products / reduction", 3: "lifted, 3 products / reduction", 5: "unlifted, 8 points per lane in a rolled loop",
             6: "compact: rolled hash / permutation loops, operand loads one product ahead", 7: "compact, capped at 4 waves (128 VGPRs)"}
    for variant in (0, 6, 7, 5, 4, 1, 2, 3):
        w.set_option("quotient_fuse", variant)
        for it in range(3):
            w.profile_reset()
            w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr)
            w.sync()
        ms, cnt = w.profile_get("quotient_evals_kernel")
        print(f"2^{log_n} quotient_fuse={variant} ({names[variant]}): {ms:.3f} ms, algorithmic {alg / ms / 1e6:.1f} GB/s = {alg / ms / 1e6 / 8000:.4f} of 8 TB/s", flush=True)
    w.set_option("quotient_fuse", 6)                  # back to the shipped default
    for b in bufs + [out]:
        b.free()
