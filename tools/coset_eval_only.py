"""Time the zero-padding-aware coset FFT (plonk_coset_eval_dev: n + 3 coefficients -> 8n evaluations) on its own.
    python tools/coset_eval_only.py [log_n ...]          (env PLONK_NTT_LOGT8 / PLONK_NTT_LOGT9 select the tile width)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_plonk_amd import fr as _fr
from distributed_plonk_amd.worker import PlonkWorker

curve = os.environ.get("CURVE", "bn254")
w = PlonkWorker(0, 0, curve)
g = _fr.FIELDS[curve].to_limbs(_fr.FIELDS[curve].generator)
for log_n in [int(x) for x in sys.argv[1:]] or [24]:
    n, m = 1 << log_n, 8 << log_n
    p, out = w.alloc((n + 3) * 32), w.alloc(m * 32)
    w.synth_fr(1, p.ptr, n + 3)
    w.profile_enable(True)
    for it in range(3):
        w.profile_reset()
        for _ in range(4):
            w.coset_eval_dev(p.ptr, n + 3, m, g, out.ptr)
        w.sync()
    res = {k: w.profile_get(k) for k in ["ntt_pass_kernel"] + [f"ntt_pass_kernel<{i}>" for i in range(5, 10)]}
    tot = res["ntt_pass_kernel"][0] / 4
    print(f"coset_eval n=2^{log_n} -> 8n: {tot:.3f} ms per transform, alg {64 * m / tot / 1e6:.1f} GB/s (2*8n*32 B)",
          {k: round(v[0] / max(v[1], 1), 4) for k, v in res.items() if v[1]}, "LOGT8=" + os.environ.get("PLONK_NTT_LOGT8", "-"), flush=True)
    p.free(); out.free()
w.close()
