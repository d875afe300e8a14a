"""bench.py at N = 1 without importing torch (measurement tool for very short GPU leases: `import torch` costs a fresh box a minute).

    python tools/bench_notorch.py --log-n 20 --steps 5 --warmup 1 --no-cpu-baseline --no-next-rows --no-other-configs [--overlap-phases on|off]

At N = 1 bench.py needs torch only for is_available / set_device / synchronize and for `torch.distributed.is_initialized()` at tear-down; the
stand-ins below do those through the library's own contexts (Bench.full_sync synchronises every context before it calls torch).  The line is
bench.py's own; it gains `"torch": "stub"`.  Never for N > 1 (the launcher's rendezvous and the timing barrier ARE torch.distributed)."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

torch = types.ModuleType("torch")
torch.cuda = types.SimpleNamespace(is_available=lambda: True, set_device=lambda d: None, synchronize=lambda: None, device_count=lambda: 1)
torch.device = lambda kind, index=0: (kind, index)
dist = types.ModuleType("torch.distributed")
dist.is_initialized = lambda: False
torch.distributed = dist
sys.modules["torch"], sys.modules["torch.distributed"] = torch, dist

if int(os.environ.get("WORLD_SIZE", "1")) != 1 or "--gpus" in sys.argv and sys.argv[sys.argv.index("--gpus") + 1] != "1":
    raise SystemExit("tools/bench_notorch.py is for N = 1 only")

import bench                                       # noqa: E402
import benchlib.headline as headline               # noqa: E402

_result_line = headline.result_line


def result_line(b, dt, phases_ms):
    out = _result_line(b, dt, phases_ms)
    if out is not None:
        out["torch"] = "stub"
    return out


headline.result_line = result_line
bench.main()
