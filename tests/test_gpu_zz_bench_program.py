"""bench.py's N > 1 program on the one GPU a test box has: `--multi-path` runs the whole multi-rank code path — communicators, the
in-library RCCL collectives, two lanes of distributed transforms, sharded batched commitments, the other scheme, the polynomial-level-
parallel leg, the verification leg — on a world of ONE rank.  Every leg that checks itself must say `true`.  (Named to sort last: a
failure here must not keep `pytest -x` from the kernel parity tests.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra):
    from conftest import free_port
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--log-n", "12", "--no-cpu-baseline", "--no-next-rows", "--no-other-configs", *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    return json.loads(lines[0])


def test_multi_rank_program_on_a_world_of_one():
    d = _run(["--multi-path"])
    assert d["n_gpus"] == 1 and d["verified"] is True and all(d["verification"].values()), d.get("verification")
    assert d.get("aborted_optional_leg") is None
    assert "error" not in (d["other_scheme"] or {}), d["other_scheme"]
    pp = d["polynomial_parallel"]
    assert pp["verified"] is True and pp["ranks"] == 1 and pp["data_path_collectives_per_step"] == 0, pp
    assert pp["operations_per_rank"] == [{"commit": 13, "coset_ifft_8n": 1, "coset_fft_8n": 25, "intt_n": 7}], pp


def test_busiest_rank_of_eight_simulated():
    """`--simulate-ranks 8`: rank 0's share of the reference's 2-D scheme with no-op exchanges, and the busiest rank's share of the
    polynomial-level-parallel scheme (2 commitments + 3 coset FFTs), which has nothing to simulate away and verifies itself."""
    d = _run(["--simulate-ranks", "8"])
    pp = d["polynomial_parallel"]
    assert pp["verified"] is True and pp["ranks"] == 8 and pp["result_collectives_per_step"] == 0, pp
    assert sum(sum(r.values()) for r in pp["operations_per_rank"]) == 46
