import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (checker only)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_workers():
    """One PlonkWorker per curve on cuda:0, shared by the GPU tests."""
    from distributed_plonk_amd.worker import PlonkWorker
    ws = {}

    def get(curve: str):
        if curve not in ws:
            ws[curve] = PlonkWorker(me=0, device=0, curve=curve)
        return ws[curve]

    yield get
    for w in ws.values():
        w.close()


def free_port() -> int:
    """A free TCP port on 127.0.0.1 for a torch.distributed rendezvous."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p
