import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "fault-tolerant-mpc_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Tests that hand torch tensors to the C-ABI need ONE HIP runtime in the process: let torch bring
    # its runtime up first (as bench.py does), the library then binds to the runtime already loaded.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def gpu_mpc_factory():
    """Creates BatchedMPC handles (HIP library) and closes them at session end."""
    import ft_mpc_amd
    made = []

    def make(**kw):
        m = ft_mpc_amd.BatchedMPC(**kw)
        made.append(m)
        return m

    yield make
    for m in made:
        m.close()
