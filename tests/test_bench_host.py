"""CPU: host-side pieces of bench.py (no GPU): the flop model bookkeeping, the usable-core count the CPU
baseline is sized from, the source hash the PMC traffic figure is keyed on, and that asking for GPUs on a box
without one fails loudly instead of exiting quietly."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_batch_flops_equals_the_per_instance_sum():
    rng = np.random.default_rng(0)
    ub = (rng.random((257, 8)) > 0.2) * 3.4
    it = rng.integers(6, 20, 257)
    ref = sum(bench.algorithmic_flops(20, int(a), int(k)) for a, k in zip((ub > 0).sum(1), it))
    assert bench.batch_flops(20, ub, it) == pytest.approx(ref, rel=1e-12)
    # N=20, NT=8 nominal, K=10 -> 17.0 MFLOP
    assert bench.algorithmic_flops(20, 8, 10) == pytest.approx(17.0e6, rel=0.01)


def test_usable_cores_and_source_hash():
    info = bench.host_cpu_info()
    assert 1 <= info["usable"] <= info["sched_affinity"] <= (info["os_cpu_count"] or 10 ** 6)
    h = bench.csrc_hash()
    assert len(h) == 16 and h == bench.csrc_hash()


def test_gpus_flag_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "no HIP device" in r.stderr and "torch.distributed.run" not in r.stderr
