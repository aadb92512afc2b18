"""GPU: bench.py honours the driver contract (one JSON line with the metric, roofline and cpu_baseline
objects) and its multi-rank path works (2 ranks rehearsed on one GPU, gloo timing bracket)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8192"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert j["roofline"]["bound"] == "valu" and 0 < j["roofline"]["frac"] < 1
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["gpu_vs_port_max_u0_err_over_fmax"] < 1e-4
    assert j["config"]["not_converged"] == 0 and j["value"] > 1e5
    # the warm-started rate is reported beside the cold one (a different linearisation point, hence a different QP)
    assert j["warm_start"]["value"] > 1e5 and j["warm_start"]["ipm_iters_mean"] > 1


def test_two_ranks_on_one_gpu():
    env = dict(os.environ, FTMPC_BENCH_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--batch", "8192"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "cpu_baseline" not in j
    assert j["value"] > 1e5


def test_gpus_flag_without_a_launcher_fans_out_in_process():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the process drives both device slots itself through the
    library's multi-GPU driver (both on device 0 here) and prints the one JSON line, strong-scaling keys included."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FTMPC_BENCH_SINGLE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8192"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["launcher"].startswith("in-process")
    assert j["value"] > 1e5 and j["config"]["not_converged"] == 0 and 0 < j["roofline"]["frac"] < 1
    assert set(j["strong_scaling"]) == {"65536", "262144"}
    for s in j["strong_scaling"].values():
        assert s["value"] > 1e5 and s["parallel_efficiency"] > 0


def test_eight_slots_without_a_launcher_print_a_complete_line():
    """`python bench.py --gpus 8` as the driver runs it on an 8-GPU node, rehearsed with all eight device slots on this
    box's one GPU (eight host threads, handles and stream sets at once): the complete line, strong-scaling keys included."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FTMPC_BENCH_SINGLE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "4096"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["config"]["launcher"].startswith("in-process")
    assert j["value"] > 1e5 and j["config"]["not_converged"] == 0 and 0 < j["roofline"]["frac"] < 1
    assert set(j["strong_scaling"]) == {"65536", "262144"}
    for s in j["strong_scaling"].values():
        assert s["value"] > 1e5 and s["parallel_efficiency"] > 0
