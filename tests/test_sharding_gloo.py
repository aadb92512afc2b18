"""CPU: the multi-GPU path (one process per rank, contiguous batch shards, host gather, no
data-path collective) rehearsed with world_size 2 over gloo.  The per-rank solver is the C oracle
here (no GPU in this container); on the GPU box the same function wraps BatchedMPC.solve."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "fault-tolerant-mpc_amd"))
    import torch.distributed as dist
    from ft_mpc_amd.sharding import solve_sharded
    from oracle import c_oracle as co, qp_oracle as qo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = qo.QPConfig(N=5, NT=8)
    x0, ub, stuck, xref = qo.make_batch(11, 5, 8, 2, 31)      # 11 instances: ragged split 5 + 6
    fn = lambda a, b, c: co.solve_batch(cfg, a, b, c, xref, return_U=False)
    out = solve_sharded(fn, x0, ub, stuck, rank=rank, world=world, dist=dist)
    dist.barrier()
    if rank == 0:
        ref = co.solve_batch(cfg, x0, ub, stuck, xref, return_U=False)
        q.put((np.abs(out["u0"] - ref["u0"]).max(), out["u0"].shape, int(out["iters"].sum() - ref["iters"].sum())))
    else:
        assert out is None
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    err, shape, diters = res
    assert err == 0.0 and shape == (11, 8) and diters == 0


def test_shard_bounds_cover_batch_exactly():
    from ft_mpc_amd.sharding import shard_bounds
    for B in (0, 1, 7, 65536, 262144):
        for w in (1, 2, 3, 8):
            edges = [shard_bounds(B, w, r) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
