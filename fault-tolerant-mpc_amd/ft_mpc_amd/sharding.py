"""Batch sharding across the GPUs of one node: instances are independent, so
the batch axis is split contiguously, one process per GPU, NO data-path collective; results are
gathered on the host of rank 0 (torch.distributed is used for that host gather only --
gloo on CPU tensors, so it works with or without RCCL)."""
from __future__ import annotations

import numpy as np


def shard_bounds(B: int, world: int, rank: int):
    """Contiguous split: rank r owns [lo, hi).  Sizes differ by at most one."""
    lo = (B * rank) // world
    hi = (B * (rank + 1)) // world
    return lo, hi


def solve_sharded(solve_fn, x0, ub, stuck, *, rank=0, world=1, dist=None, **kw):
    """Every rank passes the FULL batch arrays (or at least its own slice filled in);
    `solve_fn(x0, ub, stuck, **kw) -> dict(u0, status, iters)` is called on this rank's slice
    (per-instance entries of kw -- `warmU`, per-instance `xref`/`uref` -- are sliced alike).
    Returns the gathered dict on rank 0 and None elsewhere."""
    B = np.asarray(x0).reshape(-1, 13).shape[0]
    lo, hi = shard_bounds(B, world, rank)
    loc_kw = {}
    for k, v in kw.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B and k in ("warmU", "xref_batch", "uref_batch"):
            loc_kw[k] = v[lo:hi]
        else:
            loc_kw[k] = v
    out = solve_fn(np.asarray(x0).reshape(-1, 13)[lo:hi], np.asarray(ub)[lo:hi], np.asarray(stuck)[lo:hi], **loc_kw)
    if world == 1 or dist is None:
        return out
    keys = [k for k in ("u0", "U", "status", "iters") if out.get(k) is not None]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object({k: out[k] for k in keys}, gathered, dst=0)
    if rank != 0:
        return None
    return {k: np.concatenate([g[k] for g in gathered], axis=0) for k in keys}
