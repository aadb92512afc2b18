#!/usr/bin/env python3
"""Bitwise repeatability of a solve: the same batch three times on one handle, then on a second handle.
Usage: python scripts/determinism.py [B] [N] [NT] [faults] [dtype]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "fault-tolerant-mpc_amd"))
import ft_mpc_amd  # noqa: E402

B, N, NT, nf = (int(a) for a in (sys.argv[1:5] + ["4096", "15", "16", "2"][len(sys.argv) - 1:]))
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, nf, 6300)
xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
cfg = ft_mpc_amd.MPCConfig(N=N, NT=NT, max_iters=40, dtype=dtype)
runs = []
for h in range(2):
    m = ft_mpc_amd.BatchedMPC(cfg)
    for r in range(3 if h == 0 else 1):
        runs.append(m.solve(x0, ub, stuck, xr, return_U=True))
    m.close()
ref = runs[0]
for i, o in enumerate(runs[1:], 1):
    d = np.abs(o["U"] - ref["U"]).max(axis=1)
    bad = np.flatnonzero(d > 0)
    print(f"run {i}: instances differing {bad.size} of {B}; max |dU| {d.max():.3e}; iters differing {(o['iters'] != ref['iters']).sum()}; first {bad[:8]}")
print("status", np.bincount(ref["status"]), "iters mean", ref["iters"].mean())
