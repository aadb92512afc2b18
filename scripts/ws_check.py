#!/usr/bin/env python3
"""Kernel 8 (kernel_select auto) against the dense kernels (kernel_select dense) on the same batch: outputs, iterations, time.
Usage: python scripts/ws_check.py [B] [N] [NT] [faults]"""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "fault-tolerant-mpc_amd"))
import ft_mpc_amd
B, N, NT, nf = (int(a) for a in (sys.argv[1:5] + ["512", "15", "16", "2"][len(sys.argv) - 1:]))
x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, nf, 6400)
xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
res = {}
for ws in ("dense", "auto"):
    m = ft_mpc_amd.BatchedMPC(ft_mpc_amd.MPCConfig(N=N, NT=NT, max_iters=40, kernel_select=ws))
    out = m.solve(x0, ub, stuck, xr, return_U=True)
    t0 = time.perf_counter()
    for _ in range(3):
        out = m.solve(x0, ub, stuck, xr, return_U=True)
    dt = (time.perf_counter() - t0) / 3
    res[ws] = out
    print(f"kernel_select={ws}: {dt * 1e3:8.2f} ms per call ({B / dt:10.0f} QP/s host entry)  status {np.bincount(out['status'], minlength=3)}  iters mean {out['iters'].mean():.2f} max {out['iters'].max()}")
    m.close()
d = np.abs(res["auto"]["U"] - res["dense"]["U"]).reshape(B, -1).max(axis=1) / 3.4
print(f"max |U_ws - U_default| / f_max: {d.max():.3e}  median {np.median(d):.3e}  iters differing {(res['auto']['iters'] != res['dense']['iters']).sum()}")
