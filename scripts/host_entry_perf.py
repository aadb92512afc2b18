"""Diagnostic: PCIe-inclusive rate of the host-buffer entry point (ftmpc_solve_batch) next to the device-pointer one."""
import sys, time
sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0, '/root/repo')
import numpy as np
import ft_mpc_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N, NT = 20, 8
mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT)
x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, 2, 1003)
xr = xref.reshape(-1, order='F')
mpc.reserve(B)
for rep in range(4):
    t0 = time.perf_counter()
    out = mpc.solve(x0, ub, stuck, xr)
    dt = time.perf_counter() - t0
    print(f"host entry: B={B} {dt*1e3:.2f} ms -> {B/dt:.0f} QP/s (u0 only)")
for rep in range(2):
    t0 = time.perf_counter()
    out = mpc.solve(x0, ub, stuck, xr, return_U=True)
    dt = time.perf_counter() - t0
    print(f"host entry + full U back: {dt*1e3:.2f} ms -> {B/dt:.0f} QP/s")
