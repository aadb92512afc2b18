"""Diagnostic: latency of ONE MPC step (the reference's own use: examples/sim.py solves one QP per 0.1 s tick)."""
import sys, time
sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0, '/root/repo')
import numpy as np
import ft_mpc_amd
for (N, NT, nf, dt) in ((15, 16, 2, "f32"), (20, 8, 2, "f32"), (15, 16, 2, "f64")):
    mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype=dt)
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(1, N, NT, nf, 7)
    xr = xref.reshape(-1, order='F')
    ts = []
    for rep in range(30):
        t0 = time.perf_counter()
        out = mpc.solve(x0, ub, stuck, xr)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[5:]) * 1e3
    print(f"N={N} NT={NT} faults={nf} dtype={dt}: one step median {np.median(ts):.3f} ms (min {ts.min():.3f}), iters {int(out['iters'][0])}, status {int(out['status'][0])}")
    mpc.close()
# the reference's own two-stage step (generalized-force MPC with the hull rows, then allocation), one instance
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
for dt in ("f32", "f64"):
    N, NT = 15, 16
    mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype=dt, max_iters=40)
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(1, N, NT, 2, 7)
    xr = xref.reshape(-1, order='F')
    hull = hull_tables(mpc.D, ub, stuck)
    ts = []
    for rep in range(30):
        t0 = time.perf_counter()
        out = mpc.solve_wrench(x0, ub, stuck, xr, hull=hull)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[5:]) * 1e3
    print(f"two-stage step N={N} NT={NT} dtype={dt}: median {np.median(ts):.3f} ms (min {ts.min():.3f}), iters {int(out['iters'][0])}, status {int(out['status'][0])}, alloc {int(out['alloc_status'][0])}")
    mpc.close()
