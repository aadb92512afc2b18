"""Throughput of the SURVEY section 8(f) "next" rows on one GPU (parity-test shapes, not bench lines):
generalized-force formulation (hull rows + allocation), terminal-set modes, the cost kernel and the line-search SQP."""
import sys, time
sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0, '/root/repo')
import numpy as np
import ft_mpc_amd
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
from ft_mpc_amd.controllers.tools.terminal_ingredients import load_terminal

def best(f, reps=3):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); b = min(b, time.perf_counter() - t0)
    return b, r

T = load_terminal()
for (N, NT, B) in ((20, 16, 16384), (15, 16, 16384)):
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, 2, 1011)
    xr = np.ascontiguousarray(xref.reshape(-1, order='F'))
    mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype="f64", max_iters=40)
    t0 = time.perf_counter(); hull = hull_tables(mpc.D, ub, stuck); th = time.perf_counter() - t0
    dt, out = best(lambda: mpc.solve_wrench(x0, ub, stuck, xr, hull=hull))
    ok = out["status"] == 0
    print(f"wrench form (hull rows + allocation)  N={N} NT={NT} B={B}: {dt*1e3:8.1f} ms  {B/dt:9.0f} QP/s (host buffers in/out)  iters {out['iters'][ok].mean():.2f}  "
          f"solved {int(ok.sum())}  no-hull {int((out['status']==3).sum())}  hull tables on the host {th*1e3:.0f} ms for {hull['A'].shape[0]} fault sets", flush=True)
    dt2, o2 = best(lambda: mpc.solve(x0, ub, stuck, xr))
    print(f"   thruster form, same handle (float64 kernel, n = {N*(NT-2)}): {dt2*1e3:8.1f} ms  {B/dt2:9.0f} QP/s", flush=True)
    mpc.close()
    if 6 * N <= 96:      # kernel 11: the same formulation on one wave per instance (fp32 handle)
        m32 = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype="f32", max_iters=40)
        dt3, o3 = best(lambda: m32.solve_wrench(x0, ub, stuck, xr, hull=hull))
        both = ok & (o3["status"] == 0)
        err = np.abs(o3["u0"][both] - out["u0"][both]).max() / 3.4
        print(f"   the same on an fp32 handle (kernel 11, one wave per instance): {dt3*1e3:8.1f} ms  {B/dt3:9.0f} QP/s  iters {o3['iters'][both].mean():.2f}  "
              f"solved {int((o3['status']==0).sum())}  u0 within {err:.1e} f_max of the float64 kernel", flush=True)
        m32.close()
N, NT, B = 20, 8, 16384
x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, 2, 1012)
x0[:, 0:3] *= 0.05; x0[:, 3:6] *= 0.1
xr = np.ascontiguousarray(xref.reshape(-1, order='F'))
mt = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype="f64", max_iters=40, terminal_set=T.term_set)
dt, out = best(lambda: mt.solve(x0, ub, stuck, xr))
print(f"thruster form + 72-row terminal set   N={N} NT={NT} B={B}: {dt*1e3:8.1f} ms  {B/dt:9.0f} QP/s  reachable {int((out['status']==0).sum())}  iters {out['iters'].mean():.2f}", flush=True)
mt.close()
mq = ft_mpc_amd.BatchedMPC(N=N, NT=NT, terminal_cost=T)
B2 = 65536
x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B2, N, NT, 2, 1003)
U = np.random.default_rng(0).uniform(0, 3.4, (B2, N, NT)) * (ub[:, None, :] > 0)
dt, J = best(lambda: mq.eval_cost(x0, ub, stuck, xr, U))
print(f"nonlinear cost kernel (full terminal cost) N={N} NT={NT} B={B2}: {dt*1e3:8.1f} ms  {B2/dt:9.0f} evaluations/s (host buffers)", flush=True)
t0 = time.perf_counter(); out = mq.solve_sqp(x0[:16384], ub[:16384], stuck[:16384], xr, sqp_iters=10); dt = time.perf_counter() - t0
print(f"line-search SQP, 10 major iterations max, B=16384: {dt*1e3:8.1f} ms  {16384/dt:9.0f} NLP solves/s  major iterations {out['sqp_iters'].mean():.2f}  "
      f"IPM iterations {out['iters'].mean():.1f}  cost {np.median(out['cost0']):.0f} -> {np.median(out['cost']):.1f} (median)", flush=True)
t0 = time.perf_counter(); outd = mq.solve_sqp_device(x0, ub, stuck, xr, sqp_iters=10); dtd = time.perf_counter() - t0
t0 = time.perf_counter(); outd = mq.solve_sqp_device(x0, ub, stuck, xr, sqp_iters=10); dtd = min(dtd, time.perf_counter() - t0)
print(f"line-search SQP ON THE DEVICE (ftmpc_solve_sqp_batch), 10 major iterations, B={B2}: {dtd*1e3:8.1f} ms  {B2/dtd:9.0f} NLP solves/s  "
      f"major iterations {outd['sqp_iters'].mean():.2f}  IPM iterations {outd['iters'].mean():.1f}  cost {np.median(outd['cost0']):.0f} -> {np.median(outd['cost']):.1f} (median)", flush=True)
mq.close()
