"""Throughput and accuracy of the two-stage step (generalized-force MPC with hull rows [+ terminal set], then allocation) on an fp32
handle (kernel 11 + hand-over) and a float64 handle, regular random batch (scripts/hull32_tset_check.py: the boundary batch)."""
import sys, time
sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0, '/root/repo')
import numpy as np
import ft_mpc_amd
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
from oracle import alloc_oracle as ao, qp_oracle as qo

N, NT = 15, 16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mu32 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0      # mu_stop of the fp32 handle (0: the library's 1e-10)
x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, nf, 1011)
xr = np.ascontiguousarray(xref.reshape(-1, order='F'))
res = {}
for dt in ("f64", "f32"):
    m = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype=dt, max_iters=40, mu_stop=mu32 if dt == "f32" else 0.0)
    if dt == "f64":
        hull = hull_tables(m.D, ub, stuck)
    out = m.solve_wrench(x0, ub, stuck, xr, hull=hull, return_G=True)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); m.solve_wrench(x0, ub, stuck, xr, hull=hull); best = min(best, time.perf_counter() - t0)
    ok = out["status"] == 0
    print(f"{dt}: {best*1e3:7.2f} ms  {B/best:9.0f} QP/s (host buffers)  iters {out['iters'][ok].mean():.2f} solved {int(ok.sum())} flat hull {int((out['status']==3).sum())} "
          f"alloc status {np.bincount(out['alloc_status'], minlength=3)} handed over {m.last_handed_over()}", flush=True)
    res[dt] = out
    m.close()
both = (res["f64"]["status"] == 0) & (res["f32"]["status"] == 0)
eG = np.abs(res["f32"]["G"][both] - res["f64"]["G"][both]).max(axis=(1, 2)) / 3.4
eu = np.abs(res["f32"]["u0"][both] - res["f64"]["u0"][both]).max(axis=1) / 3.4
print("fp32 vs float64 handle: G max %.2e p99.9 %.2e median %.2e | u0 after allocation max %.2e p99.9 %.2e median %.2e (f_max)" %
      (eG.max(), np.percentile(eG, 99.9), np.median(eG), eu.max(), np.percentile(eu, 99.9), np.median(eu)))
# the allocation itself: the worst u0 instances against the oracle allocator on the wrench each kernel handed over
cfg = qo.QPConfig(N=N, NT=NT)
for b in np.flatnonzero(both)[np.argsort(-eu)[:4]]:
    for dt in ("f64", "f32"):
        want = res[dt]["tau0"][b] - cfg.D @ stuck[b]
        ur = ao.allocate(cfg.D, want, ub[b])[0]
        print("  inst %5d %s: u0 vs oracle allocation of ITS tau0 %.2e f_max, |u|^2 %.6f vs %.6f" % (b, dt, np.abs(res[dt]["u0"][b] - ur).max() / 3.4, (res[dt]["u0"][b] ** 2).sum(), (ur ** 2).sum()))
