"""Diagnostic: kernel 11 with the terminal set against the float64 kernel (hull rows + terminal set), states around the set."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import ft_mpc_amd
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
from ft_mpc_amd.controllers.tools.terminal_ingredients import load_terminal
from oracle import qp_oracle as qo, refmath as rm
N, NT = 15, 16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
term = load_terminal().term_set
At, bt = term.A, term.b.reshape(-1)
x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 9100)
rng = np.random.default_rng(9101); r = rm.spiral_r()
for b in range(B):      # tracking error on `scale` times the boundary of the set (tests/test_gpu_wrench.py:_near_terminal_set)
    e = rng.standard_normal(9); e *= scale / max((At @ e / bt).max(), 1e-9)
    R = rm.rot(x0[b, 6:10]); w = rm.OMEGA_DES + e[6:9]
    x0[b, 0:3] = e[0:3] - R.T @ r; x0[b, 3:6] = e[3:6] - R.T @ np.cross(w, r); x0[b, 10:13] = w
hull = hull_tables(qo.QPConfig(N=N, NT=NT).D, ub, stuck)
xr = xref.reshape(-1, order="F")
res = {}
for dt in ("f64", "f32"):
    m = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype=dt, max_iters=60, terminal_set=term)
    out = m.solve_wrench(x0, ub, stuck, xr, hull=hull, return_G=True)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); m.solve_wrench(x0, ub, stuck, xr, hull=hull); best = min(best, time.perf_counter() - t0)
    res[dt] = out
    print(dt, "status", np.bincount(out["status"], minlength=4), "iters mean %.2f max %d" % (out["iters"].mean(), out["iters"].max()),
          "alloc", np.bincount(out["alloc_status"], minlength=3), "%.2f ms -> %.0f QP/s (host buffers)" % (best * 1e3, B / best))
    m.close()
has = res["f64"]["status"] != 3
ok64, ok32 = (res["f64"]["status"] == 0) & has, (res["f32"]["status"] == 0) & has
print("reachable per float64 kernel %d, per kernel 11 %d, verdicts agree on %.2f %%" % (ok64.sum(), ok32.sum(), 100 * (ok64 == ok32)[has].mean()))
both = ok64 & ok32
e = np.abs(res["f32"]["G"][both] - res["f64"]["G"][both]).max(axis=(1, 2)) / 3.4
print("G err / f_max on %d instances: max %.2e p99.9 %.2e p99 %.2e median %.2e" % (both.sum(), e.max(), np.percentile(e, 99.9), np.percentile(e, 99), np.median(e)))
print("iters (f32 - f64) on those: mean %.2f max %d" % ((res["f32"]["iters"][both].astype(float) - res["f64"]["iters"][both]).mean(), (res["f32"]["iters"][both].astype(int) - res["f64"]["iters"][both]).max()))
idx = np.flatnonzero(both)[np.argsort(-e)[:10]]
print("instances above 1e-4:", int((e > 1e-4).sum()))
for b in idx:
    sl = bt - At @ np.zeros(9)      # (placeholder: slack of the set itself)
    print("inst %5d err %.2e  iters f32 %2d f64 %2d  faults %s" % (b, np.abs(res["f32"]["G"][b] - res["f64"]["G"][b]).max() / 3.4, res["f32"]["iters"][b], res["f64"]["iters"][b], np.flatnonzero(ub[b] == 0)))
# the three worst against the NumPy oracle (ipm_general with the same rows): which kernel is off?
cfg = qo.QPConfig(N=N, NT=NT)
for b in idx[:3]:
    with np.errstate(all="ignore"):
        _, T, st_, nit, qp = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref, term_set=(At, bt), iters=60)
    print("inst %5d oracle status %d iters %d | kernel 11 vs oracle %.2e, float64 kernel vs oracle %.2e (f_max)" %
          (b, st_, nit, np.abs(res["f32"]["G"][b] - T).max() / 3.4, np.abs(res["f64"]["G"][b] - T).max() / 3.4))
