"""Diagnostic: kernel 11 (fp32 hull-row kernel) against the float64 kernel on the same batch: error distribution, iterations, rates."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd')
import numpy as np
import ft_mpc_amd
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
from oracle import qp_oracle as qo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 15
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
nf = int(sys.argv[4]) if len(sys.argv) > 4 else 2
MU32 = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0      # mu_stop of the fp32 handle (0: library default)
SEED = int(sys.argv[6]) if len(sys.argv) > 6 else 7900
x0, ub, stuck, xref = qo.make_batch(B, N, NT, nf, SEED)
cfg = qo.QPConfig(N=N, NT=NT)
t0 = time.time(); hull = hull_tables(cfg.D, ub, stuck); print("hull tables %.2f s, %d sets" % (time.time() - t0, hull["A"].shape[0]))
xr = xref.reshape(-1, order="F")
res = {}
for dt in ("f64", "f32"):
    m = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype=dt, max_iters=40, mu_stop=(MU32 if dt == "f32" else 0.0))
    out = m.solve_wrench(x0, ub, stuck, xr, hull=hull, return_G=True)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); m.solve_wrench(x0, ub, stuck, xr, hull=hull); best = min(best, time.perf_counter() - t0)
    res[dt] = out
    # the C entry alone (arrays prepared, outputs preallocated): what a C caller sees
    import ctypes as C
    from ft_mpc_amd.batch import _ptr
    sel = np.flatnonzero(~np.asarray(hull["degenerate"], bool)); b = sel.size
    tk = lambda a: np.ascontiguousarray(a[sel])
    ax0, aub, ast_, ahs, ahb = tk(x0), tk(ub), tk(stuck), np.ascontiguousarray(hull["set"][sel], dtype=np.int32), tk(hull["b"])
    A = np.ascontiguousarray(hull["A"], dtype=np.float64)
    o_u0, o_t0 = np.empty((b, NT)), np.empty((b, 6)); o_st, o_it, o_as = (np.empty(b, np.int32) for _ in range(3))
    bestc = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc = m.lib.ftmpc_solve_wrench_batch(m._h, b, _ptr(ax0), _ptr(aub), _ptr(ast_), _ptr(A), A.shape[0], _ptr(ahs, C.c_int32), _ptr(ahb), int(hull["rows"]),
                                            _ptr(xr), 0, None, 0, None, _ptr(o_u0), _ptr(o_t0), None, _ptr(o_st, C.c_int32), _ptr(o_it, C.c_int32), _ptr(o_as, C.c_int32))
        bestc = min(bestc, time.perf_counter() - t0)
    assert rc == 0
    print("   C entry alone: %.2f ms -> %.0f QP/s (%d instances with a hull)" % (bestc * 1e3, b / bestc, b))
    print(dt, "status", np.bincount(out["status"], minlength=4), "iters mean %.2f max %d" % (out["iters"].mean(), out["iters"].max()),
          "alloc", np.bincount(out["alloc_status"], minlength=3), "%.2f ms -> %.0f QP/s (host buffers)" % (best * 1e3, B / best))
    m.close()
ok = (res["f64"]["status"] == 0)
e = np.abs(res["f32"]["G"][ok] - res["f64"]["G"][ok]).max(axis=(1, 2)) / 3.4
print("G err / f_max: max %.2e p99.9 %.2e p99 %.2e median %.2e" % (e.max(), np.percentile(e, 99.9), np.percentile(e, 99), np.median(e)))
e0 = np.abs(res["f32"]["u0"][ok] - res["f64"]["u0"][ok]).max(axis=1) / 3.4
print("u0 err / f_max: max %.2e p99.9 %.2e median %.2e" % (e0.max(), np.percentile(e0, 99.9), np.median(e0)))
print("f32 status on f64-converged:", np.bincount(res["f32"]["status"][ok], minlength=4))
print("iters diff (f32 - f64): min %d max %d mean %.2f" % ((res["f32"]["iters"][ok].astype(int) - res["f64"]["iters"][ok]).min(), (res["f32"]["iters"][ok].astype(int) - res["f64"]["iters"][ok]).max(), (res["f32"]["iters"][ok].astype(float) - res["f64"]["iters"][ok]).mean()))

bad = np.flatnonzero(ok)[np.argsort(-e)[:12]]
for b in bad:
    print("inst %5d err %.2e  f32 status %d iters %2d | f64 iters %2d  faults %s  hull set %d" % (b, np.abs(res["f32"]["G"][b] - res["f64"]["G"][b]).max() / 3.4, res["f32"]["status"][b], res["f32"]["iters"][b], res["f64"]["iters"][b], np.flatnonzero(ub[b] == 0), hull["set"][b]))
big = e > 1e-4
print("instances above 1e-4:", big.sum(), " by f32 status:", np.bincount(res["f32"]["status"][ok][big], minlength=3), " iters of those: mean %.1f" % res["f32"]["iters"][ok][big].mean())
