"""Diagnostic: the two-stage form with the terminal set (kernel 11 + hand-over on an fp32 handle, the float64 kernel on a float64
handle) against the NumPy oracle with its active-set polish, vehicles with the tracking error on the boundary of the set."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import ft_mpc_amd
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
from ft_mpc_amd.controllers.tools.terminal_ingredients import load_terminal
from oracle import qp_oracle as qo, refmath as rm, batch as ob

if __name__ == "__main__":
    N, NT = 15, 16
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    use_t = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
    term = load_terminal().term_set
    At, bt = term.A, term.b.reshape(-1)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 9100)
    rng = np.random.default_rng(9101); r = rm.spiral_r()
    if use_t:
        for b in range(B):      # tracking error on `scale` times the boundary of the set (tests/test_gpu_wrench.py:_near_terminal_set)
            e = rng.standard_normal(9); e *= scale / max((At @ e / bt).max(), 1e-9)
            R = rm.rot(x0[b, 6:10]); w = rm.OMEGA_DES + e[6:9]
            x0[b, 0:3] = e[0:3] - R.T @ r; x0[b, 3:6] = e[3:6] - R.T @ np.cross(w, r); x0[b, 10:13] = w
    t0 = time.perf_counter()
    ref = ob.solve_wrench_batch(N, NT, x0, ub, stuck, xref, term_set=(At, bt) if use_t else None, iters=60)
    print("oracle: %.1f s, status" % (time.perf_counter() - t0), np.bincount(ref["status"], minlength=4), "iters mean %.2f" % ref["iters"][ref["status"] == 0].mean(),
          "hull & terminal rows active together: %d" % (ref["active"][ref["status"] == 0] > 0).all(axis=1).sum(), flush=True)
    hull = hull_tables(qo.QPConfig(N=N, NT=NT).D, ub, stuck)
    xr = xref.reshape(-1, order="F")
    ok_ref = ref["status"] == 0
    for dt in ("f64", "f32"):
        m = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype=dt, max_iters=60, terminal_set=term if use_t else None)
        out = m.solve_wrench(x0, ub, stuck, xr, hull=hull, return_G=True)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); m.solve_wrench(x0, ub, stuck, xr, hull=hull); best = min(best, time.perf_counter() - t0)
        ok = out["status"] == 0
        both = ok & ok_ref
        e = np.abs(out["G"][both] - ref["G"][both]).max(axis=(1, 2)) / 3.4
        e0 = np.abs(out["G"][both][:, 0] - ref["G"][both][:, 0]).max(axis=1) / 3.4
        print("%s status" % dt, np.bincount(out["status"], minlength=4), "iters mean %.2f" % out["iters"][ok].mean(), "alloc", np.bincount(out["alloc_status"], minlength=3),
              "%.2f ms -> %.0f QP/s (host buffers)" % (best * 1e3, B / best))
        print("   verdicts differ on %d; G err / f_max on %d instances: max %.2e p99.9 %.2e p99 %.2e median %.2e above 1e-4: %d above 1e-5: %d | tau0 max %.2e" %
              ((ok != ok_ref).sum(), both.sum(), e.max(), np.percentile(e, 99.9), np.percentile(e, 99), np.median(e), (e > 1e-4).sum(), (e > 1e-5).sum(), e0.max()))
        idx = np.flatnonzero(both)[np.argsort(-e)[:6]]
        for b in idx:
            print("   inst %5d err %.2e iters %2d (oracle %2d) active hull %d term %d" % (b, np.abs(out["G"][b] - ref["G"][b]).max() / 3.4, out["iters"][b], ref["iters"][b], ref["active"][b, 0], ref["active"][b, 1]))
        m.close()
