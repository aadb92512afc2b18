"""Diagnostic: the two-stage step on an fp32 handle; dumps the wrenches whose allocation does not end with status 0."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd')
import numpy as np
import ft_mpc_amd
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
from oracle import qp_oracle as qo
rows = []
for (N, NT, B, nf, seed) in ((10, 16, 16384, 3, 7900), (15, 16, 16384, 2, 7900), (15, 16, 16384, 2, 1011)):
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nf, seed) if seed != 1011 else ft_mpc_amd.make_synthetic_batch(B, N, NT, nf, seed)
    cfg = qo.QPConfig(N=N, NT=NT)
    hull = hull_tables(cfg.D, ub, stuck)
    m = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype="f32", max_iters=40)
    out = m.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), hull=hull, return_G=True)
    bad = np.flatnonzero((out["alloc_status"] != 0) & (out["status"] == 0))
    print(N, nf, seed, "allocation not solved:", bad, out["alloc_status"][bad])
    for b in bad:
        A = hull["A"][hull["set"][b]]
        sl = hull["b"][b] - A @ out["tau0"][b]
        print("  inst", b, "hull slack of tau0: min %.3e, rows within 1e-6: %d" % (sl.min(), (sl < 1e-6).sum()), "iters", out["iters"][b])
        rows.append(np.r_[out["tau0"][b], ub[b], stuck[b], out["u0"][b], out["alloc_status"][b]])
    m.close()
np.save("/root/repo/gpurun_out/r04_alloc_fail.npy", np.array(rows))
