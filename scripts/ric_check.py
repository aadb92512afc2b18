"""Diagnostic: the float64 kernels (Riccati = auto, wrench-space = workgroup) and the C oracle on one config-5 batch: statuses, iterations."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd')
import numpy as np
import ft_mpc_amd
from oracle import c_oracle as co, qp_oracle as qo
N, NT, B = 40, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 1005)
res = {}
for sel in ("auto", "workgroup"):
    m = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype="f64", max_iters=60, kernel_select=sel)
    res[sel] = m.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    print(sel, "status", np.bincount(res[sel]["status"], minlength=3), "iters mean %.2f max %d" % (res[sel]["iters"].mean(), res[sel]["iters"].max()))
    m.close()
d = np.abs(res["auto"]["u0"] - res["workgroup"]["u0"]).max(axis=1) / 3.4
print("u0 auto vs workgroup: max %.2e" % d.max(), "iters differ on", int((res["auto"]["iters"] != res["workgroup"]["iters"]).sum()))
slow = np.argsort(-res["auto"]["iters"])[:6]
ref = co.solve_batch(qo.QPConfig(N=N, NT=NT), x0[slow], ub[slow], stuck[slow], xref, nthreads=4, max_iters=100)
for i, b in enumerate(slow):
    print("inst %5d iters auto %d workgroup %d oracle %d | u0 err auto %.2e workgroup %.2e" % (b, res["auto"]["iters"][b], res["workgroup"]["iters"][b], ref["iters"][i],
          np.abs(res["auto"]["u0"][b] - ref["u0"][i]).max() / 3.4, np.abs(res["workgroup"]["u0"][b] - ref["u0"][i]).max() / 3.4))
