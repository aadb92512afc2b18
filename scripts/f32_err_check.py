"""Diagnostic: fp32 headline kernel against the (polished, exact) C oracle on a batch: error distribution, worst instances.  args: B nfault seed"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd')
import numpy as np
import ft_mpc_amd
from oracle import qp_oracle as qo, c_oracle as co, refmath as rm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 4102
N, NT = 20, 8
mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT)
x0, ub, stuck, xref = qo.make_batch(B, N, NT, nf, seed)
out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
ref = co.solve_batch(qo.QPConfig(N=N, NT=NT), x0, ub, stuck, xref, nthreads=16, max_iters=60)
eU = np.abs(out["U"] - ref["U"]).reshape(B, -1).max(axis=1) / rm.F_MAX
e0 = np.abs(out["u0"] - ref["u0"]).max(axis=1) / rm.F_MAX
print("status", np.bincount(out["status"]), "iters %.2f max %d" % (out["iters"].mean(), out["iters"].max()))
print("U err: max %.2e  99.9%% %.2e  median %.2e ;  u0 err: max %.2e" % (eU.max(), np.quantile(eU, 0.999), np.median(eU), e0.max()))
print("count U err > 1e-5: %d, > 3e-5: %d, > 1e-4: %d" % ((eU > 1e-5).sum(), (eU > 3e-5).sum(), (eU > 1e-4).sum()))
for b in np.argsort(-eU)[:5]:
    d = np.abs(out["U"][b] - ref["U"][b]) / rm.F_MAX
    k, a = np.unravel_index(d.argmax(), d.shape)
    print("inst %d iters %d (ref %d) err %.2e at stage %d thruster %d: gpu %.6f ref %.6f ub %.2f" % (b, out["iters"][b], ref["iters"][b], eU[b], k, a, out["U"][b, k, a], ref["U"][b, k, a], ub[b, a]))
