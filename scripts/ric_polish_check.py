"""Diagnostic: kernel 12 (early polish) against the C oracle's iteration at mu 1e-13 AND the independent exact solver (BVLS) on the
instances where the two differ most -- which of them is nearer the exact solution?   args: N NT B seed"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd')
import numpy as np
import ft_mpc_amd
from oracle import qp_oracle as qo, c_oracle as co, refmath as rm
F_MAX = rm.F_MAX
N, NT, B, seed = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (40, 16, 8192, 1005)
cfg = qo.QPConfig(N=N, NT=NT)
mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype="f64", max_iters=40)
x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, seed)
out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
ref = co.solve_batch(cfg, x0, ub, stuck, xref, nthreads=16, max_iters=60)
ok = ref["status"] == 0
err = np.abs(out["U"] - ref["U"]).reshape(B, -1).max(axis=1) / F_MAX
err0 = np.abs(out["u0"] - ref["u0"]).max(axis=1) / F_MAX
print("status gpu", np.bincount(out["status"]), "ref", np.bincount(ref["status"]), " iters gpu %.2f ref %.2f" % (out["iters"].mean(), ref["iters"].mean()))
print("U apart > 1e-7: %d  > 1e-6: %d  > 1e-5: %d ;  u0 apart > 1e-7: %d" % ((err[ok] > 1e-7).sum(), (err[ok] > 1e-6).sum(), (err[ok] > 1e-5).sum(), (err0[ok] > 1e-7).sum()))
e = np.where(ok, err, 0)
for b in np.argsort(-e)[:8]:
    u0, U, _ = qo.solve_instance(cfg, x0[b], ub[b], stuck[b], xref, exact=True)
    print("inst %d  iters gpu %d ref %d  gpu-ref %.2e   gpu-exact %.2e   ref-exact %.2e" % (b, out["iters"][b], ref["iters"][b], err[b], np.abs(out["U"][b] - U).max() / F_MAX, np.abs(ref["U"][b] - U).max() / F_MAX))
