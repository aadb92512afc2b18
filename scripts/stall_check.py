import sys
sys.path.insert(0,'/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0,'/root/repo')
import numpy as np, ft_mpc_amd
from oracle import c_oracle as co, qp_oracle as qo
N,NT,B=20,8,65536
x0,ub,stuck,xref=ft_mpc_amd.make_synthetic_batch(B,N,NT,2,1003)
for mi in (18,30):
    mpc=ft_mpc_amd.BatchedMPC(N=N,NT=NT,max_iters=mi)
    out=mpc.solve(x0,ub,stuck,xref.reshape(-1,order='F'))
    bad=np.flatnonzero(out['status']!=0)
    print("max_iters",mi,"not converged:",bad, "iters hist",np.bincount(out['iters'])[8:])
    if bad.size:
        ref=co.solve_batch(qo.QPConfig(N=N,NT=NT),x0[bad],ub[bad],stuck[bad],xref,max_iters=60)
        print("  err of those vs exact:",np.abs(out['u0'][bad]-ref['u0']).max(axis=1)/3.4, "oracle iters",ref['iters'])
    mpc.close()
