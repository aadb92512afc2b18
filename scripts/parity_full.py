"""Full-batch parity study: GPU u0 vs the exact (mu 1e-13) float64 oracle; error percentiles."""
import sys, time, os
sys.path.insert(0,'/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0,'/root/repo')
import numpy as np
import ft_mpc_amd
from oracle import c_oracle as co, qp_oracle as qo
B=int(sys.argv[1]) if len(sys.argv)>1 else 65536
nf=int(sys.argv[2]) if len(sys.argv)>2 else 2
seed=int(sys.argv[3]) if len(sys.argv)>3 else 1003
N=int(os.environ.get('PF_N','20')); NT=int(os.environ.get('PF_NT','8'))
kw={}
if len(sys.argv)>4: kw['mu_stop']=float(sys.argv[4])
if len(sys.argv)>5: kw['max_iters']=int(sys.argv[5])
mpc=ft_mpc_amd.BatchedMPC(N=N,NT=NT,**kw)
x0,ub,stuck,xref=ft_mpc_amd.make_synthetic_batch(B,N,NT,nf,seed)
out=mpc.solve(x0,ub,stuck,xref.reshape(-1,order='F'),return_U=True)
t=time.time()
ref=co.solve_batch(qo.QPConfig(N=N,NT=NT),x0,ub,stuck,xref,nthreads=min(32,len(os.sched_getaffinity(0))),max_iters=60,mu_stop=1e-13)
print("oracle %.1fs"%(time.time()-t), "oracle status max",ref['status'].max(),"iters max",ref['iters'].max())
e0=np.abs(out['u0']-ref['u0']).max(axis=1)/3.4
eU=np.abs(out['U']-ref['U']).reshape(B,-1).max(axis=1)/3.4
for nm,e in (("u0",e0),("U",eU)):
    print(nm,"max %.2e p99.99 %.2e p99.9 %.2e p99 %.2e p90 %.2e med %.2e  frac>1e-4: %.2e"%(e.max(),np.percentile(e,99.99),np.percentile(e,99.9),np.percentile(e,99),np.percentile(e,90),np.median(e),(e>1e-4).mean()))
print("status counts",np.bincount(out['status']),"iters mean %.2f max %d"%(out['iters'].mean(),out['iters'].max()))
w=np.argsort(-e0)[:8]
for i in w: print(" worst inst",i,"err %.2e"%e0[i],"iters",out['iters'][i],"status",out['status'][i],"oracle iters",ref['iters'][i])
