import os, sys
import numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/fault-tolerant-mpc_amd')
import ft_mpc_amd
from oracle import qp_oracle as qo, c_oracle as co
N,NT,seed=21,11,5
rng=np.random.default_rng(900+seed); B=96
D=rng.standard_normal((6,NT))*np.array([1,1,1,0.3,0.3,0.3])[:,None]
cfg=qo.QPConfig(N=N,NT=NT,D=D)
x0,ub,stuck,xref=qo.make_batch(B,N,NT,1,4100+seed)
for b in range(B):
    k=int(rng.integers(0,3)); idx=rng.choice(np.flatnonzero(ub[b]>0),k,replace=False); ub[b,idx]=0.0; stuck[b,idx]=rng.uniform(0,1,k)*3.4
W=np.ascontiguousarray(rng.uniform(0,0.3,(B,N,NT))*ub[:,None,:])
ref=co.solve_batch(cfg,x0,ub,stuck,xref,warmU=W.copy(),nthreads=8,max_iters=60,mu_stop=1e-13)
ok=ref['status']==0
for ws in ('auto','dense'):
    for dt in ('f32','f64'):
        m=ft_mpc_amd.BatchedMPC(ft_mpc_amd.MPCConfig(N=N,NT=NT,D=D,dtype=dt,kernel_select=ws))
        out=m.solve(x0,ub,stuck,xref.reshape(-1,order='F'),warmU=W.copy(),return_U=True)
        e=np.abs(out['u0']-ref['u0']).max(axis=1)/3.4
        print('ws',ws,dt,'status',np.bincount(out['status'],minlength=3),'err max %.2e p90 %.2e med %.2e'%(e[ok].max(),np.percentile(e[ok],90),np.median(e[ok])),'iters',out['iters'].mean(), 'worst',int(e.argmax()), 'oracle iters worst', ref['iters'][e.argmax()], out['iters'][e.argmax()])
        m.close()
