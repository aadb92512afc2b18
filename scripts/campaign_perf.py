"""Closed-loop Monte-Carlo campaign throughput: B vehicles x T steps on the device (ftmpc_simulate_batch)."""
import sys, time
sys.path.insert(0,'/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0,'/root/repo')
import numpy as np, ft_mpc_amd
B=int(sys.argv[1]) if len(sys.argv)>1 else 65536
T=int(sys.argv[2]) if len(sys.argv)>2 else 20
N,NT=20,8
mpc=ft_mpc_amd.BatchedMPC(N=N,NT=NT)
x0,ub,stuck,_=ft_mpc_amd.make_synthetic_batch(B,N,NT,2,1003)
xr=np.zeros((9,T+N)); xr[8]=0.6
mpc.simulate(x0[:256],ub[:256],stuck[:256],xr,T)   # warm-up / workspace
t0=time.perf_counter(); out=mpc.simulate(x0,ub,stuck,xr,T,seed=3); dt=time.perf_counter()-t0
c=np.linalg.norm(out['x'][:,0:3],axis=1)
print("campaign B=%d T=%d: %.1f ms total, %.0f closed-loop QP-steps/s (incl. H2D/D2H of states), not converged per step: %s"%(B,T,dt*1e3,B*T/dt,out['not_converged'].tolist()))
print("median |p| start %.2f -> end %.2f m"%(np.median(np.linalg.norm(x0[:,0:3],axis=1)),np.median(c)))
