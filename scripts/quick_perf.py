import sys, time
sys.path.insert(0,'/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0,'/root/repo')
import numpy as np, torch
import os
import ft_mpc_amd
from ft_mpc_amd import _lib
if os.environ.get('FTMPC_LIB'):  # diagnostic: time another build of the library
    from pathlib import Path
    _lib._SO = Path(os.environ['FTMPC_LIB'])
B=int(sys.argv[1]) if len(sys.argv)>1 else 65536
nf=int(sys.argv[2]) if len(sys.argv)>2 else 2
N,NT=20,8
mpc=ft_mpc_amd.BatchedMPC(N=N,NT=NT)
x0,ub,stuck,xref=ft_mpc_amd.make_synthetic_batch(B,N,NT,nf,1003)
dev=torch.device('cuda:0')
t=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
dx0,dub,dst,dxr=t(x0),t(ub),t(stuck),t(xref.reshape(-1,order='F'))
u0=torch.zeros(B,NT,dtype=torch.float64,device=dev); st=torch.zeros(B,dtype=torch.int32,device=dev); it=torch.zeros(B,dtype=torch.int32,device=dev)
mpc.reserve(B); mpc.set_profiling(True)
s=torch.cuda.current_stream().cuda_stream
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    mpc.solve_device(B,dx0.data_ptr(),dub.data_ptr(),dst.data_ptr(),dxr.data_ptr(),0,0,0,0,u0.data_ptr(),0,st.data_ptr(),it.data_ptr(),s)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("B",B,"nfault",nf,"wall %.2f ms -> %.0f QP/s"%(dt*1e3,B/dt),"kernels ms",mpc.last_kernel_ms(),"iters mean %.2f max %d"%(it.float().mean().item(),it.max().item()),"status!=0:",int((st!=0).sum()))
