"""Diagnostic: per-phase cycle shares of the float64 solve kernel (stamps build, wave 0 of each workgroup)."""
import sys, ctypes as C, os
sys.path.insert(0,'/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0,'/root/repo')
import numpy as np
from pathlib import Path
from ft_mpc_amd import _lib
_lib._SO = Path(os.environ["FTMPC_LIB"]) if os.environ.get("FTMPC_LIB") else _lib._HERE / "libftmpc_hip_stamps.so"
import ft_mpc_amd
N=int(sys.argv[1]) if len(sys.argv)>1 else 15
NT=int(sys.argv[2]) if len(sys.argv)>2 else 16
B=int(sys.argv[3]) if len(sys.argv)>3 else 512
SEL=sys.argv[4] if len(sys.argv)>4 else "auto"
mpc=ft_mpc_amd.BatchedMPC(N=N,NT=NT,dtype="f64",max_iters=40,kernel_select=SEL)
x0,ub,stuck,xref=ft_mpc_amd.make_synthetic_batch(B,N,NT,2,1003)
out=mpc.solve(x0,ub,stuck,xref.reshape(-1,order='F'))
cnt=min(B,512)
buf=np.zeros((cnt,12),np.uint64)
f=mpc.lib.ftmpc_debug_read_stamps; f.argtypes=[C.c_void_p,C.c_int64,C.c_void_p]
assert f(mpc._h,cnt,buf.ctypes.data_as(C.c_void_p))==0
names=["prologue","phase1: E panels","phase2: H tiles","chol: barrier + row staging","gradient+mu","chol: W phase","solves(2)","elementwise","chol: diagonal tile + potrf (wave 0)","output","chol: off-diagonal stream (wave 0)","chol: wait for the other waves (wave 0)"]
mpc.set_profiling(True); mpc.solve(x0,ub,stuck,xref.reshape(-1,order='F')); ran=mpc.last_kernel_ms(); print(ran)
if "ftmpc_solve_ws64_kernel" in ran:
    names=["prologue + output","phase1: E panels","phase2: H_w tiles","chol: barrier + row staging","H_w post-pass | elementwise, S blocks, wrench images","solve with the factor of K (x2)","L' t and L p (x2 each)","X = (L' S) L","chol: diagonal tile + potrf (wave 0)","chol: W phase","chol: off-diagonal stream (wave 0)","chol: wait for the other waves (wave 0)"]
if "ftmpc_solve_ric64_kernel" in ran:
    names=["prologue + start gradient","Riccati sweep (factors W, Y of every stage)","backward vector sweeps (x2)","forward vector sweeps (x2)","element-wise (Mehrotra)","output","","","","","",""]
m=buf.astype(np.float64).mean(axis=0); tot=m.sum(); it=out['iters'].mean()
print("N=%d NT=%d B=%d iters mean %.2f   total ticks/QP %.0f"%(N,NT,B,it,tot))
for n_,v in zip(names,m):
    if n_: print("  %-44s %10.0f  %5.1f%%   per-iter %8.0f"%(n_,v,100*v/tot,v/it))
