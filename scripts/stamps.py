"""Diagnostic: per-phase cycle shares of the fp32 solve kernel (stamps build, see csrc/Makefile)."""
import sys, ctypes as C
sys.path.insert(0,'/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0,'/root/repo')
import numpy as np
from ft_mpc_amd import _lib
import os
from pathlib import Path
_lib._SO = Path(os.environ["FTMPC_LIB"]) if os.environ.get("FTMPC_LIB") else _lib._HERE / "libftmpc_hip_stamps.so"
import ft_mpc_amd
B=int(sys.argv[1]) if len(sys.argv)>1 else 4096
nf=int(sys.argv[2]) if len(sys.argv)>2 else 2
N=int(sys.argv[3]) if len(sys.argv)>3 else 20
NT=int(sys.argv[4]) if len(sys.argv)>4 else 8
mpc=ft_mpc_amd.BatchedMPC(N=N,NT=NT,kernel_select=(sys.argv[5] if len(sys.argv)>5 else 'auto'))
x0,ub,stuck,xref=ft_mpc_amd.make_synthetic_batch(B,N,NT,nf,1003)
out=mpc.solve(x0,ub,stuck,xref.reshape(-1,order='F'))
cnt=min(B,4096)
buf=np.zeros((cnt,12),np.uint64)
f=mpc.lib.ftmpc_debug_read_stamps; f.argtypes=[C.c_void_p,C.c_int64,C.c_void_p]
assert f(mpc._h,cnt,buf.ctypes.data_as(C.c_void_p))==0
names=["prologue","build:propagate","build:mfma","finalize+store","matvec","chol","solves(2)","elementwise","refine (f64 grad)","output","x10","x11"]
SEL=sys.argv[5] if len(sys.argv)>5 else "auto"
if N*NT>160 and SEL=="auto": names=["prologue","build: propagate","build: mfma","start gradient","factor of H_w (+ L tiles)","P = S L","X = P' L","rdg + S blocks","refine (f64 grad)","factor of K","solves(2) + elementwise","output"]
elif N*NT>160: names=["prologue","build 1: condense","build 2: H tiles","start gradient","factor","solves(2)","elementwise","refine (f64 grad)","output","factor: wave 0 on the chain (3-4 of 14 phases)","factor: wave 0 as a helper","factor: barrier wait (wave 0)"]
m=buf.astype(np.float64).mean(axis=0); tot=m.sum()
print("iters mean %.2f   total cycles/QP %.0f"%(out['iters'].mean(),tot))
for n_,v in zip(names,m):
    if n_: print("  %-18s %10.0f  %5.1f%%   per-iter %8.0f"%(n_,v,100*v/tot,v/out['iters'].mean()))
