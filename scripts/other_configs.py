"""Throughput of the non-headline BASELINE configs on one GPU (parity-test shapes, not bench lines)."""
import sys, time
sys.path.insert(0,'/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0,'/root/repo')
import numpy as np, torch, ft_mpc_amd
ONLY=sys.argv[1:]     # substrings of the names to run (default: all)
def run(name,B,N,NT,nf,dtype,seed,reps=3,sel="auto"):
    if ONLY and not any(o in name for o in ONLY): return
    mpc=ft_mpc_amd.BatchedMPC(N=N,NT=NT,dtype=dtype,kernel_select=sel)
    x0,ub,stuck,xref=ft_mpc_amd.make_synthetic_batch(B,N,NT,nf,seed)
    dev=torch.device('cuda:0'); t=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dx0,dub,dst,dxr=t(x0),t(ub),t(stuck),t(xref.reshape(-1,order='F'))
    u0=torch.zeros(B,NT,dtype=torch.float64,device=dev); st=torch.zeros(B,dtype=torch.int32,device=dev); it=torch.zeros(B,dtype=torch.int32,device=dev)
    mpc.reserve(B); mpc.set_profiling(True); s=torch.cuda.current_stream().cuda_stream
    best=1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0=time.perf_counter()
        mpc.solve_device(B,dx0.data_ptr(),dub.data_ptr(),dst.data_ptr(),dxr.data_ptr(),0,0,0,0,u0.data_ptr(),0,st.data_ptr(),it.data_ptr(),s)
        torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
    print("%-34s B=%6d N=%2d NT=%2d faults=%d %s: %8.2f ms  %9.0f QP/s  iters %.2f  not-converged %d  kernels %s"%(name,B,N,NT,nf,dtype,best*1e3,B/best,it.float().mean().item(),int((st!=0).sum()),{k:round(v,2) for k,v in mpc.last_kernel_ms().items()}))
    mpc.close()
run("cfg2 single fault",4096,20,8,1,"f32",1002)
run("cfg2-shape at 65536",65536,20,8,1,"f32",1002)
run("nominal 8 thrusters",65536,20,8,0,"f32",1001)
run("cfg3 double fault (headline)",65536,20,8,2,"f32",1003)
run("cfg4 shard (32768/GPU)",32768,20,8,2,"f32",1004)
run("reference vehicle N=15 NT=16 2f (fp32, kernel 10)",4096,15,16,2,"f32",1011)
run("reference vehicle, nominal (n=240)",4096,15,16,0,"f32",1012)
run("reference vehicle N=15 NT=16 2f",4096,15,16,2,"f64",1011)
run("cfg5 shard (2048/GPU) N=40 NT=16 (wrench-space f64)",2048,40,16,2,"f64",1005)
run("cfg5 shard, dense f64 kernel",2048,40,16,2,"f64",1005,sel="dense")
run("N=20 NT=16 2f fp32 (kernel 10, eight tiles a side)",4096,20,16,2,"f32",1013)
run("N=20 NT=16 nominal fp32",4096,20,16,0,"f32",1014)
run("N=20 NT=16 2f f64 dense (what it ran on before)",2048,20,16,2,"f64",1013,sel="dense")
