"""Diagnostic: accuracy and rate of the fp32 path against the IPM stopping threshold mu_stop (full config-3 batch)."""
import sys, time, os
sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0, '/root/repo')
import numpy as np, torch, ft_mpc_amd
from oracle import c_oracle as co, qp_oracle as qo
B, N, NT = 65536, 20, 8
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, nf, 1003)
ref = co.solve_batch(qo.QPConfig(N=N, NT=NT), x0, ub, stuck, xref, nthreads=min(16, len(os.sched_getaffinity(0))), max_iters=60, mu_stop=1e-13, return_U=False)
dev = torch.device('cuda:0'); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
dx0, dub, dst, dxr = t(x0), t(ub), t(stuck), t(xref.reshape(-1, order='F'))
u0 = torch.zeros(B, NT, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for ms in [1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6]:
    mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, mu_stop=ms); mpc.reserve(B)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mpc.solve_device(B, dx0.data_ptr(), dub.data_ptr(), dst.data_ptr(), dxr.data_ptr(), 0, 0, 0, 0, u0.data_ptr(), 0, st.data_ptr(), it.data_ptr(), s)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    err = np.abs(u0.cpu().numpy() - ref['u0']).max(axis=1) / 3.4
    print(f"mu_stop {ms:.0e}: {B/best/1e6:.3f} M QP/s  iters {it.float().mean().item():.2f}  err max {err.max():.2e} p99.9 {np.quantile(err, 0.999):.2e} median {np.median(err):.2e}  not-conv {int((st != 0).sum())}", flush=True)
    mpc.close()
