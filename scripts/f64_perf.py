"""Throughput of the float64 shapes (device-resident buffers) for the library FTMPC_LIB names (default: the shipped one).
Usage: [FTMPC_LIB=...] python scripts/f64_perf.py [kernel_select]"""
import sys, time
sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0, '/root/repo')
import numpy as np, torch, ft_mpc_amd
sel = sys.argv[1] if len(sys.argv) > 1 else "auto"
def run(name, B, N, NT, nf, seed, reps=3):
    mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype="f64", kernel_select=sel)
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, nf, seed)
    dev = torch.device('cuda:0'); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dx0, dub, dst, dxr = t(x0), t(ub), t(stuck), t(xref.reshape(-1, order='F'))
    u0 = torch.zeros(B, NT, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
    mpc.reserve(B); mpc.set_profiling(True); s = torch.cuda.current_stream().cuda_stream
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mpc.solve_device(B, dx0.data_ptr(), dub.data_ptr(), dst.data_ptr(), dxr.data_ptr(), 0, 0, 0, 0, u0.data_ptr(), 0, st.data_ptr(), it.data_ptr(), s)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-26s %-44s B=%5d: %8.2f ms %8.0f QP/s iters %.2f bad %d %s" % (str(ft_mpc_amd._lib.library_path().name), name, B, best * 1e3, B / best, it.float().mean().item(), int((st != 0).sum()), {k: round(v, 2) for k, v in mpc.last_kernel_ms().items()}), flush=True)
    mpc.close()
run("cfg5 shard N=40 NT=16 2f", 2048, 40, 16, 2, 1005)
run("cfg5 shard N=40 NT=16 2f, 4096", 4096, 40, 16, 2, 1005)
run("reference vehicle N=15 NT=16 2f", 4096, 15, 16, 2, 1011)
run("N=20 NT=16 2f", 4096, 20, 16, 2, 1013)
run("cfg5 N=40 NT=16 2f, 16384", 16384, 40, 16, 2, 1005, reps=2)
run("reference vehicle N=15 NT=16 2f, 16384", 16384, 15, 16, 2, 1011, reps=2)
