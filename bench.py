#!/usr/bin/env python3
"""bench.py -- MPC QP steps/s of the HIP path on N GPUs of one node (driver contract).

A "step" is one pass of the hot path (linearise -> condense -> IPM -> thrust command) over one
batch of synthetic random-pose / random-double-fault instances already resident in HBM.
Workload = BASELINE.json configs[2]: batch 65536 per GPU, N=20, 8 thrusters, two random
faulted thrusters, cold start, hover reference.  Batches shard across ranks with no data-path
collective (weak scaling: every rank owns its own 65536 instances).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "fault-tolerant-mpc_amd"))

F32_PEAK_TFLOPS = 157.3  # MI355X fp32 MFMA == fp32 vector peak (MI355X_MICROARCH.md)
F64_PEAK_TFLOPS = 78.6   # fp64 vector/matrix peak


def algorithmic_flops(N, na, iters):
    """SURVEY.md section 8(d) flop model evaluated at the instance's ACTIVE dimension n = N*na and its
    executed IPM iterations (DESIGN.md 'Work model')."""
    nx, nq, ng = 13, 9, 6
    n = N * na
    f_lin = N * (8 * nx ** 3 + 8 * nx ** 2 * ng + 2 * nx * ng * na)
    f_cond = N * (N - 1) * nx ** 2 * na
    f_h = 2 * nq * na ** 2 * N * (N + 1) * (N + 2) / 6
    f_g = nq * na * N * (N + 1)
    f_it = n ** 3 / 3 + 2 * n ** 2 + 10 * n
    return f_lin + f_cond + f_h + f_g + iters * f_it


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--thrusters", type=int, default=8)
    ap.add_argument("--faults", type=int, default=2)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="arithmetic of the KKT/IPM solve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ft_mpc_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        sys.exit("bench.py --gpus N (N > 1) must be launched with one rank per GPU: python -m torch.distributed.run "
                 "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if world > 1:
        # The data path has NO collective (independent instances, batch-sharded): torch.distributed is
        # only the timing bracket (barrier + max over ranks), on CPU tensors over gloo.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    # FTMPC_BENCH_SINGLE_DEVICE=1 (tests only) maps every rank to cuda:0 to rehearse the multi-rank path on a 1-GPU box
    if os.environ.get("FTMPC_BENCH_SINGLE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    N, NT, B = args.horizon, args.thrusters, args.batch
    mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, device_id=local, dtype=args.dtype)
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, args.faults, 1003 + rank)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_x0, d_ub, d_st, d_xr = to(x0), to(ub), to(stuck), to(xref.reshape(-1, order="F"))
    d_u0 = torch.zeros(B, NT, dtype=torch.float64, device=dev)
    d_status = torch.zeros(B, dtype=torch.int32, device=dev)
    d_iters = torch.zeros(B, dtype=torch.int32, device=dev)
    mpc.reserve(B)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        mpc.solve_device(B, d_x0.data_ptr(), d_ub.data_ptr(), d_st.data_ptr(), d_xr.data_ptr(), 0, 0, 0, 0,
                         d_u0.data_ptr(), 0, d_status.data_ptr(), d_iters.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel device time of one more (untimed) step: hipEvents on the launch stream
    mpc.set_profiling(True)
    kms = []
    for _ in range(3):
        step()
        torch.cuda.synchronize()
        kms.append(mpc.last_kernel_ms())
    mpc.set_profiling(False)
    names = sorted(kms[0])
    med = {k: float(np.median([r.get(k, 0.0) for r in kms])) for k in names}
    lin_ms = med.pop("ftmpc_linearize_kernel")
    dom = max(med, key=med.get)          # dominant kernel of the step
    sol_ms = med[dom]

    iters = d_iters.cpu().numpy()
    status = d_status.cpu().numpy()
    u0_gpu = d_u0.cpu().numpy()

    # SURVEY.md 8(d) asks for the warm-started rate next to the cold one (never `value`): same states,
    # linearised about the previous solution shifted by one stage (spiraling_mpc.py:324-334)
    d_U = torch.zeros(B, N, NT, dtype=torch.float64, device=dev)
    mpc.solve_device(B, d_x0.data_ptr(), d_ub.data_ptr(), d_st.data_ptr(), d_xr.data_ptr(), 0, 0, 0, 0,
                     d_u0.data_ptr(), d_U.data_ptr(), d_status.data_ptr(), d_iters.data_ptr(), stream)
    torch.cuda.synchronize()
    d_warm = torch.cat([d_U[:, 1:], d_U[:, -1:]], dim=1).contiguous()
    d_u0w = torch.zeros_like(d_u0)
    d_itw = torch.zeros_like(d_iters)

    def step_warm():
        mpc.solve_device(B, d_x0.data_ptr(), d_ub.data_ptr(), d_st.data_ptr(), d_xr.data_ptr(), 0, 0, 0, d_warm.data_ptr(),
                         d_u0w.data_ptr(), 0, d_status.data_ptr(), d_itw.data_ptr(), stream)

    step_warm()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_warm()
    barrier()
    warm_elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([warm_elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        warm_elapsed = float(t.item())
    warm_iters = float(d_itw.float().mean().item())
    warm_du0 = float((d_u0w - d_u0).abs().amax(dim=1).median().item() / 3.4)
    na = (ub > 0).sum(axis=1)
    flops = float(sum(algorithmic_flops(N, int(a), int(k)) for a, k in zip(na, iters)))

    if rank == 0:
        value = args.steps * B * world / elapsed
        achieved = flops / (sol_ms * 1e-3) / 1e12
        peak = F64_PEAK_TFLOPS if (args.dtype == "f64" or N * NT > 160) else F32_PEAK_TFLOPS
        # HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this process, so
        # the committed rocprofv3 --pmc summary of the same command is reported (per launch, with the
        # gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md); null when no summary matches.
        traffic, traffic_src = None, None
        tfile = ROOT / "profiles" / "traffic_latest.json"
        if tfile.exists():
            tj = json.loads(tfile.read_text())
            if tj.get("kernel") == dom and tj.get("batch") == B and tj.get("horizon") == N and tj.get("thrusters") == NT:
                traffic, traffic_src = tj["bytes_per_launch"], tj["source"]
        line = {
            "metric": "MPC QP steps/s (whole node) at N=20, 8 thrusters, batch 65536",
            "value": value, "unit": "QP-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if (args.dtype == "f64" or N * NT > 160) else "f32", "data": "synthetic",
            "config": {"workload": f"batch {B}/GPU, N={N}, {NT} thrusters, random {args.faults}-fault Monte-Carlo, "
                                   f"cold start, hover reference (BASELINE configs[2])",
                       "batch_per_gpu": B, "horizon": N, "thrusters": NT, "faults": args.faults,
                       "ipm_iters_mean": float(iters.mean()), "ipm_iters_max": int(iters.max()),
                       "not_converged": int((status != 0).sum()), "parallelism": f"batch-sharded x{world}, no collective"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": dom, "kernel_ms": sol_ms, "other_kernels_ms": {"ftmpc_linearize_kernel": lin_ms, **{k: v for k, v in med.items() if k != dom}},
                         "flops_per_launch": flops},
        }
        line["warm_start"] = {"value": args.steps * B * world / warm_elapsed, "unit": "QP-steps/s", "ipm_iters_mean": warm_iters,
                              "median_u0_change_over_fmax": warm_du0,
                              "note": "same states, linearised about the previous solution shifted by one stage; reported beside the cold-start `value`"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import c_oracle, qp_oracle
            cores = os.cpu_count() or 1
            qcfg = qp_oracle.QPConfig(N=N, NT=NT)
            kw = dict(max_iters=30, mu_stop=1e-11, return_U=False) if peak == F32_PEAK_TFLOPS else dict(max_iters=30, mu_stop=1e-13, return_U=False)
            t1 = time.perf_counter()
            c_oracle.solve_batch(qcfg, x0[:4 * cores], ub[:4 * cores], stuck[:4 * cores], xref, nthreads=cores, **kw)
            pilot = (time.perf_counter() - t1) / (4 * cores)
            sample = int(min(B, max(8 * cores, 12.0 / max(pilot, 1e-6))))
            t1 = time.perf_counter()
            ref = c_oracle.solve_batch(qcfg, x0[:sample], ub[:sample], stuck[:sample], xref, nthreads=cores, **kw)
            cpu_t = time.perf_counter() - t1
            err = float(np.abs(ref["u0"] - u0_gpu[:sample]).max() / 3.4)
            # the same port on ONE host thread (small sample, ~2 s)
            n1 = int(min(sample, 2048))
            t1 = time.perf_counter()
            c_oracle.solve_batch(qcfg, x0[:n1], ub[:n1], stuck[:n1], xref, nthreads=1, **kw)
            one_t = time.perf_counter() - t1
            line["cpu_baseline"] = {"value": sample / cpu_t, "unit": "QP-steps/s", "cores": cores, "kind": "port",
                                    "sample": f"first {sample} instances of the same batch, C float64 restatement "
                                              f"(oracle/ftmpc_oracle.c), {cores} threads, {cpu_t:.1f} s; "
                                              f"reference IPOPT path not runnable offline",
                                    "gpu_vs_port_max_u0_err_over_fmax": err,
                                    "single_thread": {"value": n1 / one_t, "unit": "QP-steps/s", "sample": f"first {n1} instances, 1 thread, {one_t:.1f} s"}}
        print(json.dumps(line))
    mpc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
