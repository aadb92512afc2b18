#!/usr/bin/env python3
"""bench.py -- MPC QP steps/s of the HIP path on N GPUs of one node (driver contract).

A "step" is one pass of the hot path (linearise -> condense -> IPM -> thrust command) over one
batch of synthetic random-pose / random-double-fault instances already resident in HBM.
Workload = BASELINE.json configs[2]: batch 65536 per GPU, N=20, 8 thrusters, two random
faulted thrusters, cold start, hover reference.  Batches shard across GPUs with no data-path
collective (weak scaling: every GPU owns its own 65536 instances).

Three ways to run N > 1:
  * `python bench.py --gpus N`                      this process drives all N GPUs itself through the library's
                                                    in-process multi-GPU driver (include/ftmpc.h ftmpc_multi_*:
                                                    one host thread + one handle + one stream set per device);
  * `python -m torch.distributed.run ... bench.py --gpus N`   one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE from
                                                    the environment), gloo only for the timing bracket;
  * FTMPC_BENCH_SINGLE_DEVICE=1 (tests)             either of the above with every rank / slot on device 0.
"""
import argparse
import hashlib
import json
import math
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
METRIC = "MPC QP steps/s (whole node) at N=20, 8 thrusters, batch 65536"


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


def executed_flops_wrench(N, na, iters):
    """Work the wrench-space kernels (ftmpc_solve_ws32_kernel, ftmpc_solve_ws64_kernel) EXECUTE for one instance with `na`
    healthy thrusters: condensing in the six wrench components, one Cholesky of the 6N x 6N Hessian H_w, and per
    interior-point iteration the assembly of K = I + L'SL (stage blocks, P' = L'S with S block diagonal, the lower triangle
    of X = P'L), its Cholesky, and two Newton solves through wrench space (two triangular products with L and two
    triangular sweeps each) plus the element-wise work over the N na thruster variables (kernels 8 / 9)."""
    nx, nq, ng = 13, 9, 6
    nw, nt = 6 * N, N * na
    f_lin = N * (8 * nx ** 3 + 8 * nx ** 2 * ng + 2 * nx * ng * na)
    f_cond = N * (N - 1) * nx ** 2 * ng
    f_h = 2 * nq * ng ** 2 * N * (N + 1) * (N + 2) / 6
    f_g = nq * ng * N * (N + 1)
    f_once = nw ** 3 / 3 + 2 * nw ** 2 + 2 * ng * nt        # factor of H_w, start gradient
    f_it = (42 * N * na) + 6 * nw ** 2 + nw ** 3 / 3 + nw ** 3 / 3 + 2 * (4 * nw ** 2 + 4 * ng * nt) + 30 * nt
    return f_lin + f_cond + f_h + f_g + f_once + iters * f_it


def executed_flops_riccati(N, na, iters):
    """Work the Riccati kernel (ftmpc_solve_ric64_kernel) EXECUTES for one instance with `na` healthy thrusters: no condensing
    and no Hessian; the start gradient by one forward and one adjoint sweep; per interior-point iteration the backward Riccati
    sweep (per stage S A, S Bt, Bt'SBt, Bt'SA, A'SA, the Cholesky + inverse of the na x na block, W Rux, Y'Y) and two right-hand
    sides of one backward and one forward vector sweep each (four matrix-vector products per stage and sweep), plus the
    element-wise work over the N na thruster variables (kernel 12)."""
    nx, ng = 13, 6
    f_lin = N * (8 * nx ** 3 + 8 * nx ** 2 * ng + 2 * nx * ng * na)
    mv = 2 * (nx * nx + 2 * nx * na + na * na)                     # A x, Bt u / Bt' s, Y x / Y' w, W v: one sweep stage
    f_once = N * 2 * mv
    f_stage = 2 * nx ** 3 + 2 * nx * nx * na + 2 * nx * na * na + 2 * na * nx * nx + 2 * nx ** 3 + 2 * na ** 3 / 3 + na * na * nx + 2 * nx * nx * na
    f_it = N * (f_stage + 4 * mv) + 40 * N * na
    return f_lin + f_once + iters * f_it


def batch_flops(N, ub, iters, model=algorithmic_flops):
    if model is not algorithmic_flops:
        na = (ub > 0).sum(axis=1)
        tot = 0.0
        for a in np.unique(na):
            sel = na == a
            tot += float(model(N, int(a), 0)) * int(sel.sum()) + float(model(N, int(a), 1) - model(N, int(a), 0)) * float(iters[sel].sum())
        return tot
    return _batch_flops_dense(N, ub, iters)


def workload_name(args, B, N, NT, f64):
    """What the arguments describe, and which BASELINE.json config that is (if any)."""
    base = None
    if (N, NT) == (20, 8) and not f64:
        if args.faults == 1 and B == 4096:
            base = "configs[1]"
        elif args.faults == 2 and B == 65536:
            base = "configs[2]"
        elif args.faults == 2 and B == 32768:
            base = "configs[3] shard (262144 / 8 GPUs)"
    if (N, NT) == (40, 16) and f64 and args.faults == 2 and B == 2048:
        base = "configs[4] shard (16384 / 8 GPUs)"
    w = f"batch {B}/GPU, N={N}, {NT} thrusters, random {args.faults}-fault Monte-Carlo, cold start, hover reference"
    return w + (f" (BASELINE {base})" if base else " (not a BASELINE config: see config.batch_per_gpu / horizon / thrusters)")


def _batch_flops_dense(N, ub, iters):
    na = (ub > 0).sum(axis=1)
    tot = 0.0
    for a in np.unique(na):
        sel = na == a
        tot += float(algorithmic_flops(N, int(a), 0)) * int(sel.sum()) + float(algorithmic_flops(N, int(a), 1) - algorithmic_flops(N, int(a), 0)) * float(iters[sel].sum())
    return tot


def csrc_hash():
    """sha256 over the kernel sources: the committed PMC traffic figure is only quoted for the code it was measured on."""
    h = hashlib.sha256()
    for f in sorted((ROOT / "fault-tolerant-mpc_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".h", ".inc"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def host_cpu_info():
    """Cores this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        t = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if t[0] != "max":
            quota = float(t[0]) / float(t[1])
    except Exception:
        try:
            q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            p = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    usable = aff if quota is None else max(1, min(aff, int(math.floor(quota + 1e-9))))
    return {"os_cpu_count": os.cpu_count(), "sched_affinity": aff, "cgroup_quota_cpus": quota, "usable": usable}


def traffic_for(dom, B, N, NT):
    tfile = ROOT / "profiles" / "traffic_latest.json"
    if not tfile.exists():
        return None, None
    tj = json.loads(tfile.read_text())
    if tj.get("kernel") == dom and tj.get("batch") == B and tj.get("horizon") == N and tj.get("thrusters") == NT:
        if tj.get("csrc_sha") == csrc_hash():
            return tj["bytes_per_launch"], tj["source"]
        return None, f"stale: {tj.get('source')} was measured on csrc {tj.get('csrc_sha')}, this is {csrc_hash()}"
    return None, None


def cpu_baseline(N, NT, x0, ub, stuck, xref, u0_gpu, f32):
    """The C float64 restatement (oracle/ftmpc_oracle.c) on this box's usable host cores, on a bounded prefix of the
    same batch (about 12 s of CPU work), plus the same port on one thread."""
    from oracle import c_oracle, qp_oracle
    info = host_cpu_info()
    cores = info["usable"]
    B = x0.shape[0]
    qcfg = qp_oracle.QPConfig(N=N, NT=NT)
    kw = dict(max_iters=30, mu_stop=1e-11 if f32 else 1e-13, return_U=False)
    npilot = min(B, 4 * cores)
    t1 = time.perf_counter()
    c_oracle.solve_batch(qcfg, x0[:npilot], ub[:npilot], stuck[:npilot], xref, nthreads=cores, **kw)
    pilot = (time.perf_counter() - t1) / npilot
    sample = int(min(B, max(8 * cores, 12.0 / max(pilot, 1e-6))))
    t1 = time.perf_counter()
    ref = c_oracle.solve_batch(qcfg, x0[:sample], ub[:sample], stuck[:sample], xref, nthreads=cores, **kw)
    cpu_t = time.perf_counter() - t1
    err = float(np.abs(ref["u0"] - u0_gpu[:sample]).max() / 3.4)
    n1 = int(min(sample, 2048))
    t1 = time.perf_counter()
    c_oracle.solve_batch(qcfg, x0[:n1], ub[:n1], stuck[:n1], xref, nthreads=1, **kw)
    one_t = time.perf_counter() - t1
    return {"value": sample / cpu_t, "unit": "QP-steps/s", "cores": cores, "kind": "port",
            "sample": f"first {sample} instances of the same batch, C float64 restatement (oracle/ftmpc_oracle.c), "
                      f"{cores} threads, {cpu_t:.1f} s; reference IPOPT path not runnable offline",
            "host": info,
            "gpu_vs_port_max_u0_err_over_fmax": err,
            "single_thread": {"value": n1 / one_t, "unit": "QP-steps/s", "sample": f"first {n1} instances, 1 thread, {one_t:.1f} s"},
            "parallel_speedup": (sample / cpu_t) / (n1 / one_t)}


def make_line(args, world, elapsed, B, N, NT, iters, status, ub, kernel_ms, parallelism):
    lin_ms = kernel_ms.pop("ftmpc_linearize_kernel", 0.0)
    dom = max(kernel_ms, key=kernel_ms.get)
    sol_ms = kernel_ms[dom]
    flops = batch_flops(N, ub, iters)
    f64 = "f64" in dom or "ws64" in dom or "ric64" in dom
    peak = F64_PEAK_TFLOPS if f64 else F32_PEAK_TFLOPS
    achieved = flops / (sol_ms * 1e-3) / 1e12
    traffic, traffic_src = traffic_for(dom, B, N, NT)
    # Kernels 8 / 9 / 10 solve the Newton systems through the 6N-variable wrench-space form and kernel 12 by the Riccati recursion:
    # they EXECUTE less than SURVEY.md 8(d)'s dense model (the work the QP step stands for).  For them `achieved` / `frac` are
    # the EXECUTED work -- a fraction of the peak the silicon really delivered -- and the dense-model figure is the note.
    note = {}
    exec_model = executed_flops_riccati if "ric64" in dom else (executed_flops_wrench if ("ws32" in dom or "wsw32" in dom or "ws64" in dom) else None)
    if exec_model is not None:      # (names as ROUTED on this handle: the dense wg32 kernel is not one of them)
        ex = batch_flops(N, ub, iters, exec_model)
        note = {"dense_model": {"flops_per_launch": flops, "achieved": achieved, "frac": achieved / peak,
                                "model": "bench.py:algorithmic_flops (SURVEY.md 8(d): dense condensed IPM at the active dimension and the executed iterations)"},
                "work_model": f"bench.py:{exec_model.__name__}: the work this kernel performs; dense_model.*: the work the QP step stands for "
                              "(this kernel does not form or factorise the (N na) x (N na) matrix)"}
        flops = ex
        achieved = ex / (sol_ms * 1e-3) / 1e12
    default_shape = (B, N, NT, args.faults) == (65536, 20, 8, 2) and not f64
    return {
        "metric": METRIC if default_shape else f"MPC QP steps/s (whole node) at N={N}, {NT} thrusters, batch {B}",
        "value": args.steps * B * world / elapsed, "unit": "QP-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if f64 else "f32", "data": "synthetic",
        "config": {"workload": workload_name(args, B, N, NT, f64),
                   "batch_per_gpu": B, "horizon": N, "thrusters": NT, "faults": args.faults,
                   "ipm_iters_mean": float(iters.mean()), "ipm_iters_max": int(iters.max()),
                   "not_converged": int((status != 0).sum()), "parallelism": parallelism},
        # neither HBM nor the matrix cores bind this path, the fp32 (fp64) vector-FMA rate and LDS do; gfx950's fp32
        # MFMA rate equals its fp32 vector rate (157.3 TFLOP/s), so `peak` is that one number either way
        "roofline": {"bound": "valu", "bound_note": "vector FMA + LDS latency of the per-wavefront Cholesky (SURVEY.md 8(d)); "
                                                    "not hbm (algorithmic traffic ~4 GB/s), not mfma (matrix cores carry the tile products only)",
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": dom, "kernel_ms": sol_ms,
                     "other_kernels_ms": {"ftmpc_linearize_kernel": lin_ms, **{k: v for k, v in kernel_ms.items() if k != dom}},
                     "flops_per_launch": flops, "csrc_sha": csrc_hash(), **note},
    }


# ------------------------------------------------------------------------------------------------
# N > 1 from the plain command: this process drives every GPU through the in-process multi-GPU driver
# ------------------------------------------------------------------------------------------------
def run_multi_inprocess(args):
    import ft_mpc_amd
    from ft_mpc_amd.sharding import MultiGPUMPC
    G, N, NT, B = args.gpus, args.horizon, args.thrusters, args.batch
    devices = [0] * G if os.environ.get("FTMPC_BENCH_SINGLE_DEVICE") == "1" else list(range(G))
    cfg = ft_mpc_amd.MPCConfig(N=N, NT=NT, dtype=args.dtype)
    m = MultiGPUMPC(cfg, devices=devices)

    def batch(total, seed0):
        # device g gets exactly what rank g of the torchrun path generates: its own seeded batch
        per = total // G
        parts = [ft_mpc_amd.make_synthetic_batch(per, N, NT, args.faults, seed0 + g) for g in range(G)]
        return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
                np.concatenate([p[2] for p in parts]), parts[0][3])

    def timed(mm, steps):
        t0 = time.perf_counter()
        mm.step(steps)            # returns when every device has finished its `steps` steps (max over devices)
        return time.perf_counter() - t0

    x0, ub, stuck, xref = batch(B * G, 1003)
    xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
    m.upload(x0, ub, stuck, xr)
    m.step(args.warmup)
    elapsed = timed(m, args.steps)
    m.set_profiling(True)
    kms = []
    for _ in range(3):
        m.step(1)
        kms.append(m.last_kernel_ms(0))
    m.set_profiling(False)
    med = {k: float(np.median([r.get(k, 0.0) for r in kms])) for k in sorted(kms[0])}
    out = m.download()
    lo, hi = m.shard_bounds(B * G, 0)
    line = make_line(args, G, elapsed, B, N, NT, out["iters"][lo:hi], out["status"], ub[lo:hi], med,
                     f"batch-sharded x{G}, one process, one host thread + handle + stream per GPU, no collective")
    line["config"]["ipm_iters_mean"] = float(out["iters"].mean())
    line["config"]["launcher"] = "in-process (ftmpc_multi_*)"
    # strong scaling (fixed total work): BASELINE configs[2] total (65 536) and configs[3] (262 144) over the G GPUs, against
    # the same totals on one GPU of this run; parallel efficiency = rate_G / (G * rate_1)
    strong = {}
    one = MultiGPUMPC(cfg, devices=[devices[0]])
    # (config 5 -- `--dtype f64 --horizon 40 --thrusters 16` -- has its own total: 16 384 over the node)
    totals = (16384,) if (args.dtype == "f64" and N == 40 and NT == 16) else (65536, 262144)
    for total in totals:
        total = (total // G) * G
        sx0, sub, sst, _ = batch(total, 1004)
        m.upload(sx0, sub, sst, xr)
        m.step(1)
        tG = timed(m, args.steps)
        one.upload(sx0, sub, sst, xr)
        one.step(1)
        t1 = timed(one, args.steps)
        strong[str(total)] = {"value": args.steps * total / tG, "unit": "QP-steps/s", "ms_per_step": tG / args.steps * 1e3,
                              "one_gpu_value": args.steps * total / t1, "parallel_efficiency": (t1 / tG) / G}
    one.close()
    line["strong_scaling"] = strong
    print(json.dumps(line))
    m.close()


# ------------------------------------------------------------------------------------------------
# one rank per GPU (torchrun) and the single-GPU default
# ------------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist
    import ft_mpc_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # The data path has NO collective (independent instances, batch-sharded): torch.distributed is
        # only the timing bracket (barrier + max over ranks), on CPU tensors over gloo.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
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

    def max_over_ranks(t):
        if world > 1:
            tt = torch.tensor([t], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return t

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # per-kernel device time of three more (untimed) steps: hipEvents on the launch stream
    mpc.set_profiling(True)
    kms = []
    for _ in range(3):
        step()
        torch.cuda.synchronize()
        kms.append(mpc.last_kernel_ms())
    mpc.set_profiling(False)
    med = {k: float(np.median([r.get(k, 0.0) for r in kms])) for k in sorted(kms[0])}

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
    warm_elapsed = max_over_ranks(time.perf_counter() - t0)
    warm_iters = float(d_itw.float().mean().item())
    warm_du0 = float((d_u0w - d_u0).abs().amax(dim=1).median().item() / 3.4)

    if rank == 0:
        line = make_line(args, world, elapsed, B, N, NT, iters, status, ub, med, f"batch-sharded x{world}, no collective")
        line["config"]["launcher"] = "torch.distributed.run (one rank per GPU)" if world > 1 else "single process"
        line["warm_start"] = {"value": args.steps * B * world / warm_elapsed, "unit": "QP-steps/s", "ipm_iters_mean": warm_iters,
                              "median_u0_change_over_fmax": warm_du0,
                              "note": "same states, linearised about the previous solution shifted by one stage; reported beside the cold-start `value`"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, NT, x0, ub, stuck, xref, u0_gpu, line["dtype"] == "f32")
        print(json.dumps(line))
    mpc.close()
    if world > 1:
        dist.destroy_process_group()


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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        run_multi_inprocess(args)       # no launcher: fan out inside this process, before anything here touches a GPU
    else:
        run_rank(args)


if __name__ == "__main__":
    main()
