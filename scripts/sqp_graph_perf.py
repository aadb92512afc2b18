"""Diagnostic: ftmpc_solve_sqp_batch, direct launches against the replayed hipGraph (FTMPC_SQP_GRAPH=0 / 1), per batch size."""
import os, sys, time, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd')
    import numpy as np
    import ft_mpc_amd
    from oracle import qp_oracle as qo
    N, NT = 20, 8
    for B in (64, 256, 1024, 4096, 16384):
        mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT)
        x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 99)
        xr = xref.reshape(-1, order="F")
        for _ in range(3):
            mpc.solve_sqp_device(x0, ub, stuck, xr, sqp_iters=10)
        t = time.perf_counter()
        reps = 10
        for _ in range(reps):
            mpc.solve_sqp_device(x0, ub, stuck, xr, sqp_iters=10)
        dt = (time.perf_counter() - t) / reps
        print("B %6d  %8.3f ms per call  %9.0f NLP solves/s  graph replays %d" % (B, 1e3 * dt, B / dt, mpc.sqp_graph_launches()), flush=True)
else:
    for g in ("0", "1"):
        print("FTMPC_SQP_GRAPH=" + g, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, FTMPC_SQP_GRAPH=g), check=True)
