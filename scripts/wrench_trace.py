"""Diagnostic: one two-stage step (ftmpc_solve_wrench_batch) repeated a few times, for rocprofv3 --kernel-trace --stats."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd')
import numpy as np
import ft_mpc_amd
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
from oracle import qp_oracle as qo
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 15
tset = len(sys.argv) > 3 and sys.argv[3] == "tset"
NT, B = 16, 16384
x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 7900)
hull = hull_tables(qo.QPConfig(N=N, NT=NT).D, ub, stuck)
m = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype=dt, max_iters=40, terminal_set=True if tset else None)
for _ in range(5):
    out = m.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), hull=hull)
print(np.bincount(out["status"], minlength=4), out["iters"].mean())
