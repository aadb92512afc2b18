"""Diagnostic: per-phase cycle shares of kernel 11 (ftmpc_solve_hull32_kernel; stamps build, see csrc/Makefile) and device-side timing of the two-stage step."""
import sys, ctypes as C, os, time
sys.path.insert(0, '/root/repo/fault-tolerant-mpc_amd'); sys.path.insert(0, '/root/repo')
import numpy as np
from pathlib import Path
from ft_mpc_amd import _lib
_lib._SO = Path(os.environ["FTMPC_LIB"]) if os.environ.get("FTMPC_LIB") else _lib._HERE / "libftmpc_hip_stamps.so"
import ft_mpc_amd
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
from oracle import qp_oracle as qo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 15
NT = 16
x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 7900)
hull = hull_tables(qo.QPConfig(N=N, NT=NT).D, ub, stuck)
mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype="f32", max_iters=40)
out = mpc.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), hull=hull)
ok = out["status"] == 0
cnt = min(ok.sum(), 4096)
buf = np.zeros((4096, 12), np.uint64)
f = mpc.lib.ftmpc_debug_read_stamps; f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
assert f(mpc._h, 4096, buf.ctypes.data_as(C.c_void_p)) == 0
# (degenerate-hull instances are not sent to the kernel: the first ok.sum() slots are the solved ones)
m = buf[:cnt].astype(np.float64).mean(axis=0); tot = m.sum(); it = out["iters"][ok].mean()
names = ["prologue", "build: propagate", "build: mfma", "start gradient", "row setup", "factorisation (seeds + float64 Cholesky)", "", "mu + weights + G blocks (float64)",
         "refine (f64 grad)", "(factor -> sweeps)", "sweeps (2, float64) + rows + elementwise + H dd", "output"]
print("iters mean %.2f   total cycles/QP %.0f" % (it, tot))
for n_, v in zip(names, m):
    if n_: print("  %-52s %10.0f  %5.1f%%   per-iter %8.0f" % (n_, v, 100 * v / tot, v / it))
