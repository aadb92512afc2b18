"""ctypes binding of include/ftmpc.h (libftmpc_hip.so, built in-tree by csrc/Makefile).

The product path has no CPU fallback: `load_library()` raises if the shared object is
missing, and `ftmpc_create` fails when no gfx950 device is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_CSRC = _HERE.parent / "csrc"
# FTMPC_LIB selects another build of the SAME library (csrc/Makefile targets `stamps`, `plain`, experiments): diagnostics only
_SO = Path(os.environ["FTMPC_LIB"]).resolve() if os.environ.get("FTMPC_LIB") else _HERE / "libftmpc_hip.so"

MAX_NT = 16
MAX_TERM_ROWS = 80
MAX_HULL_ROWS = 128
MAX_TCOST = 24
KERNEL_SLOTS = 7
KERNEL_AUTO, KERNEL_DENSE, KERNEL_WORKGROUP = 0, 1, 2

# every symbol include/ftmpc.h declares (tests check the list against the header)
SYMBOLS = (
    "ftmpc_default_config", "ftmpc_create", "ftmpc_destroy", "ftmpc_last_error", "ftmpc_reserve",
    "ftmpc_solve_batch", "ftmpc_solve_batch_device", "ftmpc_solve_sqp_batch", "ftmpc_sqp_graph_launches", "ftmpc_solve_wrench_batch", "ftmpc_eval_cost_batch", "ftmpc_simulate_batch", "ftmpc_simulate_batch_ex", "ftmpc_simulate_wrench_batch", "ftmpc_last_handed_over", "ftmpc_allocate_batch", "ftmpc_shift_warm", "ftmpc_set_profiling",
    "ftmpc_last_kernel_ms", "ftmpc_kernel_name", "ftmpc_routed_kernel_name", "ftmpc_multi_routed_kernel_name", "ftmpc_debug_build_qp", "ftmpc_version", "ftmpc_build_id",
    "ftmpc_multi_create", "ftmpc_multi_destroy", "ftmpc_multi_last_error", "ftmpc_multi_device_count", "ftmpc_multi_worker_cpus",
    "ftmpc_multi_shard_bounds", "ftmpc_multi_solve_batch", "ftmpc_multi_upload", "ftmpc_multi_step",
    "ftmpc_multi_download", "ftmpc_multi_set_profiling", "ftmpc_multi_last_kernel_ms",
)


class FtmpcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ftmpc error {code}: {msg}")
        self.code = code


class ftmpc_config(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("NT", C.c_int32), ("dtype", C.c_int32), ("max_iters", C.c_int32),
        ("device_id", C.c_int32), ("struct_size", C.c_int32),
        ("dt", C.c_double), ("mass", C.c_double), ("J", C.c_double * 9),
        ("D", C.c_double * (6 * MAX_NT)), ("Q", C.c_double * 9), ("R", C.c_double * 6),
        ("P", C.c_double * 81), ("r", C.c_double * 3), ("f_virt", C.c_double * 3),
        ("rho", C.c_double), ("mu_stop", C.c_double),
        ("terminal_set", C.c_int32), ("term_rows", C.c_int32),
        ("term_A", C.c_double * (MAX_TERM_ROWS * 9)), ("term_b", C.c_double * MAX_TERM_ROWS),
        ("terminal_cost_terms", C.c_int32), ("tc_npoly", C.c_int32), ("tc_nroot", C.c_int32), ("tc_reserved", C.c_int32),
        ("tc_poly_coef", C.c_double * MAX_TCOST), ("tc_poly_exp", C.c_int32 * (MAX_TCOST * 9)),
        ("tc_root_coef", C.c_double * MAX_TCOST), ("tc_root_eps", C.c_double * MAX_TCOST), ("tc_root_pow", C.c_double * MAX_TCOST),
        ("tc_root_exp", C.c_int32 * (MAX_TCOST * 9)), ("tc_const", C.c_double),
        ("kernel_select", C.c_int32), ("stage_chunks", C.c_int32), ("lin_split_max", C.c_int64),
        ("state_bounds", C.c_int32), ("sb_reserved", C.c_int32), ("xlb", C.c_double * 13), ("xub", C.c_double * 13),
    ]


def library_path() -> Path:
    return _SO


def build_library(force: bool = False) -> Path:
    """Compiles csrc/*.hip for gfx950 into ft_mpc_amd/libftmpc_hip.so (hipcc cross-compiles
    without a GPU).  Idempotent: make decides whether anything is stale."""
    if force and _SO.exists():
        _SO.unlink()
    subprocess.run(["make", "-C", str(_CSRC)], check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not _SO.exists():
        raise FileNotFoundError(
            f"{_SO} is missing: build it with `make -C {_CSRC}` (or __graft_entry__.build()). "
            "There is no CPU fallback for the MPC QP-step path.")
    lib = C.CDLL(str(_SO))
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p
    lib.ftmpc_default_config.argtypes = [C.POINTER(ftmpc_config), C.c_int32, C.c_int32]
    lib.ftmpc_create.argtypes = [C.POINTER(ftmpc_config), C.POINTER(vp)]
    lib.ftmpc_destroy.argtypes = [vp]
    lib.ftmpc_last_error.argtypes = [vp]
    lib.ftmpc_last_error.restype = C.c_char_p
    lib.ftmpc_reserve.argtypes = [vp, C.c_int64]
    lib.ftmpc_solve_batch.argtypes = [vp, C.c_int64, dp, dp, dp, dp, C.c_int64, dp, C.c_int64, dp, dp, dp, ip, ip]
    lib.ftmpc_solve_batch_device.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_int64, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    lib.ftmpc_solve_wrench_batch.argtypes = [vp, C.c_int64, dp, dp, dp, dp, C.c_int32, ip, dp, C.c_int32, dp, C.c_int64, dp, C.c_int64,
                                             dp, dp, dp, dp, ip, ip, ip]
    lib.ftmpc_solve_sqp_batch.argtypes = [vp, C.c_int64, dp, dp, dp, dp, C.c_int64, dp, C.c_int64, dp, C.c_int32, C.c_int32, C.c_double,
                                          dp, dp, dp, dp, ip, ip, ip]
    lib.ftmpc_eval_cost_batch.argtypes = [vp, C.c_int64, dp, dp, dp, dp, C.c_int64, dp, C.c_int64, dp, dp]
    lib.ftmpc_simulate_batch.argtypes = [vp, C.c_int64, C.c_int32, dp, dp, dp, dp, dp, dp, C.c_uint64, dp, ip]
    lib.ftmpc_simulate_batch_ex.argtypes = [vp, C.c_int64, C.c_int32, dp, dp, dp, dp, dp, dp, C.c_uint64, C.c_int32, C.c_int32, C.c_double, dp, ip]
    lib.ftmpc_simulate_wrench_batch.argtypes = [vp, C.c_int64, C.c_int32, dp, dp, dp, dp, C.c_int32, ip, dp, C.c_int32, dp, dp, dp, C.c_uint64, dp, ip, ip]
    lib.ftmpc_last_handed_over.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.ftmpc_sqp_graph_launches.argtypes = [vp]
    lib.ftmpc_sqp_graph_launches.restype = C.c_int64
    lib.ftmpc_allocate_batch.argtypes = [vp, C.c_int64, dp, dp, dp, ip, ip]
    lib.ftmpc_shift_warm.argtypes = [C.c_int64, C.c_int32, C.c_int32, dp]
    lib.ftmpc_set_profiling.argtypes = [vp, C.c_int32]
    lib.ftmpc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_int32]
    lib.ftmpc_kernel_name.argtypes = [C.c_int32]
    lib.ftmpc_kernel_name.restype = C.c_char_p
    lib.ftmpc_routed_kernel_name.argtypes = [vp, C.c_int32]
    lib.ftmpc_routed_kernel_name.restype = C.c_char_p
    lib.ftmpc_multi_routed_kernel_name.argtypes = [vp, C.c_int32]
    lib.ftmpc_multi_routed_kernel_name.restype = C.c_char_p
    lib.ftmpc_debug_build_qp.argtypes = [vp, C.c_int64, dp, dp, dp, dp, C.c_int64, dp, C.c_int64, dp, C.c_int64,
                                         dp, C.c_int64, dp, dp, dp, ip]
    lib.ftmpc_version.restype = C.c_int32
    lib.ftmpc_build_id.restype = C.c_char_p
    lib.ftmpc_multi_create.argtypes = [C.POINTER(ftmpc_config), ip, C.c_int32, C.POINTER(vp)]
    lib.ftmpc_multi_destroy.argtypes = [vp]
    lib.ftmpc_multi_last_error.argtypes = [vp]
    lib.ftmpc_multi_last_error.restype = C.c_char_p
    lib.ftmpc_multi_device_count.argtypes = [vp]
    lib.ftmpc_multi_device_count.restype = C.c_int32
    lib.ftmpc_multi_shard_bounds.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.ftmpc_multi_solve_batch.argtypes = [vp, C.c_int64, dp, dp, dp, dp, C.c_int64, dp, C.c_int64, dp, dp, dp, ip, ip]
    lib.ftmpc_multi_upload.argtypes = [vp, C.c_int64, dp, dp, dp, dp, C.c_int64, dp, C.c_int64, dp]
    lib.ftmpc_multi_step.argtypes = [vp, C.c_int32, C.c_int32]
    lib.ftmpc_multi_download.argtypes = [vp, dp, dp, ip, ip]
    lib.ftmpc_multi_set_profiling.argtypes = [vp, C.c_int32]
    lib.ftmpc_multi_last_kernel_ms.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), C.c_int32]
    lib.ftmpc_multi_worker_cpus.argtypes = [vp, C.c_int32]
    lib.ftmpc_multi_worker_cpus.restype = C.c_int32
    for name in SYMBOLS:
        if name not in ("ftmpc_last_error", "ftmpc_kernel_name", "ftmpc_routed_kernel_name", "ftmpc_multi_routed_kernel_name", "ftmpc_version", "ftmpc_build_id", "ftmpc_multi_last_error",
                        "ftmpc_multi_device_count", "ftmpc_multi_worker_cpus", "ftmpc_sqp_graph_launches"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib
