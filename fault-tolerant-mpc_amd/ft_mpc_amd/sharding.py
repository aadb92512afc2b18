"""Batch sharding across the GPUs of one node: instances are independent, so
the batch axis is split contiguously, one process per GPU, NO data-path collective; results are
gathered on the host of rank 0 (torch.distributed is used for that host gather only --
gloo on CPU tensors, so it works with or without RCCL)."""
from __future__ import annotations

import numpy as np


def shard_bounds(B: int, world: int, rank: int):
    """Contiguous split: rank r owns [lo, hi).  Sizes differ by at most one."""
    lo = (B * rank) // world
    hi = (B * (rank + 1)) // world
    return lo, hi


def solve_sharded(solve_fn, x0, ub, stuck, *, rank=0, world=1, dist=None, **kw):
    """Every rank passes the FULL batch arrays (or at least its own slice filled in);
    `solve_fn(x0, ub, stuck, **kw) -> dict(u0, status, iters)` is called on this rank's slice
    (per-instance entries of kw -- `warmU`, per-instance `xref`/`uref` -- are sliced alike).
    Returns the gathered dict on rank 0 and None elsewhere."""
    B = np.asarray(x0).reshape(-1, 13).shape[0]
    lo, hi = shard_bounds(B, world, rank)
    loc_kw = {}
    for k, v in kw.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B and k in ("warmU", "xref_batch", "uref_batch"):
            loc_kw[k] = v[lo:hi]
        else:
            loc_kw[k] = v
    out = solve_fn(np.asarray(x0).reshape(-1, 13)[lo:hi], np.asarray(ub)[lo:hi], np.asarray(stuck)[lo:hi], **loc_kw)
    if world == 1 or dist is None:
        return out
    keys = [k for k in ("u0", "U", "status", "iters") if out.get(k) is not None]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object({k: out[k] for k in keys}, gathered, dst=0)
    if rank != 0:
        return None
    return {k: np.concatenate([g[k] for g in gathered], axis=0) for k in keys}


class MultiGPUMPC:
    """The batch axis across the GPUs of one node from ONE process (include/ftmpc.h, ftmpc_multi_*): one host
    thread + one handle + one stream set per device inside the library, contiguous shards, no collective,
    outputs gathered by every device writing its slice of the caller's arrays.
    `devices`: None (every visible GPU), an int (devices 0..n-1) or a list of ordinals (an ordinal may repeat:
    several handles on one GPU)."""

    def __init__(self, cfg=None, devices=None, **kw):
        import ctypes as C
        from . import _lib
        from .batch import BatchedMPC, MPCConfig
        self.cfg = cfg or MPCConfig(**kw)
        self.lib = _lib.load_library()
        self._C = C
        c = BatchedMPC.make_c_config(self.lib, self.cfg)
        if devices is None:
            ids, n = None, 0
        elif isinstance(devices, int):
            ids, n = None, int(devices)
        else:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            ids, n = arr, len(devices)
        self._h = C.c_void_p()
        rc = self.lib.ftmpc_multi_create(C.byref(c), ids, n, C.byref(self._h))
        if rc != 0:
            raise _lib.FtmpcError(rc, self.lib.ftmpc_multi_last_error(None).decode())
        self.n_devices = int(self.lib.ftmpc_multi_device_count(self._h))
        self._B = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ftmpc_multi_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            from . import _lib
            raise _lib.FtmpcError(rc, self.lib.ftmpc_multi_last_error(self._h).decode())

    def shard_bounds(self, B, slot):
        C = self._C
        lo, hi = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.ftmpc_multi_shard_bounds(self._h, int(B), int(slot), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def _prep(self, x0, ub, stuck, xref, uref, warmU):
        from .batch import _f64
        N, NT = self.cfg.N, self.cfg.NT
        x0 = _f64(x0).reshape(-1, 13)
        B = x0.shape[0]
        ub = _f64(ub, (B, NT))
        stuck = _f64(stuck, (B, NT))
        xref = _f64(xref)
        xs = 0 if xref.size == 9 * (N + 1) else xref.size // B
        us = 0
        if uref is not None:
            uref = _f64(uref)
            us = 0 if uref.size == 6 * (N + 1) else uref.size // B
        if warmU is not None and not (isinstance(warmU, np.ndarray) and warmU.dtype == np.float64
                                      and warmU.flags.c_contiguous and warmU.size == B * N * NT):
            raise ValueError("warmU must be a C-contiguous float64 array of B*N*NT")
        return B, x0, ub, stuck, xref, xs, uref, us

    def solve(self, x0, ub, stuck, xref, uref=None, warmU=None, return_U=False):
        """Same contract as BatchedMPC.solve (host arrays in, host arrays out), sharded over the devices."""
        from .batch import _ptr
        C = self._C
        N, NT = self.cfg.N, self.cfg.NT
        B, x0, ub, stuck, xref, xs, uref, us = self._prep(x0, ub, stuck, xref, uref, warmU)
        u0 = np.empty((B, NT))
        U = np.empty((B, N, NT)) if return_U else None
        status = np.empty(B, np.int32)
        iters = np.empty(B, np.int32)
        self._check(self.lib.ftmpc_multi_solve_batch(self._h, B, _ptr(x0), _ptr(ub), _ptr(stuck), _ptr(xref), xs, _ptr(uref), us,
                                                     _ptr(warmU), _ptr(u0), _ptr(U), _ptr(status, C.c_int32), _ptr(iters, C.c_int32)))
        return dict(u0=u0, U=U, status=status, iters=iters)

    # -- shards resident in HBM between steps ---------------------------------------------------
    def upload(self, x0, ub, stuck, xref, uref=None, warmU=None):
        from .batch import _ptr
        B, x0, ub, stuck, xref, xs, uref, us = self._prep(x0, ub, stuck, xref, uref, warmU)
        self._check(self.lib.ftmpc_multi_upload(self._h, B, _ptr(x0), _ptr(ub), _ptr(stuck), _ptr(xref), xs, _ptr(uref), us, _ptr(warmU)))
        self._B = B

    def step(self, steps=1, keep_U=False):
        """`steps` MPC steps over the resident shards on every device at once; returns when all devices are idle."""
        self._check(self.lib.ftmpc_multi_step(self._h, int(steps), 1 if keep_U else 0))

    def download(self, return_U=False):
        from .batch import _ptr
        C = self._C
        N, NT, B = self.cfg.N, self.cfg.NT, self._B
        u0 = np.empty((B, NT))
        U = np.empty((B, N, NT)) if return_U else None
        status = np.empty(B, np.int32)
        iters = np.empty(B, np.int32)
        self._check(self.lib.ftmpc_multi_download(self._h, _ptr(u0), _ptr(U), _ptr(status, C.c_int32), _ptr(iters, C.c_int32)))
        return dict(u0=u0, U=U, status=status, iters=iters)

    def set_profiling(self, on):
        self._check(self.lib.ftmpc_multi_set_profiling(self._h, 1 if on else 0))

    def last_kernel_ms(self, slot=0):
        C = self._C
        from . import _lib
        ms = (C.c_float * _lib.KERNEL_SLOTS)()
        self._check(self.lib.ftmpc_multi_last_kernel_ms(self._h, int(slot), ms, _lib.KERNEL_SLOTS))
        return {self.lib.ftmpc_multi_routed_kernel_name(self._h, k).decode(): float(ms[k]) for k in range(_lib.KERNEL_SLOTS) if ms[k] > 0}

    def worker_cpus(self, slot=0):
        """Host cores the worker thread of device slot `slot` is bound to (0: affinity left alone)."""
        return int(self.lib.ftmpc_multi_worker_cpus(self._h, int(slot)))


def solve_multi_gpu(cfg, x0, ub, stuck, xref, devices=None, **kw):
    """One call: shard the batch over `devices` (default: every visible GPU), solve, gather on the host."""
    m = MultiGPUMPC(cfg, devices=devices)
    try:
        return m.solve(x0, ub, stuck, xref, **kw)
    finally:
        m.close()
