"""ctypes wrapper of oracle/ftmpc_oracle.c (TEST INFRASTRUCTURE ONLY, see refmath.py header)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from . import refmath as rm

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "libftmpc_oracle.so"
MAX_NT = 16


class oracle_config(C.Structure):
    _fields_ = [("N", C.c_int32), ("NT", C.c_int32), ("max_iters", C.c_int32), ("flags", C.c_int32),
                ("dt", C.c_double), ("mass", C.c_double), ("J", C.c_double * 9),
                ("D", C.c_double * (6 * MAX_NT)), ("Q", C.c_double * 9), ("R", C.c_double * 6),
                ("P", C.c_double * 81), ("r", C.c_double * 3), ("f_virt", C.c_double * 3),
                ("rho", C.c_double), ("mu_stop", C.c_double)]


def build():
    subprocess.run(["make", "-C", str(_HERE)], check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _SO.exists():
            build()
        _lib = C.CDLL(str(_SO))
    return _lib


def make_config(qcfg, max_iters=40, mu_stop=1e-13, polish=True):
    """qcfg: oracle.qp_oracle.QPConfig;  polish=False: the interior-point iteration to mu_stop alone (flags bit 0)"""
    c = oracle_config()
    c.N, c.NT, c.max_iters = qcfg.N, qcfg.NT, max_iters
    c.flags = 0 if polish else 1
    c.dt, c.mass, c.rho, c.mu_stop = qcfg.dt, qcfg.mass, qcfg.rho, mu_stop
    c.J[:] = list(np.asarray(qcfg.J, float).reshape(9))
    flat = np.zeros(6 * MAX_NT)
    flat[:6 * qcfg.NT] = np.asarray(qcfg.D, float).reshape(-1)
    c.D[:] = list(flat)
    c.Q[:] = list(qcfg.Q)
    c.R[:] = list(qcfg.R)
    c.P[:] = list(np.asarray(qcfg.P, float).reshape(81))
    c.r[:] = list(qcfg.r)
    c.f_virt[:] = list(qcfg.f_virt)
    return c


def _p(a, t=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def solve_batch(qcfg, x0, ub, stuck, xref, uref=None, warmU=None, max_iters=40, mu_stop=1e-13, nthreads=1,
                return_U=True, polish=True):
    """Exact solution of the QP-spec for a batch: float64 interior-point iteration, left at mu 1e-7 for the active-set polish
    (ftmpc_oracle.c:polish_box -- the solution on the verified active set; where it does not verify, the iteration to mu_stop).
    xref: 9x(N+1) array (shared) or [B, 9(N+1)] column-major-flattened per instance."""
    N, NT = qcfg.N, qcfg.NT
    x0 = np.ascontiguousarray(x0, float).reshape(-1, 13)
    B = x0.shape[0]
    ub = np.ascontiguousarray(ub, float).reshape(B, NT)
    stuck = np.ascontiguousarray(stuck, float).reshape(B, NT)
    xr = np.asarray(xref, float)
    if xr.shape == (9, N + 1):
        xr = np.ascontiguousarray(xr.reshape(-1, order="F"))
        xs = 0
    else:
        xr = np.ascontiguousarray(xr).reshape(B, -1)
        xs = xr.shape[1]
    ur, us = None, 0
    if uref is not None:
        ur = np.asarray(uref, float)
        if ur.shape == (6, N + 1):
            ur = np.ascontiguousarray(ur.reshape(-1, order="F"))
        else:
            ur = np.ascontiguousarray(ur).reshape(B, -1)
            us = ur.shape[1]
    warm = None if warmU is None else np.ascontiguousarray(warmU, float).reshape(B, N * NT)
    u0 = np.zeros((B, NT))
    U = np.zeros((B, N, NT)) if return_U else None
    status = np.zeros(B, np.int32)
    iters = np.zeros(B, np.int32)
    c = make_config(qcfg, max_iters, mu_stop, polish)
    f = lib().ftmpc_oracle_solve_batch
    f.restype = C.c_int
    f.argtypes = [C.POINTER(oracle_config), C.c_int64] + [C.POINTER(C.c_double)] * 4 + [C.c_int64, C.POINTER(C.c_double),
                  C.c_int64] + [C.POINTER(C.c_double)] * 3 + [C.POINTER(C.c_int32)] * 2 + [C.c_int32]
    rc = f(C.byref(c), B, _p(x0), _p(ub), _p(stuck), _p(xr), xs, _p(ur), us, _p(warm), _p(u0), _p(U),
           _p(status, C.c_int32), _p(iters, C.c_int32), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"ftmpc_oracle_solve_batch rc={rc}")
    return dict(u0=u0, U=U, status=status, iters=iters)


def build_qp(qcfg, x0, ub, stuck, xref, uref=None, warmU=None):
    N, NT = qcfg.N, qcfg.NT
    nm = N * NT
    H = np.zeros(nm * nm)
    g = np.zeros(nm)
    lo = np.zeros(nm)
    hi = np.zeros(nm)
    xr = np.ascontiguousarray(np.asarray(xref, float).reshape(9, N + 1).reshape(-1, order="F"))
    ur = None if uref is None else np.ascontiguousarray(np.asarray(uref, float).reshape(6, N + 1).reshape(-1, order="F"))
    warm = None if warmU is None else np.ascontiguousarray(warmU, float).reshape(-1)
    c = make_config(qcfg)
    f = lib().ftmpc_oracle_build_qp
    f.restype = C.c_int
    f.argtypes = [C.POINTER(oracle_config)] + [C.POINTER(C.c_double)] * 10
    n = f(C.byref(c), _p(np.ascontiguousarray(x0, float)), _p(np.ascontiguousarray(ub, float)),
          _p(np.ascontiguousarray(stuck, float)), _p(xr), _p(ur), _p(warm), _p(H), _p(g), _p(lo), _p(hi))
    return H[:n * n].reshape(n, n).copy(), g[:n].copy(), lo[:n].copy(), hi[:n].copy()


def plant_step(qcfg, x, u, ub, stuck):
    c = make_config(qcfg)
    xn = np.zeros(13)
    f = lib().ftmpc_oracle_plant_step
    f.restype = C.c_int
    f.argtypes = [C.POINTER(oracle_config)] + [C.POINTER(C.c_double)] * 5
    a = lambda v: np.ascontiguousarray(v, float)
    xx, uu, bb, ss = a(x), a(u), a(ub), a(stuck)
    f(C.byref(c), _p(xx), _p(uu), _p(bb), _p(ss), _p(xn))
    return xn


def solve_batch_complete(qcfg, x0, ub, stuck, xref, uref=None, warmU=None, nthreads=1, max_iters=60, mu_stop=1e-13, polish=True):
    """A reference for EVERY instance of the batch (the parity tests must not drop the ones this port's interior-point
    iteration does not finish in `max_iters`): stragglers get 400 iterations, and whatever is still not converged is
    solved by the independent exact solver (BVLS on the same condensed QP, oracle/qp_oracle.py:solve_exact).
    Returns the dict of solve_batch plus `how` [B]: 0 = first pass, 1 = long pass, 2 = BVLS."""
    from . import qp_oracle as qo
    x0 = np.ascontiguousarray(x0, float).reshape(-1, 13)
    B = x0.shape[0]
    N, NT = qcfg.N, qcfg.NT
    ub = np.ascontiguousarray(ub, float).reshape(B, NT)
    stuck = np.ascontiguousarray(stuck, float).reshape(B, NT)
    W = None if warmU is None else np.ascontiguousarray(warmU, float).reshape(B, N, NT)
    ref = solve_batch(qcfg, x0, ub, stuck, xref, uref=uref, warmU=None if W is None else W.copy(), max_iters=max_iters,
                      mu_stop=mu_stop, nthreads=nthreads, polish=polish)
    how = np.zeros(B, np.int32)
    xr = np.asarray(xref, float)
    per_x = xr.shape != (9, N + 1)
    ur = None if uref is None else np.asarray(uref, float)
    per_u = ur is not None and ur.shape != (6, N + 1)
    todo = np.flatnonzero(ref["status"] != 0)
    if todo.size:
        sub = solve_batch(qcfg, x0[todo], ub[todo], stuck[todo], xr.reshape(B, -1)[todo] if per_x else xr,
                          uref=None if ur is None else (ur.reshape(B, -1)[todo] if per_u else ur),
                          warmU=None if W is None else W[todo].copy(), max_iters=400, mu_stop=mu_stop, nthreads=nthreads, polish=polish)
        for k in ("u0", "U", "status", "iters"):
            ref[k][todo] = sub[k]
        how[todo] = 1
    for b in np.flatnonzero(ref["status"] != 0):
        xb = xr.reshape(B, -1)[b].reshape(9, N + 1, order="F") if per_x else xr
        ub_ = None if ur is None else (ur.reshape(B, -1)[b].reshape(6, N + 1, order="F") if per_u else ur)
        u0, U, _ = qo.solve_instance(qcfg, x0[b], ub[b], stuck[b], xb, uref=ub_, warmU=None if W is None else W[b], exact=True)
        ref["u0"][b], ref["U"][b], ref["status"][b] = u0, U, 0
        how[b] = 2
    ref["how"] = how
    return ref


def exact_where_apart(qcfg, ref, U_test, x0, ub, stuck, xref, uref=None, warmU=None, tol=1e-7, tol_u0=None, cap=64, f_max=None):
    """The port's interior-point iterate at mu 1e-13 is not the exact solution where a bound is weakly active (s z = mu with s ~ z:
    up to ~1.5e-6 f_max measured on the degenerate batches), while the kernels that finish by the active-set polish return the
    solution on the verified active set, exact to ~1e-12.  For the instances where the result under test and `ref` are more than
    `tol` apart (`tol_u0` on the first stage, when given) this replaces ref["u0"], ref["U"] by the INDEPENDENT exact solution (BVLS, oracle/qp_oracle.py:solve_exact) --
    the tests then hold the result under test to `tol` against an exact reference.  Returns the indices replaced (at most `cap`:
    more than that is a failure of the solver under test, not of the reference)."""
    from . import qp_oracle as qo
    from . import refmath as rm
    f_max = rm.F_MAX if f_max is None else f_max
    x0 = np.ascontiguousarray(x0, float).reshape(-1, 13)
    B = x0.shape[0]
    N, NT = qcfg.N, qcfg.NT
    ub = np.ascontiguousarray(ub, float).reshape(B, NT)
    stuck = np.ascontiguousarray(stuck, float).reshape(B, NT)
    W = None if warmU is None else np.ascontiguousarray(warmU, float).reshape(B, N, NT)
    xr = np.asarray(xref, float)
    per_x = xr.shape != (9, N + 1)
    ur = None if uref is None else np.asarray(uref, float)
    per_u = ur is not None and ur.shape != (6, N + 1)
    err = np.abs(np.asarray(U_test).reshape(B, -1) - ref["U"].reshape(B, -1)).max(axis=1) / f_max
    apart = err > tol
    if tol_u0 is not None:
        apart |= np.abs(np.asarray(U_test).reshape(B, N, NT)[:, 0] - ref["u0"]).max(axis=1) / f_max > tol_u0
    idx = np.flatnonzero(apart & (ref["status"] == 0))
    assert idx.size <= cap, (idx.size, float(err.max()))
    for b in idx:
        xb = xr.reshape(B, -1)[b].reshape(9, N + 1, order="F") if per_x else xr
        ub_ = None if ur is None else (ur.reshape(B, -1)[b].reshape(6, N + 1, order="F") if per_u else ur)
        u0, U, _ = qo.solve_instance(qcfg, x0[b], ub[b], stuck[b], xb, uref=ub_, warmU=None if W is None else W[b], exact=True)
        ref["u0"][b], ref["U"][b] = u0, U
    return idx

