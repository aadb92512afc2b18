"""oracle/batch.py -- the NumPy oracle over a batch, on several host cores.

TEST INFRASTRUCTURE ONLY (see oracle/refmath.py header): lets the GPU parity tests hold kernel 11 / the float64 kernel's
general-constraint modes against oracle/qp_oracle.py instance by instance on thousands of instances (36 ms each on one core).
Workers are SPAWNED (never forked: the parent holds a GPU context) and import numpy and the oracle (sqp_batch also unpickles the
terminal-ingredients object it is given)."""
from __future__ import annotations

import os

import numpy as np


def _wrench_chunk(args):
    from oracle import qp_oracle as qo
    kw, x0, ub, stuck, xref, uref, warmG, term, iters, lo = args
    kw = dict(kw)
    mu_polish = kw.pop("mu_polish", None)
    cfg = qo.QPConfig(**kw)
    n = x0.shape[0]
    G = np.full((n, cfg.N, 6), np.nan)
    st = np.zeros(n, np.int32)
    nit = np.zeros(n, np.int32)
    act = np.zeros((n, 2), np.int32)
    with np.errstate(all="ignore"):
        for b in range(n):
            try:
                _, T, st[b], nit[b], qp = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref, uref=uref, warmG=None if warmG is None else warmG[b],
                                                                   term_set=term, iters=iters, mu_polish=mu_polish)
            except ValueError:       # flat hull: the healthy thrusters do not span R^6
                st[b] = 3
                continue
            G[b] = T
            nh = qp["nhull"]
            act[b] = (qp["z"][:nh] > qp["s"][:nh]).sum(), (qp["z"][nh:] > qp["s"][nh:]).sum()
    return lo, G, st, nit, act


def solve_wrench_batch(N, NT, x0, ub, stuck, xref, uref=None, warmG=None, term_set=None, iters=60, workers=None, mu_polish=None):
    """oracle/qp_oracle.py:solve_wrench_instance for every instance of a batch.  Returns dict(G [B,N,6], status [B] (3: flat hull),
    iters [B], active [B,2] = rows with z > s among the hull / terminal rows)."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    B = x0.shape[0]
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
        workers = max(1, min(workers, 16, B // 8 or 1))
    kw = dict(N=N, NT=NT, mu_polish=mu_polish)      # (mu_polish: see qp_oracle.ipm_general)
    step = max(1, (B + 4 * workers - 1) // (4 * workers))
    jobs = [(kw, x0[lo:lo + step], ub[lo:lo + step], stuck[lo:lo + step], xref, uref, None if warmG is None else warmG[lo:lo + step], term_set, iters, lo)
            for lo in range(0, B, step)]
    out = dict(G=np.zeros((B, N, 6)), status=np.zeros(B, np.int32), iters=np.zeros(B, np.int32), active=np.zeros((B, 2), np.int32))
    if workers == 1:
        results = map(_wrench_chunk, jobs)
    else:
        saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
        for k in saved:
            os.environ[k] = "1"      # (the workers inherit the environment: one BLAS thread each)
        try:
            pool = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"))
            results = list(pool.map(_wrench_chunk, jobs))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    for lo, G, st, nit, act in results:
        n = G.shape[0]
        out["G"][lo:lo + n], out["status"][lo:lo + n], out["iters"][lo:lo + n], out["active"][lo:lo + n] = G, st, nit, act
    if workers != 1:
        pool.shutdown()
    return out


def _sqp_chunk(args):
    from oracle import qp_oracle as qo
    kw, x0, ub, stuck, xref, terminal, sqp_iters, lo = args
    cfg = qo.QPConfig(**kw)
    n = x0.shape[0]
    U = np.zeros((n, cfg.N, cfg.NT))
    J = np.zeros(n)
    J0 = np.zeros(n)
    for b in range(n):
        U[b], J[b], hist = qo.sqp_linesearch(cfg, x0[b], ub[b], stuck[b], xref, terminal=terminal, sqp_iters=sqp_iters)
        J0[b] = hist[0]
    return lo, U, J, J0


def sqp_batch(N, NT, x0, ub, stuck, xref, terminal=None, sqp_iters=10, workers=None):
    """oracle/qp_oracle.py:sqp_linesearch (2 s per instance at N = 20) for every instance of a batch, on spawned workers.
    `terminal` must pickle (the host mirror's TerminalIngredients does).  Returns dict(U [B,N,NT], cost [B], cost0 [B])."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    B = x0.shape[0]
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
        workers = max(1, min(workers, 16, B))
    step = max(1, (B + workers - 1) // workers)
    jobs = [(dict(N=N, NT=NT), x0[lo:lo + step], ub[lo:lo + step], stuck[lo:lo + step], xref, terminal, sqp_iters, lo) for lo in range(0, B, step)]
    out = dict(U=np.zeros((B, N, NT)), cost=np.zeros(B), cost0=np.zeros(B))
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"
    try:
        with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as pool:
            results = list(pool.map(_sqp_chunk, jobs))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for lo, U, J, J0 in results:
        n = U.shape[0]
        out["U"][lo:lo + n], out["cost"][lo:lo + n], out["cost0"][lo:lo + n] = U, J, J0
    return out
