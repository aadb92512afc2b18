"""Batched front-end of the HIP MPC QP-step path (numpy in / numpy out, or device pointers).

`BatchedMPC.solve` is the batched form of SpiralingController.get_control
(reference: ft_mpc/controllers/spiraling_mpc.py:288-317): B robot states + fault scenarios in,
B thruster command vectors out.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .models.sys_model import allocation_matrix_16, allocation_matrix_8

F_MAX = 3.4  # sys_model.py:60


@dataclass
class MPCConfig:
    """Problem constants; defaults are the reference's (see include/ftmpc.h ftmpc_config)."""
    N: int = 20
    NT: int = 8
    dt: float = 0.1
    mass: float = 16.8
    J: np.ndarray = field(default_factory=lambda: np.diag([0.2, 0.3, 0.25]))
    D: np.ndarray = None
    Q: np.ndarray = field(default_factory=lambda: np.array([1, 1, 1, 1, 1, 1, 2, 2, 2], float))
    R: np.ndarray = field(default_factory=lambda: np.array([0.1, 0.1, 0.1, 0.01, 0.01, 0.01]))
    P: np.ndarray = None
    r: np.ndarray = None
    f_virt: np.ndarray = field(default_factory=lambda: np.array([0.0, 3.5, 0.0]))
    rho: float = 0.05
    max_iters: int = 0            # <= 0: library default (30)
    mu_stop: float = 0.0          # <= 0: library default (1e-11 for f32, 1e-13 for f64)
    device_id: int = 0
    dtype: str = "f32"            # arithmetic of the KKT/IPM solve: "f32" | "f64" (N*NT > 240 always runs f64)
    # terminal set  term_A (c_N[0:9] - xref_N) <= term_b  (config/terminal.yaml term_set; spiraling_mpc.py:199-202):
    # a TerminalSet / (A, b) pair, or True for the shipped config/terminal.yaml.  Needs dtype "f64".
    terminal_set: object = None
    # non-quadratic part of the terminal cost (config/terminal.yaml `cost` beyond e'P e; spiraling_mpc.py:196): a
    # TerminalIngredients object, or True for the shipped config/terminal.yaml.  Every QP then carries its exact
    # gradient at the linearisation point and eval_cost includes it (see BatchedMPC.solve_sqp).
    terminal_cost: object = None
    # implementation switches (include/ftmpc.h ftmpc_config; diagnostics and A/B runs): "auto" | "dense" (Newton systems
    # always in the thruster variables, never through the wrench-space form) | "workgroup" (the wrench-space form on a
    # workgroup per instance, kernel 8, where "auto" gives the instance one wave, kernel 10), the batch size up to which the linearisation
    # is split by tangent direction (0: library default, < 0: never) and the staging ranges of the host-buffer entry
    kernel_select: str = "auto"
    lin_split_max: int = 0
    stage_chunks: int = 0
    # state bounds  xlb <= c_k <= xub  on the 13 orbit-centre states [p, v, omega, q] of the stages 1 .. N-1: the reference's optional
    # controller params "xub" / "xlb" (spiraling_mpc.py:129-130,179-185; None = no bounds, +-inf = no row for that component)
    xlb: np.ndarray = None
    xub: np.ndarray = None


def _ptr(a, ct=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class BatchedMPC:
    """One handle == one GPU.  Not re-entrant (like the reference controller, which keeps its
    warm start in `self.optimal_solution`); use one instance per host thread / GPU."""

    @staticmethod
    def make_c_config(lib, cfg):
        """MPCConfig -> the C struct of include/ftmpc.h (defaults from ftmpc_default_config)."""
        c = _lib.ftmpc_config()
        rc = lib.ftmpc_default_config(C.byref(c), cfg.N, cfg.NT)
        if rc != 0:
            raise _lib.FtmpcError(rc, "bad N/NT")
        c.dt, c.mass, c.rho, c.mu_stop = cfg.dt, cfg.mass, cfg.rho, cfg.mu_stop
        c.max_iters, c.device_id = cfg.max_iters, cfg.device_id
        if cfg.dtype not in ("f32", "f64"):
            raise ValueError("dtype must be 'f32' or 'f64'")
        c.dtype = 1 if cfg.dtype == "f64" else 0
        if cfg.kernel_select not in ("auto", "dense", "workgroup"):
            raise ValueError("kernel_select must be 'auto', 'dense' or 'workgroup'")
        c.kernel_select = {"auto": _lib.KERNEL_AUTO, "dense": _lib.KERNEL_DENSE, "workgroup": _lib.KERNEL_WORKGROUP}[cfg.kernel_select]
        c.lin_split_max, c.stage_chunks = int(cfg.lin_split_max), int(cfg.stage_chunks)
        c.J[:] = list(_f64(cfg.J, 9))
        D = cfg.D
        if D is None:
            D = allocation_matrix_16() if cfg.NT == 16 else (allocation_matrix_8() if cfg.NT == 8 else None)
        if D is None:
            raise ValueError("MPCConfig.D (6 x NT) is required for NT not in (8, 16)")
        D = _f64(D, (6, cfg.NT))
        flat = np.zeros(6 * _lib.MAX_NT)
        flat[:6 * cfg.NT] = D.reshape(-1)
        c.D[:] = list(flat)
        c.Q[:] = list(_f64(cfg.Q, 9))
        c.R[:] = list(_f64(cfg.R, 6))
        if cfg.P is not None:
            c.P[:] = list(_f64(cfg.P, 81))
        if cfg.r is not None:
            c.r[:] = list(_f64(cfg.r, 3))
        c.f_virt[:] = list(_f64(cfg.f_virt, 3))
        if cfg.xlb is not None or cfg.xub is not None:      # (as the reference: one given, the other defaults to no bound)
            NOB = 1e300
            lo = np.full(13, -np.inf) if cfg.xlb is None else _f64(cfg.xlb, 13)
            hi = np.full(13, np.inf) if cfg.xub is None else _f64(cfg.xub, 13)
            c.state_bounds = 1
            c.xlb[:] = list(np.clip(lo, -NOB, NOB))
            c.xub[:] = list(np.clip(hi, -NOB, NOB))
        ts = cfg.terminal_set
        if ts is not None and ts is not False:
            if ts is True:
                from .controllers.tools.terminal_ingredients import load_terminal
                ts = load_terminal().term_set
            A, b = (ts.A, ts.b) if hasattr(ts, "A") else ts
            A = _f64(A).reshape(-1, 9)
            b = _f64(b).reshape(-1)
            if A.shape[0] != b.size or A.shape[0] > _lib.MAX_TERM_ROWS:
                raise ValueError(f"terminal set must have at most {_lib.MAX_TERM_ROWS} rows of 9 coefficients")
            c.terminal_set, c.term_rows = 1, A.shape[0]
            flatA = np.zeros(_lib.MAX_TERM_ROWS * 9)
            flatA[:A.size] = A.reshape(-1)
            flatb = np.zeros(_lib.MAX_TERM_ROWS)
            flatb[:b.size] = b
            c.term_A[:] = list(flatA)
            c.term_b[:] = list(flatb)
        tc = cfg.terminal_cost
        if tc is not None and tc is not False:
            if tc is True:
                from .controllers.tools.terminal_ingredients import load_terminal
                tc = load_terminal()
            t = tc.device_tables(_lib.MAX_TCOST, _lib.MAX_TCOST)
            c.terminal_cost_terms, c.tc_npoly, c.tc_nroot = 1, len(t["poly_coef"]), len(t["root_coef"])

            def put(dst, src, n):
                a = np.zeros(n, dtype=np.asarray(src).dtype if np.asarray(src).size else float)
                a[:np.asarray(src).size] = np.asarray(src).reshape(-1)
                dst[:] = list(a)
            put(c.tc_poly_coef, t["poly_coef"], _lib.MAX_TCOST)
            put(c.tc_poly_exp, t["poly_exp"].astype(int), _lib.MAX_TCOST * 9)
            put(c.tc_root_coef, t["root_coef"], _lib.MAX_TCOST)
            put(c.tc_root_eps, t["root_eps"], _lib.MAX_TCOST)
            put(c.tc_root_pow, t["root_pow"], _lib.MAX_TCOST)
            put(c.tc_root_exp, t["root_exp"].astype(int), _lib.MAX_TCOST * 9)
            c.tc_const = float(t["const"])
            if cfg.P is None:
                c.P[:] = list(_f64(tc.P, 81))
        return c

    def __init__(self, cfg: MPCConfig | None = None, **kw):
        cfg = cfg or MPCConfig(**kw)
        self.cfg = cfg
        self.lib = _lib.load_library()
        c = self.make_c_config(self.lib, cfg)
        self.D = np.array(list(c.D))[:6 * cfg.NT].reshape(6, cfg.NT)     # the struct packs D at row stride NT
        self.r = np.array(list(c.r))
        self.P = np.array(list(c.P)).reshape(9, 9)
        self._c = c
        self._h = C.c_void_p()
        rc = self.lib.ftmpc_create(C.byref(c), C.byref(self._h))
        if rc != 0:
            msg = self.lib.ftmpc_last_error(None).decode()
            raise _lib.FtmpcError(rc, msg)

    # -- lifetime -----------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ftmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise _lib.FtmpcError(rc, self.lib.ftmpc_last_error(self._h).decode())

    def reserve(self, max_batch: int):
        self._check(self.lib.ftmpc_reserve(self._h, int(max_batch)))

    def kernel_name(self, slot: int) -> str:
        return self.lib.ftmpc_kernel_name(int(slot)).decode()

    # -- host-buffer path ---------------------------------------------------------------
    def _refs(self, B, xref, uref):
        N = self.cfg.N
        xref = _f64(xref)
        xs = 0 if xref.size == 9 * (N + 1) else xref.size // B
        us = 0
        if uref is not None:
            uref = _f64(uref)
            us = 0 if uref.size == 6 * (N + 1) else uref.size // B
        return xref, xs, uref, us

    def solve(self, x0, ub, stuck, xref, uref=None, warmU=None, return_U=False, relinearize=0):
        """x0 [B,13], ub/stuck [B,NT], xref 9x(N+1) column-major flat (shared) or [B, 9(N+1)],
        uref likewise with 6 rows or None (hover), warmU [B,N,NT] or None (updated in place).
        relinearize=k runs k further QP steps, each linearised about the previous solution (sequential
        QP towards the reference's nonlinear program; iterations accumulate).
        Returns dict(u0 [B,NT], U [B,N,NT]|None, status [B], iters [B])."""
        if relinearize > 0:
            out = self.solve(x0, ub, stuck, xref, uref=uref, warmU=warmU, return_U=True)
            W = np.ascontiguousarray(out["U"])
            iters = out["iters"].copy()
            for _ in range(int(relinearize)):
                out = self.solve(x0, ub, stuck, xref, uref=uref, warmU=W, return_U=True)   # W <- U* in place
                iters += out["iters"]
            if warmU is not None:
                warmU[...] = W.reshape(warmU.shape)
            out["iters"] = iters
            if not return_U:
                out["U"] = None
            return out
        N, NT = self.cfg.N, self.cfg.NT
        x0 = _f64(x0).reshape(-1, 13)
        B = x0.shape[0]
        ub = _f64(ub, (B, NT))
        stuck = _f64(stuck, (B, NT))
        xref, xs, uref, us = self._refs(B, xref, uref)
        if warmU is not None and not (isinstance(warmU, np.ndarray) and warmU.dtype == np.float64
                                      and warmU.flags.c_contiguous and warmU.size == B * N * NT):
            raise ValueError("warmU must be a C-contiguous float64 array of B*N*NT (updated in place)")
        u0 = np.empty((B, NT))
        U = np.empty((B, N, NT)) if return_U else None
        status = np.empty(B, np.int32)
        iters = np.empty(B, np.int32)
        self._check(self.lib.ftmpc_solve_batch(self._h, B, _ptr(x0), _ptr(ub), _ptr(stuck), _ptr(xref), xs,
                                               _ptr(uref), us, _ptr(warmU), _ptr(u0), _ptr(U),
                                               _ptr(status, C.c_int32), _ptr(iters, C.c_int32)))
        return dict(u0=u0, U=U, status=status, iters=iters)

    # -- towards the reference's nonlinear program: sequential QP with a line search on the true cost -----------
    def eval_cost(self, x0, ub, stuck, xref, U, uref=None):
        """Cost of the NONLINEAR program (nonlinear rollout, full terminal cost when MPCConfig.terminal_cost is set)
        for thruster sequences U [B,N,NT] -> [B]   (include/ftmpc.h ftmpc_eval_cost_batch)."""
        N, NT = self.cfg.N, self.cfg.NT
        x0 = _f64(x0).reshape(-1, 13)
        B = x0.shape[0]
        ub = _f64(ub, (B, NT))
        stuck = _f64(stuck, (B, NT))
        U = _f64(U, (B, N, NT))
        xref, xs, uref, us = self._refs(B, xref, uref)
        J = np.empty(B)
        self._check(self.lib.ftmpc_eval_cost_batch(self._h, B, _ptr(x0), _ptr(ub), _ptr(stuck), _ptr(xref), xs, _ptr(uref), us,
                                                   _ptr(U), _ptr(J)))
        return J

    def solve_sqp(self, x0, ub, stuck, xref, uref=None, warmU=None, sqp_iters=10, tol=1e-9, backtracks=8):
        """Globalised sequential QP towards the reference's NLP (spiraling_mpc.py:87-238: nonlinear dynamics, full
        terminal cost) in thruster space.  Each major iteration solves the QP linearised about the current U (Hessian
        2 (B'QB + R + rho I) with the quadratic terminal weight P, gradient exact -- including the non-quadratic terminal
        terms when MPCConfig.terminal_cost is set), then backtracks alpha = 1, 1/2, ... along U_qp - U on the TRUE cost
        (eval_cost) until it decreases; an instance stops when no step decreases its cost by more than tol (1 + |J|).
        Plain re-linearisation (`solve(relinearize=k)`) has no such safeguard and oscillates on the smoothed |.|^(1/4)
        terms of the terminal cost.  Returns dict(u0, U, cost [B], cost0 [B] (at the start point), sqp_iters [B],
        iters [B] (IPM iterations summed), status [B] of the last QP)."""
        N, NT = self.cfg.N, self.cfg.NT
        x0 = _f64(x0).reshape(-1, 13)
        B = x0.shape[0]
        ub = _f64(ub, (B, NT))
        stuck = _f64(stuck, (B, NT))
        U = np.zeros((B, N, NT)) if warmU is None else np.clip(_f64(warmU, (B, N, NT)), 0.0, ub[:, None, :])
        J = self.eval_cost(x0, ub, stuck, xref, U, uref)
        J0 = J.copy()
        active = np.ones(B, bool)
        n_major = np.zeros(B, np.int32)
        ipm = np.zeros(B, np.int32)
        status = np.zeros(B, np.int32)
        for _ in range(int(sqp_iters)):
            if not active.any():
                break
            W = U.copy()                      # solve() overwrites its warm-start buffer with the QP solution
            out = self.solve(x0, ub, stuck, xref, uref=uref, warmU=W, return_U=True)     # W <- U_qp
            ipm += np.where(active, out["iters"], 0)
            status = np.where(active, out["status"], status)
            step = np.clip(out["U"], 0.0, ub[:, None, :]) - U      # the fp32 kernels return ub rounded to float32
            alpha = np.ones(B)
            todo = active & (out["status"] != 2)
            improved = np.zeros(B, bool)
            for _bt in range(int(backtracks)):
                if not todo.any():
                    break
                Jt = self.eval_cost(x0, ub, stuck, xref, U + alpha[:, None, None] * step, uref)
                ok = todo & (Jt < J - tol * (1.0 + np.abs(J)))
                U[ok] += alpha[ok, None, None] * step[ok]
                J[ok] = Jt[ok]
                improved |= ok
                todo &= ~ok
                alpha[todo] *= 0.5
            n_major += improved.astype(np.int32)
            active &= improved
        return dict(u0=U[:, 0, :].copy(), U=U, cost=J, cost0=J0, sqp_iters=n_major, iters=ipm, status=status)

    def solve_sqp_device(self, x0, ub, stuck, xref, uref=None, warmU=None, sqp_iters=10, tol=1e-9, backtracks=8):
        """The same line-search SQP with the whole loop on the device (include/ftmpc.h ftmpc_solve_sqp_batch): inputs go up
        once, results come down once; every instance runs `sqp_iters` QP steps and `backtracks` cost evaluations per step
        (a stopped instance's are discarded).  Same return dict as solve_sqp."""
        N, NT = self.cfg.N, self.cfg.NT
        x0 = _f64(x0).reshape(-1, 13)
        B = x0.shape[0]
        ub = _f64(ub, (B, NT))
        stuck = _f64(stuck, (B, NT))
        xref, xs, uref, us = self._refs(B, xref, uref)
        W = None if warmU is None else _f64(warmU, (B, N, NT))
        u0, U = np.empty((B, NT)), np.empty((B, N, NT))
        J, J0 = np.empty(B), np.empty(B)
        nmaj, ipm, st = np.empty(B, np.int32), np.empty(B, np.int32), np.empty(B, np.int32)
        self._check(self.lib.ftmpc_solve_sqp_batch(self._h, B, _ptr(x0), _ptr(ub), _ptr(stuck), _ptr(xref), xs, _ptr(uref), us, _ptr(W),
                                                   int(sqp_iters), int(backtracks), float(tol), _ptr(u0), _ptr(U), _ptr(J), _ptr(J0),
                                                   _ptr(nmaj, C.c_int32), _ptr(ipm, C.c_int32), _ptr(st, C.c_int32)))
        return dict(u0=u0, U=U, cost=J, cost0=J0, sqp_iters=nmaj, iters=ipm, status=st)

    # -- the reference's two-stage structure: 6-D generalized-force QP with the input hull, then allocation ----
    def solve_wrench(self, x0, ub, stuck, xref, uref=None, warmG=None, return_G=False, hull=None):
        """One MPC step in generalized-force space (reference: spiraling_mpc.py:87-238 with the per-stage hull rows
        :133-137,175-177) followed by the min-norm allocation (control_allocator.py:65-94).
        x0 [B,13], ub/stuck [B,NT]; warmG [B,N,6] or None (in/out: total wrenches, already shifted);
        hull: dict from controllers.tools.input_bounds.hull_tables (built here when None).
        Returns dict(u0 [B,NT], tau0 [B,6], G [B,N,6]|None, status [B], iters [B], alloc_status [B]); instances whose
        healthy thrusters do not span R^6 have no hull: status 3, u0 = tau0 = NaN (use `solve` for them)."""
        from .controllers.tools.input_bounds import hull_tables
        N, NT = self.cfg.N, self.cfg.NT
        x0 = _f64(x0).reshape(-1, 13)
        B = x0.shape[0]
        ub = _f64(ub, (B, NT))
        stuck = _f64(stuck, (B, NT))
        if hull is None:
            hull = hull_tables(self.D, ub, stuck)
        ok = ~np.asarray(hull["degenerate"], bool)
        u0 = np.full((B, NT), np.nan)
        tau0 = np.full((B, 6), np.nan)
        G = np.full((B, N, 6), np.nan) if return_G else None
        status = np.full(B, 3, np.int32)
        iters = np.zeros(B, np.int32)
        ast = np.zeros(B, np.int32)
        if not ok.any():
            return dict(u0=u0, tau0=tau0, G=G, status=status, iters=iters, alloc_status=ast)
        sel = np.flatnonzero(ok)
        full = sel.size == B
        take = (lambda a: a) if full else (lambda a: np.ascontiguousarray(a[sel]))
        xref, xs, uref, us = self._refs(B, xref, uref)
        if xs:
            xref = take(xref.reshape(B, -1))
        if uref is not None and us:
            uref = take(uref.reshape(B, -1))
        W = None
        if warmG is not None:
            if not (isinstance(warmG, np.ndarray) and warmG.dtype == np.float64 and warmG.flags.c_contiguous and warmG.size == B * N * 6):
                raise ValueError("warmG must be a C-contiguous float64 array of B*N*6 (updated in place)")
            W = warmG.reshape(B, N, 6) if full else np.ascontiguousarray(warmG.reshape(B, N, 6)[sel])
        b = sel.size
        A = np.ascontiguousarray(hull["A"], dtype=np.float64)
        hs = np.ascontiguousarray(take(hull["set"]), dtype=np.int32)
        hb = np.ascontiguousarray(take(hull["b"]), dtype=np.float64)
        o_u0, o_t0 = np.empty((b, NT)), np.empty((b, 6))
        o_G = np.empty((b, N, 6)) if return_G else None
        o_st, o_it, o_as = np.empty(b, np.int32), np.empty(b, np.int32), np.empty(b, np.int32)
        self._check(self.lib.ftmpc_solve_wrench_batch(
            self._h, b, _ptr(take(x0)), _ptr(take(ub)), _ptr(take(stuck)), _ptr(A), A.shape[0], _ptr(hs, C.c_int32), _ptr(hb),
            int(hull["rows"]), _ptr(xref), xs, _ptr(uref), us, _ptr(W), _ptr(o_u0), _ptr(o_t0), _ptr(o_G),
            _ptr(o_st, C.c_int32), _ptr(o_it, C.c_int32), _ptr(o_as, C.c_int32)))
        u0[sel], tau0[sel], status[sel], iters[sel], ast[sel] = o_u0, o_t0, o_st, o_it, o_as
        if return_G:
            G[sel] = o_G
        if warmG is not None and not full:
            warmG.reshape(B, N, 6)[sel] = W
        return dict(u0=u0, tau0=tau0, G=G, status=status, iters=iters, alloc_status=ast)

    # -- closed loop on the device (SimulationEnvironment.run_simulation, batched) ------------
    def simulate(self, x0, ub, stuck, xref_traj, T, uref_traj=None, noise=(1e-3, 1e-3, 1e-3, 1e-3), seed=0,
                 return_inputs=False, sqp_iters=0, backtracks=8, tol=1e-9, formulation="thruster", hull=None):
        """T closed-loop steps (MPC step -> plant RK4 -> noise -> renormalise) without host round trips.
        sqp_iters > 0: every step solves the nonlinear program by that many major iterations of the line-search SQP
        (solve_sqp_device) instead of one QP step.  formulation="wrench": every step is the reference's two-stage structure
        (solve_wrench: generalized-force MPC with the input hull, then allocation); every vehicle's healthy thrusters must span R^6.
        xref_traj: 9 x (T+N) (column t..t+N is the window of step t), uref_traj: 6 x (T+N) or None.
        Returns dict(x [B,13] final states, u [T,B,NT]|None, not_converged [T])."""
        N, NT = self.cfg.N, self.cfg.NT
        x = _f64(x0).reshape(-1, 13).copy()
        B = x.shape[0]
        ub = _f64(ub, (B, NT))
        stuck = _f64(stuck, (B, NT))
        xr = _f64(xref_traj)
        if xr.shape != (9, T + N):
            raise ValueError(f"xref_traj must be 9 x (T+N) = 9 x {T + N}")
        xr = np.ascontiguousarray(xr.reshape(-1, order="F"))
        ur = None
        if uref_traj is not None:
            ur = _f64(uref_traj)
            if ur.shape != (6, T + N):
                raise ValueError("uref_traj must be 6 x (T+N)")
            ur = np.ascontiguousarray(ur.reshape(-1, order="F"))
        nz = _f64(noise, 4)
        uh = np.empty((T, B, NT)) if return_inputs else None
        bad = np.zeros(T, np.int32)
        if formulation == "wrench":
            if sqp_iters:
                raise ValueError("the two-stage loop solves one QP per step (sqp_iters must be 0)")
            from .controllers.tools.input_bounds import hull_tables
            if hull is None:
                hull = hull_tables(self.D, ub, stuck)
            if np.asarray(hull["degenerate"], bool).any():
                raise ValueError("a vehicle's healthy thrusters do not span R^6: no input hull (use the thruster formulation)")
            A = np.ascontiguousarray(hull["A"], dtype=np.float64)
            hs = np.ascontiguousarray(hull["set"], dtype=np.int32)
            hb = np.ascontiguousarray(hull["b"], dtype=np.float64)
            abad = np.zeros(T, np.int32)
            self._check(self.lib.ftmpc_simulate_wrench_batch(self._h, B, int(T), _ptr(x), _ptr(ub), _ptr(stuck), _ptr(A), A.shape[0],
                                                             _ptr(hs, C.c_int32), _ptr(hb), int(hull["rows"]), _ptr(xr), _ptr(ur), _ptr(nz),
                                                             C.c_uint64(int(seed)), _ptr(uh), _ptr(bad, C.c_int32), _ptr(abad, C.c_int32)))
            return dict(x=x, u=uh, not_converged=bad, alloc_failed=abad)
        if formulation != "thruster":
            raise ValueError("formulation must be 'thruster' or 'wrench'")
        self._check(self.lib.ftmpc_simulate_batch_ex(self._h, B, int(T), _ptr(x), _ptr(ub), _ptr(stuck), _ptr(xr), _ptr(ur),
                                                     _ptr(nz), C.c_uint64(int(seed)), int(sqp_iters), int(backtracks), float(tol),
                                                     _ptr(uh), _ptr(bad, C.c_int32)))
        return dict(x=x, u=uh, not_converged=bad)

    def sqp_graph_launches(self) -> int:
        """Calls of solve_sqp_device on this handle that were replayed from the recorded hipGraph (ftmpc_sqp_graph_launches)."""
        return int(self.lib.ftmpc_sqp_graph_launches(self._h))

    def last_handed_over(self) -> int:
        """Instances of the last two-stage step that the one-wave fp32 kernel handed to the float64 kernel (ftmpc_last_handed_over)."""
        c = C.c_int64(0)
        self._check(self.lib.ftmpc_last_handed_over(self._h, C.byref(c)))
        return int(c.value)

    # -- standalone thruster allocation (ControlAllocator.get_physical_input, batched) ---------
    def allocate(self, tau, ub):
        """min |u|^2 s.t. D u = tau, 0 <= u <= ub for B generalized forces tau [B,6] (reference:
        controllers/tools/control_allocator.py:27-40,65-94).  Returns dict(u [B,NT], status [B], iters [B]);
        status 2 = tau not attainable with these bounds (the reference exit()s there)."""
        NT = self.cfg.NT
        tau = _f64(tau).reshape(-1, 6)
        B = tau.shape[0]
        ub = _f64(ub, (B, NT))
        u = np.empty((B, NT))
        status = np.zeros(B, np.int32)
        iters = np.zeros(B, np.int32)
        self._check(self.lib.ftmpc_allocate_batch(self._h, B, _ptr(tau), _ptr(ub), _ptr(u), _ptr(status, C.c_int32),
                                                  _ptr(iters, C.c_int32)))
        return dict(u=u, status=status, iters=iters)

    # -- device-pointer path (HBM-resident inputs; used by bench.py with torch tensors) --
    def solve_device(self, B, x0, ub, stuck, xref, xref_stride, uref, uref_stride, warmU, out_u0, out_U,
                     status, iters, stream=0):
        """All buffer arguments are integer device addresses (e.g. torch.Tensor.data_ptr())."""
        vp = lambda p: C.c_void_p(int(p)) if p else None
        self._check(self.lib.ftmpc_solve_batch_device(self._h, int(B), vp(x0), vp(ub), vp(stuck), vp(xref),
                                                      int(xref_stride), vp(uref), int(uref_stride), vp(warmU),
                                                      vp(out_u0), vp(out_U), vp(status), vp(iters), vp(stream)))

    def set_profiling(self, on: bool):
        self._check(self.lib.ftmpc_set_profiling(self._h, 1 if on else 0))

    def last_kernel_ms(self):
        """{kernel name: device ms} of the last profiled solve (kernels that were launched)."""
        ms = (C.c_float * _lib.KERNEL_SLOTS)()
        self._check(self.lib.ftmpc_last_kernel_ms(self._h, ms, _lib.KERNEL_SLOTS))
        return {self.lib.ftmpc_routed_kernel_name(self._h, k).decode(): float(ms[k]) for k in range(_lib.KERNEL_SLOTS) if ms[k] > 0}

    # -- test hook ----------------------------------------------------------------------
    def debug_build_qp(self, x0, ub, stuck, xref, inst, uref=None, warmU=None):
        N, NT = self.cfg.N, self.cfg.NT
        x0 = _f64(x0).reshape(-1, 13)
        B = x0.shape[0]
        ub = _f64(ub, (B, NT))
        stuck = _f64(stuck, (B, NT))
        xref, xs, uref, us = self._refs(B, xref, uref)
        warm = None if warmU is None else _f64(warmU)
        nmax = N * NT
        H = np.zeros(nmax * nmax)
        g = np.zeros(nmax)
        lo = np.zeros(nmax)
        hi = np.zeros(nmax)
        n = C.c_int32(0)
        self._check(self.lib.ftmpc_debug_build_qp(self._h, B, _ptr(x0), _ptr(ub), _ptr(stuck), _ptr(xref), xs,
                                                  _ptr(uref), us, _ptr(warm), int(inst), _ptr(H), H.size,
                                                  _ptr(g), _ptr(lo), _ptr(hi), C.byref(n)))
        n = n.value
        return H[:n * n].reshape(n, n).copy(), g[:n].copy(), lo[:n].copy(), hi[:n].copy()


def make_synthetic_batch(B, N, NT, nfault, seed, f_max=F_MAX, omega_des=(0.0, 0.0, 0.6)):
    """Seeded random-pose / random-fault batch of BASELINE.md section 4:
    p~U(-2,2)^3, v~U(-.5,.5)^3, q uniform on S^3, omega~omega_des+U(-.2,.2)^3, hover reference,
    `nfault` distinct broken thrusters per instance with iid U(0,1) intensities.
    Returns (x0 [B,13], ub [B,NT], stuck [B,NT], xref 9x(N+1))."""
    rng = np.random.default_rng(seed)
    od = np.asarray(omega_des, float)
    x0 = np.zeros((B, 13))
    x0[:, 0:3] = rng.uniform(-2, 2, (B, 3))
    x0[:, 3:6] = rng.uniform(-0.5, 0.5, (B, 3))
    q = rng.standard_normal((B, 4))
    x0[:, 6:10] = q / np.linalg.norm(q, axis=1, keepdims=True)
    x0[:, 10:13] = od + rng.uniform(-0.2, 0.2, (B, 3))
    ub = np.full((B, NT), f_max)
    stuck = np.zeros((B, NT))
    if nfault > 0:
        keys = rng.random((B, NT))
        idx = np.argsort(keys, axis=1)[:, :nfault]
        inten = rng.uniform(0, 1, (B, nfault))
        rows = np.arange(B)[:, None]
        ub[rows, idx] = 0.0
        stuck[rows, idx] = inten * f_max
    xref = np.zeros((9, N + 1))
    xref[6:9, :] = od.reshape(3, 1)
    return x0, ub, stuck, xref
