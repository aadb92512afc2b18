"""
oracle/qp_oracle.py -- float64 restatement of the QP-spec and of the
fixed-iteration Mehrotra primal-dual IPM the HIP kernels run.

TEST INFRASTRUCTURE ONLY (see oracle/refmath.py header).  Parity status: the QP-spec is a
build decision derived from the reference physics/weights (the reference solves an NLP
with IPOPT, spiraling_mpc.py:217-230, tol=1e-3, which is not runnable here): the
*physics* under it is pinned by oracle/refmath.py's golden checks, the *QP solution* is
pinned by an independent exact solver (scipy.optimize.lsq_linear(method='bvls')) and by
KKT-residual checks; parity against IPOPT's NLP output is UNPINNED.

QP-spec (per instance; thruster space; condensed):
  decision  U = (u_0..u_{N-1}), u_k in R^NT, 0 <= u_k <= ub (ub_i = 0 for a broken thruster)
  model     c_hat = c_bar + Bbar (U - Ubar): horizon-stacked linearisation of the RK4 centre
            dynamics (spiral_model.py:44-76 through sys_model.py:138-162) about the nonlinear
            rollout c_bar from c0 = robot_to_center(x0) under Ubar = clip(warm,0,ub) (cold: 0),
            total generalized force gen_k = D (u_k + stuck)
  cost      sum_{k=1}^{N-1} e_k' Q e_k + e_N' P e_N + sum_{k=0}^{N-1} [ut_k' R ut_k + rho |u_k|^2]
            e_k = c_hat_k[0:9] - xref_k ; ut_k = gen_k - ur_k - [f_virt;0]
            (spiraling_mpc.py:171,188: the reference's deviation input u_t with applied input
             u_t + u_r + u_comp and u_comp = [f_virt;0] - D f_fault, spiral_parameters.py:37)
            ur_k = [Rot(qbar_k)^T uref_k[0:3]; uref_k[3:6]]   (spiraling_mpc.py:156-166, with the
            linearisation quaternion in place of the decision-variable quaternion)
            P = quadratic part of terminal.yaml ; rho = allocator min-energy weight
            (control_allocator.py:32) folded in as strict-convexity regulariser
  output    u0 = U*_0 (thruster forces; 0 at broken thrusters), U*, iterations, status
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import refmath as rm


@dataclass
class QPConfig:
    N: int = 20
    NT: int = 8
    dt: float = rm.DT
    mass: float = rm.MASS
    J: np.ndarray = field(default_factory=lambda: rm.INERTIA.copy())
    D: np.ndarray = None
    f_max: float = rm.F_MAX
    Q: np.ndarray = field(default_factory=lambda: rm.Q_DIAG.copy())
    R: np.ndarray = field(default_factory=lambda: rm.R_DIAG.copy())
    P: np.ndarray = field(default_factory=rm.terminal_P_quadratic)
    r: np.ndarray = field(default_factory=rm.spiral_r)
    f_virt: np.ndarray = field(default_factory=lambda: rm.F_VIRT.copy())
    rho: float = 0.05
    iters: int = 16

    def __post_init__(self):
        if self.D is None:
            self.D = rm.allocation_matrix_16() if self.NT == 16 else rm.allocation_matrix_8()
        assert self.D.shape == (6, self.NT)


def linearize(cfg: QPConfig, x0, ub, stuck, warmU=None):
    """Nonlinear rollout + per-stage Jacobians.  Returns cbar (N+1,13), A (N,13,13),
    Bg (N,13,6), Ubar (N,NT)."""
    N, NT = cfg.N, cfg.NT
    Ubar = np.zeros((N, NT)) if warmU is None else np.clip(np.asarray(warmU, float).reshape(N, NT), 0.0, ub)
    c = rm.robot_to_center(x0, cfg.r)
    cbar = np.zeros((N + 1, 13))
    A = np.zeros((N, 13, 13))
    Bg = np.zeros((N, 13, 6))
    cbar[0] = c
    for k in range(N):
        gen = cfg.D @ (Ubar[k] + stuck)
        c, A[k], Bg[k] = rm.rk4_with_jac(c, gen, cfg.r, cfg.dt, cfg.mass, cfg.J)
        cbar[k + 1] = c
    return cbar, A, Bg, Ubar


def build_qp(cfg: QPConfig, x0, ub, stuck, xref, uref=None, warmU=None):
    """Dense condensed QP over the ACTIVE thrusters only.

    Returns dict(H (n,n), g (n,), Ubar_act (n,), ub_act (n,), act (indices into NT), n,
    cbar, Bbar) with  cost(U) = 1/2 d'H d + g'd + const, d = U_act - Ubar_act,
    H = 2 (Bbar' Qbar Bbar + Rbar),  g = gradient of the cost at Ubar.
    The factor 2 is kept so that H,g are the true Hessian/gradient of the reference's
    un-halved stage cost (spiraling_mpc.py:76).
    """
    N, NT = cfg.N, cfg.NT
    ub = np.asarray(ub, float).reshape(NT)
    stuck = np.asarray(stuck, float).reshape(NT)
    xref = np.asarray(xref, float).reshape(9, N + 1)
    cbar, A, Bg, Ubar = linearize(cfg, x0, ub, stuck, warmU)
    act = np.flatnonzero(ub > 0)
    na = act.size
    n = N * na
    Da = cfg.D[:, act]
    # Bbar row-blocks: G[k] = d c_k / d U_act  (13 x n), k = 0..N
    G = np.zeros((N + 1, 13, n))
    for k in range(N):
        G[k + 1] = A[k] @ G[k]
        G[k + 1][:, k * na:(k + 1) * na] = Bg[k] @ Da
    Qm = np.diag(cfg.Q)
    H = np.zeros((n, n))
    g = np.zeros(n)
    for k in range(1, N + 1):
        W = cfg.P if k == N else Qm
        E = G[k][0:9]
        e = cbar[k][0:9] - xref[:, k]
        H += 2 * E.T @ W @ E
        g += 2 * E.T @ W @ e
    Rm = np.diag(cfg.R)
    fv = np.concatenate([cfg.f_virt, np.zeros(3)])
    for k in range(N):
        ur = np.zeros(6)
        if uref is not None:
            u_r = np.asarray(uref, float).reshape(6, N + 1)[:, k]
            ur = np.concatenate([rm.rot(cbar[k][9:13]).T @ u_r[0:3], u_r[3:6]])
        ut = cfg.D @ (Ubar[k] + stuck) - ur - fv
        sl = slice(k * na, (k + 1) * na)
        H[sl, sl] += 2 * (Da.T @ Rm @ Da + cfg.rho * np.eye(na))
        g[sl] += 2 * (Da.T @ Rm @ ut + cfg.rho * Ubar[k][act])
    return dict(H=H, g=g, Ubar=Ubar[:, act].reshape(-1), ub=np.tile(ub[act], N), act=act,
                n=n, na=na, cbar=cbar, A=A, Bg=Bg)


def ipm_box(H, g, lo, hi, iters=16, dtype=np.float64, polish=False, trace=None, mu_stop=None):
    """Mehrotra predictor-corrector IPM with an iteration cap for
         min 1/2 d'H d + g'd   s.t.  lo <= d <= hi      (lo < hi componentwise)
    This is the algorithm the HIP kernel (fault-tolerant-mpc_amd/csrc/ftmpc_solve.hip) and the C
    oracle (oracle/ftmpc_oracle.c) run, step for step:
      * start at the box centre, duals on the central path at mu0 = max(0.02 |grad|_inf * width, 1e-3), step fraction 0.9995
      * slacks s_l = d - lo, s_u = hi - d are carried as state (s += a*ds), never recomputed by
        subtraction; d is read back from the slack of the nearer bound
      * the gradient H d + g is re-evaluated every iteration with float64 accumulation.  The fp32
        kernel evaluates it once, then lets it follow the step (H dd = rhs - Sigma dd from the Newton
        system just solved) and replaces it once by a float64 gradient computed through the stage
        records: the same iterates up to rounding
      * one Cholesky of H + diag(z_l/s_l + z_u/s_u) per iteration, two solves
      * stop when the mean complementarity falls below mu_stop
    `dtype=np.float32` emulates the kernel's arithmetic.  Returns (d, z_l, z_u, iterations).
    """
    T = dtype
    H = H.astype(T)
    g = g.astype(T)
    lo = lo.astype(T)
    hi = hi.astype(T)
    n = g.size
    H64 = H.astype(np.float64)
    g64 = g.astype(np.float64)
    sl = (hi - lo) * T(0.5)
    su = sl.copy()
    if mu_stop is None:
        mu_stop = 1e-13 if T is np.float64 else 1e-10
    mu_stop = T(mu_stop)
    tau = T(0.9995)
    nit = 0
    zl = zu = None
    for it in range(iters + 1):
        d = np.where(sl < su, lo + sl, hi - su)
        grad = (H64 @ d.astype(np.float64) + g64).astype(T)
        if zl is None:
            mu0 = max(T(np.abs(grad).max()) * T((hi - lo).max()) * T(0.02), T(1e-3))
            zl = mu0 / sl
            zu = mu0 / su
        mu = (sl @ zl + su @ zu) / T(2 * n)
        if not (mu >= mu_stop) or it == iters:
            break
        nit += 1
        rd = grad - zl + zu
        Sig = zl / sl + zu / su
        M = H + np.diag(Sig)
        L = np.linalg.cholesky(M) if T is np.float64 else _chol(M)
        # predictor (sigma = 0): rhs = -rd - zl + zu = -grad
        da = _solve(L, -grad)
        dzl_a = -zl - zl * da / sl
        dzu_a = -zu + zu * da / su
        ap = min(T(1), _max_step(sl, da, su))
        ad = min(T(1), _max_step_dual(zl, dzl_a, zu, dzu_a))
        mu_aff = ((sl + ap * da) @ (zl + ad * dzl_a) + (su - ap * da) @ (zu + ad * dzu_a)) / T(2 * n)
        sigma = min(max((mu_aff / mu) ** 3, T(0)), T(1))
        # corrector
        rcl = sl * zl + da * dzl_a - sigma * mu
        rcu = su * zu - da * dzu_a - sigma * mu
        rhs = -rd - rcl / sl + rcu / su
        dd = _solve(L, rhs)
        dzl = (-rcl - zl * dd) / sl
        dzu = (-rcu + zu * dd) / su
        ap = min(T(1), tau * _max_step(sl, dd, su))
        ad = min(T(1), tau * _max_step_dual(zl, dzl, zu, dzu))
        sl = sl + ap * dd
        su = su - ap * dd
        zl = zl + ad * dzl
        zu = zu + ad * dzu
        if trace is not None:
            trace.append((float(mu), float(np.abs(rd).max()), float(ap), float(ad)))
    d = np.where(sl < su, lo + sl, hi - su)
    if polish:
        d = polish_active_set(H, g, lo, hi, d, sl, su, zl, zu, T)
    return d, zl, zu, nit


def _chol(M):
    """Cholesky in the working precision (fp32 emulation: plain right-looking loop)."""
    T = M.dtype.type
    n = M.shape[0]
    L = np.tril(M).copy()
    for j in range(n):
        L[j, j] = np.sqrt(L[j, j])
        L[j + 1:, j] = L[j + 1:, j] / L[j, j]
        L[j + 1:, j + 1:] -= np.tril(np.outer(L[j + 1:, j], L[j + 1:, j])).astype(M.dtype)
    return L


def _solve(L, b):
    import scipy.linalg as sla
    y = sla.solve_triangular(L, b, lower=True, check_finite=False).astype(L.dtype)
    return sla.solve_triangular(L.T, y, lower=False, check_finite=False).astype(L.dtype)


def _max_step(sl, dd, su):
    T = sl.dtype.type
    a = T(1e30)
    neg = dd < 0
    if neg.any():
        a = min(a, (-sl[neg] / dd[neg]).min())
    pos = dd > 0
    if pos.any():
        a = min(a, (su[pos] / dd[pos]).min())
    return min(a, T(1e30))


def _max_step_dual(zl, dzl, zu, dzu):
    T = zl.dtype.type
    a = T(1e30)
    m = dzl < 0
    if m.any():
        a = min(a, (-zl[m] / dzl[m]).min())
    m = dzu < 0
    if m.any():
        a = min(a, (-zu[m] / dzu[m]).min())
    return a


def polish_active_set(H, g, lo, hi, d, sl, su, zl, zu, T=np.float64):
    """Active-set polish: pin variables whose dual dominates its slack, solve the free block."""
    at_lo = zl > sl
    at_hi = zu > su
    free = ~(at_lo | at_hi)
    x = np.where(at_lo, lo, np.where(at_hi, hi, d)).astype(T)
    if free.any():
        Hff = H[np.ix_(free, free)].astype(np.float64)
        rhs = -(g[free].astype(np.float64) + H[np.ix_(free, ~free)].astype(np.float64) @ x[~free].astype(np.float64))
        x[free] = np.linalg.solve(Hff, rhs).astype(T)
        x = np.clip(x, lo, hi)
    return x


def solve_exact(H, g, lo, hi):
    """Independent exact solution: bounded least squares on the Cholesky factor
    (SURVEY.md section 8(c)): 1/2 d'Hd + g'd = 1/2 |L' d + L^-1 g|^2 + const."""
    from scipy.optimize import lsq_linear
    import scipy.linalg as sla
    L = np.linalg.cholesky(H)
    b = -sla.solve_triangular(L, g, lower=True)
    res = lsq_linear(L.T, b, bounds=(lo, hi), method="bvls", tol=1e-15, max_iter=20000)
    return res.x


def kkt_residual(H, g, lo, hi, d):
    """max-norm of the projected-gradient optimality measure."""
    grad = H @ d + g
    pg = d - np.clip(d - grad, lo, hi)
    return np.abs(pg).max()


def solve_instance(cfg: QPConfig, x0, ub, stuck, xref, uref=None, warmU=None, dtype=np.float64,
                   iters=None, exact=False, polish=False):
    """Full oracle step.  Returns (u0 (NT,), U (N,NT), qp dict)."""
    qp = build_qp(cfg, x0, ub, stuck, xref, uref, warmU)
    lo = -qp["Ubar"]
    hi = qp["ub"] - qp["Ubar"]
    if exact:
        d = solve_exact(qp["H"], qp["g"], lo, hi)
    else:
        d, _, _, _ = ipm_box(qp["H"], qp["g"], lo, hi, iters or cfg.iters, dtype, polish=polish)
    U = np.zeros((cfg.N, cfg.NT))
    U[:, qp["act"]] = (qp["Ubar"] + d.astype(float)).reshape(cfg.N, qp["na"])
    return U[0].copy(), U, qp


# --------------------------------------------------------------------------------------
# synthetic instance generator; kept bit-identical
# to ft_mpc_amd.batch.make_synthetic_batch (tests compare the two)
# --------------------------------------------------------------------------------------
def make_batch(B, N, NT, nfault, seed, f_max=rm.F_MAX):
    rng = np.random.default_rng(seed)
    x0 = np.zeros((B, 13))
    x0[:, 0:3] = rng.uniform(-2, 2, (B, 3))
    x0[:, 3:6] = rng.uniform(-0.5, 0.5, (B, 3))
    q = rng.standard_normal((B, 4))
    x0[:, 6:10] = q / np.linalg.norm(q, axis=1, keepdims=True)
    x0[:, 10:13] = rm.OMEGA_DES + rng.uniform(-0.2, 0.2, (B, 3))
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
    xref[6:9, :] = rm.OMEGA_DES.reshape(3, 1)
    return x0, ub, stuck, xref


# ======================================================================================
# Generalized-force (6-D) formulation with the input hull, and the terminal set
# (reference: spiraling_mpc.py:133-137,175-177 hull rows per stage,
#  :199-202 terminal set rows on e_N, controllers/tools/input_bounds.py:43-76 the hull itself).
#
# QP-spec, wrench form (per instance):
#   decision  T = (tau_0..tau_{N-1}), tau_k in R^6 the TOTAL generalized force on the body
#             ( = the reference's  u_t + u_r + u_comp + D f_fault ,  spiraling_mpc.py:171,175 )
#   model     as above with gen_k = tau_k ; linearised about Tbar (warm start, else D stuck: thrusters off)
#   cost      sum e_k'Q e_k + e_N'P e_N + sum ut_k'R ut_k ,  ut_k = tau_k - ur_k - [f_virt;0]  ( = u_t )
#   s.t.      A_hull tau_k <= b_hull  for every stage   (hull of {D u : 0 <= u_i <= ub_i healthy, u_i = stuck_i broken})
#             [ A_T (c_hat_N[0:9] - xref_N) <= b_T ]     optional terminal set
#   output    tau_0 ; the thruster command is the min-norm allocation of  tau_0 - D stuck  (control_allocator.py:65-94)
# ======================================================================================
def zonotope_hrep(D, ub, stuck, tol=1e-9):
    """H-representation A tau <= b of {D u : 0 <= u <= ub (healthy), u_i = stuck_i (ub_i = 0)} without enumerating
    its 2^na corners: the set is a zonotope with generators g_i = D[:, i] ub_i about the centre D (ub/2 + stuck), whose
    facet normals are the directions orthogonal to 5 linearly independent generators.  Rows come in +/- pairs, unit
    normals, sorted lexicographically (scipy's Qhull + np.unique gives the same rows, input_bounds.py:69-74)."""
    from itertools import combinations
    D = np.asarray(D, float)
    act = np.flatnonzero(np.asarray(ub) > 0)
    Gn = D[:, act] * np.asarray(ub, float)[act]
    centre = D @ (np.asarray(ub, float) / 2 + np.asarray(stuck, float))
    if act.size < 6 or np.linalg.matrix_rank(Gn, tol=1e-9 * np.abs(Gn).max()) < 6:
        raise ValueError("degenerate hull: the healthy thrusters do not span R^6 (scipy's Qhull fails on this set too)")
    combos = np.array(list(combinations(range(act.size), 5)))
    M = np.transpose(Gn[:, combos], (1, 2, 0))                     # [ncomb, 5, 6]
    _, sv, Vt = np.linalg.svd(M)
    ok = sv[:, 4] > tol * sv[:, 0]                                   # the 5 generators are independent
    nrm = Vt[ok, 5, :]
    # canonical sign, then unique
    first = np.argmax(np.abs(nrm) > 1e-7, axis=1)
    sgn = np.sign(nrm[np.arange(len(nrm)), first])
    nrm = nrm * sgn[:, None]
    nrm = nrm[np.lexsort(np.round(nrm, 7).T[::-1])]
    keep = np.ones(len(nrm), bool)
    keep[1:] = np.abs(np.diff(np.round(nrm, 7), axis=0)).max(axis=1) > 0
    nrm = nrm[keep]
    half = 0.5 * np.abs(nrm @ Gn).sum(axis=1)
    A = np.vstack([nrm, -nrm])
    b = np.concatenate([nrm @ centre + half, -(nrm @ centre) + half])
    order = np.lexsort(np.round(A, 7).T[::-1])
    return A[order], b[order]


def linearize_wrench(cfg: QPConfig, x0, stuck, warmG=None):
    N = cfg.N
    Tbar = np.tile(cfg.D @ np.asarray(stuck, float), (N, 1)) if warmG is None else np.asarray(warmG, float).reshape(N, 6).copy()
    c = rm.robot_to_center(x0, cfg.r)
    cbar = np.zeros((N + 1, 13))
    A = np.zeros((N, 13, 13))
    Bg = np.zeros((N, 13, 6))
    cbar[0] = c
    for k in range(N):
        c, A[k], Bg[k] = rm.rk4_with_jac(c, Tbar[k], cfg.r, cfg.dt, cfg.mass, cfg.J)
        cbar[k + 1] = c
    return cbar, A, Bg, Tbar


def build_qp_wrench(cfg: QPConfig, x0, ub, stuck, xref, uref=None, warmG=None, hull=None, term_set=None):
    """Dense condensed QP in d = T - Tbar (n = 6N) with inequality rows  C d <= h :
    per stage A_hull d_k <= b - A_hull Tbar_k, then (optional) A_T GN d <= b_T - A_T ebar_N."""
    N = cfg.N
    ub = np.asarray(ub, float)
    stuck = np.asarray(stuck, float)
    xref = np.asarray(xref, float).reshape(9, N + 1)
    cbar, A, Bg, Tbar = linearize_wrench(cfg, x0, stuck, warmG)
    n = 6 * N
    G = np.zeros((N + 1, 13, n))
    for k in range(N):
        G[k + 1] = A[k] @ G[k]
        G[k + 1][:, 6 * k:6 * k + 6] = Bg[k]
    Qm = np.diag(cfg.Q)
    H = np.zeros((n, n))
    g = np.zeros(n)
    for k in range(1, N + 1):
        W = cfg.P if k == N else Qm
        E = G[k][0:9]
        e = cbar[k][0:9] - xref[:, k]
        H += 2 * E.T @ W @ E
        g += 2 * E.T @ W @ e
    Rm = np.diag(cfg.R)
    fv = np.concatenate([cfg.f_virt, np.zeros(3)])
    for k in range(N):
        ur = np.zeros(6)
        if uref is not None:
            u_r = np.asarray(uref, float).reshape(6, N + 1)[:, k]
            ur = np.concatenate([rm.rot(cbar[k][9:13]).T @ u_r[0:3], u_r[3:6]])
        sl = slice(6 * k, 6 * k + 6)
        H[sl, sl] += 2 * Rm
        g[sl] += 2 * Rm @ (Tbar[k] - ur - fv)
    Ah, bh = hull if hull is not None else zonotope_hrep(cfg.D, ub, stuck)
    mh = Ah.shape[0]
    C = np.zeros((N * mh, n))
    h = np.zeros(N * mh)
    for k in range(N):
        C[k * mh:(k + 1) * mh, 6 * k:6 * k + 6] = Ah
        h[k * mh:(k + 1) * mh] = bh - Ah @ Tbar[k]
    centre = cfg.D @ (ub / 2 + stuck)
    d0 = (centre[None, :] - Tbar).reshape(-1)
    eN = cbar[N][0:9] - xref[:, N]
    GN = G[N][0:9]
    if term_set is not None:
        At, bt = term_set
        C = np.vstack([C, At @ GN])
        h = np.concatenate([h, np.asarray(bt, float).reshape(-1) - At @ eN])
    return dict(H=H, g=g, C=C, h=h, d0=d0, Tbar=Tbar, n=n, mh=mh, nhull=N * mh, cbar=cbar, GN=GN, eN=eN, hull=(Ah, bh))


def build_qp_box_terminal(cfg: QPConfig, x0, ub, stuck, xref, term_set, uref=None, warmU=None):
    """The thruster-space QP of build_qp plus the terminal-set rows, in the general form C d <= h
    (box rows first: -d <= Ubar, d <= ub - Ubar)."""
    qp = build_qp(cfg, x0, ub, stuck, xref, uref, warmU)
    n, N, na = qp["n"], cfg.N, qp["na"]
    G = np.zeros((13, n))
    Da = cfg.D[:, qp["act"]]
    for k in range(N):
        G = qp["A"][k] @ G
        G[:, k * na:(k + 1) * na] = qp["Bg"][k] @ Da
    GN = G[0:9]
    eN = qp["cbar"][N][0:9] - np.asarray(xref, float).reshape(9, N + 1)[:, N]
    At, bt = term_set
    bt = np.asarray(bt, float).reshape(-1)
    C = np.vstack([-np.eye(n), np.eye(n), At @ GN])
    h = np.concatenate([qp["Ubar"], qp["ub"] - qp["Ubar"], bt - At @ eN])
    d0 = 0.5 * qp["ub"] - qp["Ubar"]
    qp.update(C=C, h=h, d0=d0, GN=GN, eN=eN, nhull=2 * n)
    return qp


def build_qp_box_state(cfg: QPConfig, x0, ub, stuck, xref, xlb=None, xub=None, uref=None, warmU=None):
    """The thruster-space QP of build_qp plus the reference's optional STATE BOUNDS  xlb <= c_k <= xub  on the orbit-centre state
    of every stage k = 1 .. N-1 (spiraling_mpc.py:129-130,179-185: `con_ineq.append(x_t)` for t < N with bounds xlb / xub, 13
    components, +-inf = no row; the row of stage 0 does not depend on the decision variables), in the general form C d <= h:
    box rows first (-d <= Ubar, d <= ub - Ubar), then per stage the finite upper rows  G_k d <= xub - cbar_k  and the finite lower
    rows  -G_k d <= cbar_k - xlb  (G_k = d c_k / d U, 13 x n).  `srow`: (stage, component, +1 | -1) of every state row."""
    qp = build_qp(cfg, x0, ub, stuck, xref, uref, warmU)
    n, N, na = qp["n"], cfg.N, qp["na"]
    xub = np.full(13, np.inf) if xub is None else np.asarray(xub, float).reshape(13)
    xlb = np.full(13, -np.inf) if xlb is None else np.asarray(xlb, float).reshape(13)
    Da = cfg.D[:, qp["act"]]
    G = np.zeros((13, n))
    rows, hs, srow = [], [], []
    for k in range(N):
        G = qp["A"][k] @ G
        G[:, k * na:(k + 1) * na] = qp["Bg"][k] @ Da      # now G = d c_{k+1} / d U
        if k + 1 < N:
            c = qp["cbar"][k + 1]
            for i in range(13):
                if np.isfinite(xub[i]):
                    rows.append(G[i].copy()); hs.append(xub[i] - c[i]); srow.append((k + 1, i, 1))
                if np.isfinite(xlb[i]):
                    rows.append(-G[i]); hs.append(c[i] - xlb[i]); srow.append((k + 1, i, -1))
    C = np.vstack([-np.eye(n), np.eye(n)] + ([np.array(rows)] if rows else []))
    h = np.concatenate([qp["Ubar"], qp["ub"] - qp["Ubar"], np.array(hs, float)])
    qp.update(C=C, h=h, d0=0.5 * qp["ub"] - qp["Ubar"], nhull=2 * n, srow=srow)
    return qp


def solve_box_state_instance(cfg: QPConfig, x0, ub, stuck, xref, xlb=None, xub=None, uref=None, warmU=None, iters=60, mu_stop=1e-10):
    """Thruster-space QP with the state bounds.  Returns (u0 (NT,), U (N,NT), status, iterations, qp dict)."""
    qp = build_qp_box_state(cfg, x0, ub, stuck, xref, xlb, xub, uref, warmU)
    d, s, z, nit, st = ipm_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d0"], qp["nhull"], iters=iters, mu_stop=mu_stop)
    U = np.zeros((cfg.N, cfg.NT))
    U[:, qp["act"]] = (qp["Ubar"] + (d if st != 2 else 0.0)).reshape(cfg.N, qp["na"])
    qp.update(d=d, z=z, s=s)
    return U[0].copy(), U, st, nit, qp


def ipm_general(H, g, C, h, d0, nfeas, iters=40, mu_stop=1e-10, rp_stop=1e-9, trace=None, polish=True, mu_polish=None):
    """Mehrotra predictor-corrector for  min 1/2 d'Hd + g'd  s.t.  C d + s = h, s >= 0  -- the algorithm of the
    float64 kernel's general-constraint mode (csrc/ftmpc_solve_f64.hip, MODE != 0), step for step:
      * start at d0; the first `nfeas` rows (box or hull rows) are strictly feasible there and keep s = h - C d exactly;
        the remaining rows (terminal set) start at s = max(h - C d0, 0.1) and carry the primal residual r_p = C d + s - h,
        which every step shrinks by (1 - alpha_p)
      * duals on the central path at mu0 = max(0.02 |grad|_inf * max(s), 1e-3); step fraction 0.9995
      * one Cholesky of H + C' diag(z/s) C per iteration, two solves
      * stop when mu < mu_stop and |r_p|_inf < rp_stop (status 0) or at the iteration cap (status 1); C' diag(z/s) C
        with z/s ~ 1/mu on the active rows ruins the conditioning long before float64 runs out in the box form, hence
        mu_stop = 1e-10 here, and a factorisation that breaks down once mu < 1e-7 ends the iteration as converged
        (status 0; earlier: status 2)
      * polish=True (what the kernels do since round 4): a converged iterate is finished by polish_general -- the exact
        solution on the active set the iterate identifies, verified by its signs; every polish round that factorises counts
        as an iteration.  An interior-point iterate at mu 1e-10 is up to 7e-5 f_max from the exact solution where rows
        are weakly active (z ~ s ~ 1e-5); the polished one agrees with the active-set certificate solve_general_exact to 1e-9.
      * mu_polish (kernel 13, csrc/ftmpc_solve_ricw.hip: 1e-7): the iteration is LEFT at that mu (primal residual closed) for the
        polish; verified -> done, the last interior-point iterations saved; else the iterate is taken up again and run to mu_stop.
    Returns (d, s, z, iterations, status)."""
    n, m = g.size, h.size
    d = d0.copy()
    res = h - C @ d
    s = res.copy()
    s[nfeas:] = np.maximum(res[nfeas:], 0.1)
    if (s[:nfeas] <= 0).any():
        raise ValueError("start point is not strictly inside the box / hull rows")
    rp = C @ d + s - h
    grad = H @ d + g
    mu0 = max(0.02 * np.abs(grad).max() * s.max(), 1e-3)
    z = mu0 / s
    tau = 0.9995
    nit, status = 0, 1
    for it in range(iters + 1):
        mu = float(s @ z) / m
        rpn = float(np.abs(rp).max())
        if not (np.isfinite(mu) and np.isfinite(rpn)):
            status = 2
            break
        if mu < mu_stop and rpn < rp_stop:
            status = 0
            break
        if polish and mu_polish is not None and mu < mu_polish and rpn < rp_stop:
            mu_polish = None      # (once)
            dp, lam, rounds, verified = polish_general(H, g, C, h, d, s, z)
            nit += rounds
            if verified:
                return dp, np.maximum(h - C @ dp, 0.0), lam, nit, 0
        if it == iters:
            break
        nit += 1
        w = z / s
        M = H + C.T @ (w[:, None] * C)
        try:
            L = np.linalg.cholesky(M)
        except np.linalg.LinAlgError:
            status = 0 if (mu < 1e-7 and rpn < rp_stop) else 2
            nit -= 1
            break
        # predictor
        rc = s * z
        rhs = -(grad + C.T @ z) + C.T @ ((rc - z * rp) / s)
        da = _solve(L, rhs)
        ds_a = -rp - C @ da
        dz_a = (-rc - z * ds_a) / s
        ap = min(1.0, _pos_step(s, ds_a))
        ad = min(1.0, _pos_step(z, dz_a))
        mu_aff = float((s + ap * ds_a) @ (z + ad * dz_a)) / m
        sigma = min(max((mu_aff / mu) ** 3, 0.0), 1.0)
        # corrector
        rc = s * z + ds_a * dz_a - sigma * mu
        rhs = -(grad + C.T @ z) + C.T @ ((rc - z * rp) / s)
        dd = _solve(L, rhs)
        ds = -rp - C @ dd
        dz = (-rc - z * ds) / s
        ap = min(1.0, tau * _pos_step(s, ds))
        ad = min(1.0, tau * _pos_step(z, dz))
        d = d + ap * dd
        s = s + ap * ds
        z = z + ad * dz
        rp = (1.0 - ap) * rp
        grad = grad + ap * (H @ dd)
        if trace is not None:
            trace.append((mu, rpn, ap, ad))
    if polish and status == 0:
        dp, lam, rounds, verified = polish_general(H, g, C, h, d, s, z)
        nit += rounds
        if verified:
            d, z, s = dp, lam, np.maximum(h - C @ dp, 0.0)
    return d, s, z, nit, status


POLISH_W0 = 1e6       # penalty weight of the polish relative to max diag(H) / |c_i|^2
POLISH_ROUNDS = 3
POLISH_INNER = 2
POLISH_RES_TOL = 1e-10


def polish_general(H, g, C, h, d, s, z, W0=POLISH_W0, rounds=POLISH_ROUNDS, inner=POLISH_INNER):
    """Active-set polish of a converged interior-point iterate -- the mirror of the kernels' finishing stage
    (csrc/ftmpc_solve_f64.hip general-constraint mode).  Active set A = {i : z_i > s_i}.  Per round: the equality-constrained
    problem on A is solved by `inner` steps of the method of multipliers with penalty W_i = W0 max diag(H) / |c_i|^2 on
    the rows of A (one Cholesky of H + C_A' W C_A, the Newton matrix's own shape, and one solve per step):
        (H + C_A' W C_A) dl = -(H d + g) + C_A' (W s_A - lam),   lam += W (C_A dl - s_A),   d += dl,    s_A = h_A - C_A d
    started from lam = z_A; then the signs are checked -- a row of A with lam < 0 leaves, a row outside with h - C d < -1e-10
    enters -- and the round is repeated with the corrected set until nothing changes (verified).
    Returns (d, lam (all rows), rounds that factorised, verified)."""
    d = d.copy()
    m = h.size
    cn2 = np.maximum((C * C).sum(axis=1), 1e-300)
    hs = float(np.abs(np.diag(H)).max())
    act = z > s
    lam_full = np.where(act, z, 0.0)
    done = 0
    for _ in range(rounds):
        A = np.flatnonzero(act)
        CA = C[A]
        W = W0 * hs / cn2[A]
        try:
            L = np.linalg.cholesky(H + CA.T @ (W[:, None] * CA))
        except np.linalg.LinAlgError:
            return d, lam_full, done, False
        done += 1
        lam = lam_full[A].copy()
        for _ in range(inner):
            sA = h[A] - CA @ d
            rhs = -(H @ d + g) + CA.T @ (W * sA - lam)
            dl = _solve(L, rhs)
            lam = lam + W * (CA @ dl - sA)
            d = d + dl
        res = h - C @ d
        lam_full = np.zeros(m)
        lam_full[A] = lam
        new = act.copy()
        new[A[lam < 0.0]] = False
        new[(~act) & (res < -POLISH_RES_TOL)] = True
        if (new == act).all():
            return d, lam_full, done, True
        act = new
        lam_full = np.maximum(lam_full, 0.0)
    return d, lam_full, done, False


def solve_general_exact(H, g, C, h, d, z, s, maxit=200):
    """Solver-independent certificate for  min 1/2 d'Hd + g'd  s.t.  C d <= h : a primal-dual active-set iteration on the
    KKT system [H C_A'; C_A 0] (least-squares solve: dependent active rows are harmless), started from the active set an
    interior-point iterate suggests; one row leaves (most negative multiplier) or enters (most violated) per step.  Returns
    (d, z) with kkt_general ~ 1e-12, or (None, None) when it does not settle within maxit steps."""
    m, n = h.size, g.size
    act = np.asarray(z) > np.asarray(s)
    for _ in range(maxit):
        A = np.flatnonzero(act)
        CA = C[A]
        na = A.size
        K = np.block([[H, CA.T], [CA, np.zeros((na, na))]])
        sol = np.linalg.lstsq(K, np.r_[-g, h[A]], rcond=None)[0]
        dn, lam = sol[:n], sol[n:]
        viol = np.where(~act, C @ dn - h, 0.0)
        if na and lam.min() < -1e-12:
            act[A[np.argmin(lam)]] = False
        elif viol.max() > 1e-11:
            act[np.argmax(viol)] = True
        else:
            zz = np.zeros(m)
            zz[A] = lam
            return dn, zz
    return None, None


def _pos_step(v, dv):
    m = dv < 0
    return float((-v[m] / dv[m]).min()) if m.any() else 1e30


def kkt_general(H, g, C, h, d, z):
    """Optimality certificate of a convex QP with rows C d <= h: (stationarity, primal violation, dual violation,
    complementarity), each in max-norm.  All four ~ 0  <=>  d is THE solution (H positive definite)."""
    r = h - C @ d
    return (float(np.abs(H @ d + g + C.T @ z).max()), float(max(0.0, (-r).max())), float(max(0.0, (-z).max())),
            float(np.abs(r * z).max()))


def solve_wrench_instance(cfg: QPConfig, x0, ub, stuck, xref, uref=None, warmG=None, hull=None, term_set=None, iters=40, mu_stop=1e-10,
                          mu_polish=None):
    """Full oracle step of the generalized-force formulation.  Returns (tau0 (6,), T (N,6), status, iterations, qp dict).
    mu_polish: see ipm_general (1e-7 mirrors kernel 13's iteration counts; the solution is the same exact one either way)."""
    qp = build_qp_wrench(cfg, x0, ub, stuck, xref, uref, warmG, hull, term_set)
    d, s, z, nit, st = ipm_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d0"], qp["nhull"], iters=iters, mu_stop=mu_stop, mu_polish=mu_polish)
    T = qp["Tbar"] + (d.reshape(cfg.N, 6) if st != 2 else 0.0)
    qp.update(d=d, z=z, s=s)
    return T[0].copy(), T, st, nit, qp


def solve_box_terminal_instance(cfg: QPConfig, x0, ub, stuck, xref, term_set, uref=None, warmU=None, iters=40, mu_stop=1e-10):
    """Thruster-space QP with the terminal-set rows.  Returns (u0 (NT,), U (N,NT), status, iterations, qp dict)."""
    qp = build_qp_box_terminal(cfg, x0, ub, stuck, xref, term_set, uref, warmU)
    d, s, z, nit, st = ipm_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d0"], qp["nhull"], iters=iters, mu_stop=mu_stop)
    U = np.zeros((cfg.N, cfg.NT))
    U[:, qp["act"]] = (qp["Ubar"] + (d if st != 2 else 0.0)).reshape(cfg.N, qp["na"])
    qp.update(d=d, z=z, s=s)
    return U[0].copy(), U, st, nit, qp


# ======================================================================================
# The reference's nonlinear program in thruster space and a line-search SQP on it
# (reference spiraling_mpc.py:87-238: RK4 dynamics, stage costs, full terminal cost)
# ======================================================================================
def nlp_cost(cfg: QPConfig, x0, ub, stuck, xref, U, uref=None, terminal_cost=None):
    """sum_{k=1}^{N-1} e_k'Q e_k + V(e_N) + sum_k [ut_k'R ut_k + rho |u_k|^2] along the NONLINEAR rollout;
    V = e'P e, or terminal_cost(e) (callable on the 9-vector: the full terminal.yaml cost) when given."""
    N = cfg.N
    ub = np.asarray(ub, float)
    U = np.where(ub > 0, np.asarray(U, float).reshape(N, cfg.NT), 0.0)
    xref = np.asarray(xref, float).reshape(9, N + 1)
    c = rm.robot_to_center(x0, cfg.r)
    fv = np.concatenate([cfg.f_virt, np.zeros(3)])
    J = 0.0
    for k in range(N):
        gen = cfg.D @ (U[k] + stuck)
        ur = np.zeros(6)
        if uref is not None:
            u_r = np.asarray(uref, float).reshape(6, N + 1)[:, k]
            ur = np.concatenate([rm.rot(c[9:13]).T @ u_r[0:3], u_r[3:6]])
        ut = gen - ur - fv
        J += ut @ (cfg.R * ut) + cfg.rho * (U[k] @ U[k])
        c = rm.rk4(lambda s: rm.centre_dx_dt(s, gen, cfg.r, cfg.mass, cfg.J), c, cfg.dt)
        e = c[0:9] - xref[:, k + 1]
        if k + 1 < N:
            J += e @ (cfg.Q * e)
        else:
            J += float(terminal_cost(e)) if terminal_cost is not None else e @ cfg.P @ e
    return float(J)


def sqp_linesearch(cfg: QPConfig, x0, ub, stuck, xref, uref=None, warmU=None, terminal=None, sqp_iters=10, tol=1e-9, backtracks=8):
    """Mirror of ft_mpc_amd.BatchedMPC.solve_sqp for one instance.  terminal: object with .cost(e) and
    .grad(e, quadratic=False) (the non-quadratic gradient enters g through W e_N), or None.  QPs are solved exactly (BVLS)."""
    N, NT = cfg.N, cfg.NT
    ub = np.asarray(ub, float)
    U = np.zeros((N, NT)) if warmU is None else np.clip(np.asarray(warmU, float).reshape(N, NT), 0.0, ub)
    tc = (lambda e: terminal.cost(e)) if terminal is not None else None
    J = nlp_cost(cfg, x0, ub, stuck, xref, U, uref, tc)
    hist = [J]
    for _ in range(sqp_iters):
        qp = build_qp(cfg, x0, ub, stuck, xref, uref, U)
        if terminal is not None:
            # gradient of the non-quadratic terminal terms at the linearisation point, through GN
            na = qp["na"]
            G = np.zeros((13, qp["n"]))
            Da = cfg.D[:, qp["act"]]
            for k in range(N):
                G = qp["A"][k] @ G
                G[:, k * na:(k + 1) * na] = qp["Bg"][k] @ Da
            eN = qp["cbar"][N][0:9] - np.asarray(xref, float).reshape(9, N + 1)[:, N]
            qp["g"] = qp["g"] + G[0:9].T @ terminal.grad(eN, quadratic=False)
        d = solve_exact(qp["H"], qp["g"], -qp["Ubar"], qp["ub"] - qp["Ubar"])
        Uq = np.zeros((N, NT))
        Uq[:, qp["act"]] = (qp["Ubar"] + d).reshape(N, qp["na"])
        step = Uq - U
        alpha, done = 1.0, True
        for _bt in range(backtracks):
            Jt = nlp_cost(cfg, x0, ub, stuck, xref, U + alpha * step, uref, tc)
            if Jt < J - tol * (1.0 + abs(J)):
                U, J, done = U + alpha * step, Jt, False
                break
            alpha *= 0.5
        hist.append(J)
        if done:
            break
    return U, J, hist


# ---------------------------------------------------------------------------------------------------------
# The Newton systems of ipm_box through wrench space (what kernel 8, csrc/ftmpc_solve_ws.hip, does).
#   The thrusters enter the dynamics only through the stage wrenches D_a u_k, so the condensed Hessian of build_qp is
#       H = DD' H_w DD + 2 rho I,      DD = blockdiag(D_a) (6N x N na),  H_w = the Hessian of build_qp_wrench,
#   and (H + Sigma) x = r with Dg = 2 rho + Sigma (diagonal) has the solution
#       x = Dg^-1 (r - DD' L K^-1 L' DD Dg^-1 r),     H_w = L L',  K = I + L' S L,  S = DD Dg^-1 DD' (6 x 6 blocks).
#   K is 6N x 6N with eigenvalues >= 1 whatever Sigma does; S may be singular (thrusters on their bounds, or healthy
#   thrusters that do not span R^6).  The forms with S^-1 or H_w^-1 are the ones that fail in float32.
# ---------------------------------------------------------------------------------------------------------
def wrench_form(cfg: QPConfig, x0, ub, stuck, xref, uref=None, warmU=None):
    """-> (H_w [6N,6N], D_a [6,na]) with build_qp(...)['H'] == kron(I_N, D_a)' H_w kron(I_N, D_a) + 2 rho I."""
    ub = np.asarray(ub, float)
    act = ub > 0
    stuck = np.asarray(stuck, float)
    warmG = None
    if warmU is not None:
        Ub = np.clip(np.asarray(warmU, float).reshape(cfg.N, cfg.NT), 0.0, ub)
        warmG = (cfg.D @ (Ub + stuck).T).T
    qw = build_qp_wrench(cfg, x0, ub, stuck, xref, uref=uref, warmG=warmG, hull=(np.zeros((1, 6)), np.ones(1)))
    return qw["H"], cfg.D[:, act]


def schur_newton_solver(Hw, Da, N, rho, dtype=np.float64):
    """-> make(Sig) -> solve(r): the wrench-space form above in the given arithmetic (float32 emulates kernel 8)."""
    T = dtype
    na = Da.shape[1]
    Da_t = Da.astype(T)
    Lw = (np.linalg.cholesky(Hw) if T is np.float64 else _chol(Hw.astype(T))).astype(T)

    def make(Sig):
        Dg = (T(2 * rho) + Sig.astype(T)).astype(T)
        S = np.zeros((6 * N, 6 * N), T)
        for k in range(N):
            S[6 * k:6 * k + 6, 6 * k:6 * k + 6] = (Da_t * (T(1) / Dg[k * na:(k + 1) * na])) @ Da_t.T
        K = (np.eye(6 * N, dtype=T) + Lw.T @ (S @ Lw)).astype(T)
        L = np.linalg.cholesky(K) if T is np.float64 else _chol(K)

        def solve(r):
            r = r.astype(T)
            t = np.concatenate([Da_t @ (r[k * na:(k + 1) * na] / Dg[k * na:(k + 1) * na]) for k in range(N)]).astype(T)
            p = _solve(L, (Lw.T @ t).astype(T))
            y = (Lw @ p).astype(T)
            return np.concatenate([(r[k * na:(k + 1) * na] - Da_t.T @ y[6 * k:6 * k + 6]) / Dg[k * na:(k + 1) * na] for k in range(N)])
        return solve
    return make
