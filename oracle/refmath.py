"""
oracle/refmath.py -- float64 NumPy restatement of the reference's pinned physics.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this module; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.  It is the checker,
never the thing measured or shipped.

Parity status: PINNED for every formula/constant below (they are restated from the
reference's source text and data files and checked against golden values in
tests/golden/, see oracle/gen_golden.py).  The reference's *solver outputs*
(CasADi/IPOPT NLP, cvxpy/OSQP allocation) are "parity unpinned": the reference holds no
tests or golden vectors and those third-party solvers (casadi 3.6.7, cvxpy 1.6.4,
osqp 1.0.1) are absent from this image.

Citations are relative to /root/reference/.
"""
from __future__ import annotations

import json
import re
from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------------------
# constants (ft_mpc/models/sys_model.py:52-60, ft_mpc/config/reactive.yaml:2,32-33)
# --------------------------------------------------------------------------------------
MASS = 16.8
INERTIA = np.diag([0.2, 0.3, 0.25])
F_MAX = 3.4
DT = 0.1
NX = 13
NOPT = 9  # costed centre states p,v,omega (spiraling_mpc.py:57)
Q_DIAG = np.array([1, 1, 1, 1, 1, 1, 2, 2, 2], dtype=float)
R_DIAG = np.array([0.1, 0.1, 0.1, 0.01, 0.01, 0.01], dtype=float)
OMEGA_DES = np.array([0.0, 0.0, 0.6])  # spiral_parameters.py:33
F_VIRT = np.array([0.0, 3.5, 0.0])  # spiral_parameters.py:34-36


def allocation_matrix_16() -> np.ndarray:
    """6x16 thruster allocation matrix D (sys_model.py:73-123).

    Restated from the thruster geometry rather than entry by entry: thrusters 0-7 push
    along -/+x, 8-11 along -/+y, 12-15 along -/+z; torque = lever x force with lever
    arms d1=.12, d2=.09, d3=.05.  tests/test_oracle_pinned.py cross-checks this against
    the entry table (data/InertialProperties.md:30-41) and r x F from
    util/animate.py:66-110.
    """
    d1, d2, d3 = 0.12, 0.09, 0.05
    D = np.zeros((6, 16))
    fx = np.array([-1, -1, 1, 1, -1, -1, 1, 1.0])
    D[0, 0:8] = fx
    D[4, 0:8] = d3 * np.array([-1, 1, 1, -1, -1, 1, 1, -1.0])
    D[5, 0:8] = d1 * np.array([1, 1, -1, -1, -1, -1, 1, 1.0])
    D[1, 8:12] = [-1, -1, 1, 1]
    D[5, 8:12] = d2 * np.array([-1, 1, 1, -1.0])
    D[2, 12:16] = [-1, 1, -1, 1]
    D[3, 12:16] = d1 * np.array([-1, 1, 1, -1.0])
    return D


def allocation_matrix_8() -> np.ndarray:
    """SYNTHETIC 6x8 allocation matrix for the NT=8 benchmark configs.

    NOT a reference quantity (the reference has no 6-DoF 8-thruster
    vehicle).  Eight canted thrusters on the corners of the 0.30 x 0.24 x 0.10 m body
    (the corner coordinates of util/animate.py:72-80), each pushing along a unit vector
    with components (-sx*0.6, -sy*0.64*e, -sz*0.48*e') -- chosen so that D positively
    spans R^6 (checked in tests/test_oracle_pinned.py by an LP).  Kept bit-identical to
    ft_mpc_amd.models.sys_model.allocation_matrix_8 (tests compare the two).
    """
    pos = np.array([[sx * 0.15, sy * 0.12, sz * 0.05]
                    for sx in (1, -1) for sy in (1, -1) for sz in (1, -1)])
    D = np.zeros((6, 8))
    for i, p in enumerate(pos):
        sx, sy, sz = np.sign(p)
        # alternate the cant so that torques of both signs exist about every axis
        flip = 1.0 if (i % 2 == 0) else -1.0
        flip2 = 1.0 if ((i // 2) % 2 == 0) else -1.0
        flip3 = 1.0 if ((i // 4) % 2 == 0) else -1.0
        f = np.array([-sx * 0.6 * flip2, -sy * 0.64 * flip, -sz * 0.48 * flip3])
        D[0:3, i] = f
        D[3:6, i] = np.cross(p, f)
    return D


# --------------------------------------------------------------------------------------
# rotations / quaternion kinematics
# --------------------------------------------------------------------------------------
def rot(q: np.ndarray) -> np.ndarray:
    """World->body rotation from quaternion [x,y,z,w] (util/utils.py:4-19).
    No unit-norm assumption (the reference keeps the raw quadratic forms)."""
    x, y, z, w = np.asarray(q, dtype=float).reshape(4)
    return np.array([
        [x * x - y * y - z * z + w * w, 2 * (x * y + z * w), 2 * (x * z - y * w)],
        [2 * (x * y - z * w), -x * x + y * y - z * z + w * w, 2 * (y * z + x * w)],
        [2 * (x * z + y * w), 2 * (y * z - x * w), -x * x - y * y + z * z + w * w],
    ])


def rot_inv(q):
    """Body->world (util/utils.py:21-31)."""
    return rot(q).T


def rot_full(q):
    """6x6 blockdiag(Rot(q), I3) (util/utils.py:57-70)."""
    M = np.eye(6)
    M[0:3, 0:3] = rot(q)
    return M


def omega_op(w: np.ndarray) -> np.ndarray:
    """4x4 Omega(omega) with qdot = 1/2 Omega q (sys_model.py:8-29)."""
    wx, wy, wz = np.asarray(w, dtype=float).reshape(3)
    return np.array([
        [0.0, wz, -wy, wx],
        [-wz, 0.0, wx, wy],
        [wy, -wx, 0.0, wz],
        [-wx, -wy, -wz, 0.0],
    ])


def skew(a):
    ax, ay, az = np.asarray(a, dtype=float).reshape(3)
    return np.array([[0, -az, ay], [az, 0, -ax], [-ay, ax, 0.0]])


# --------------------------------------------------------------------------------------
# fault bookkeeping (sys_model.py:228-243, util/broken_thruster.py)
# --------------------------------------------------------------------------------------
@dataclass
class FaultState:
    """stuck[i] = intensity*f_max for broken thrusters, ub[i] = 0 for broken else f_max."""
    nt: int
    f_max: float = F_MAX
    stuck: np.ndarray = field(default=None)
    ub: np.ndarray = field(default=None)

    def __post_init__(self):
        if self.stuck is None:
            self.stuck = np.zeros(self.nt)
        if self.ub is None:
            self.ub = np.full(self.nt, self.f_max)

    def set_fault(self, index: int, intensity: float):
        self.stuck[index] = intensity * self.f_max
        self.ub[index] = 0.0
        return self


# --------------------------------------------------------------------------------------
# plant (16-thruster space) and orbit-centre models
# --------------------------------------------------------------------------------------
def plant_dx_dt(x, u, D, stuck, ub, mass=MASS, J=INERTIA):
    """SystemModel.dx_dt (sys_model.py:177-226): x=[p,v,q,omega]; broken thrusters'
    commands are zeroed and their stuck force added."""
    x = np.asarray(x, dtype=float).reshape(13)
    u = np.where(np.asarray(ub) > 0, np.asarray(u, dtype=float).reshape(-1), 0.0)
    gen = D @ (u + stuck)
    F, tau = gen[0:3], gen[3:6]
    v, q, w = x[3:6], x[6:10], x[10:13]
    Jw = J @ w
    return np.concatenate([
        v,
        rot(q).T @ F / mass,
        0.5 * omega_op(w) @ q,
        np.linalg.solve(J, tau - np.cross(w, Jw)),
    ])


def centre_dx_dt(c, gen, r, mass=MASS, J=INERTIA):
    """SpiralModel.dx_dt (spiral_model.py:44-76): c=[p_c,v_c,omega,q];
    `gen` is the TOTAL generalized force [F;tau] acting on the body (the reference passes
    u and adds D f_fault inside; callers here do that sum)."""
    c = np.asarray(c, dtype=float).reshape(13)
    F, tau = gen[0:3], gen[3:6]
    v, w, q = c[3:6], c[6:9], c[9:13]
    dw = np.linalg.solve(J, tau - np.cross(w, J @ w))
    a_b = F / mass + np.cross(dw, r) + np.cross(w, np.cross(w, r))
    return np.concatenate([v, rot(q).T @ a_b, dw, 0.5 * omega_op(w) @ q])


def rk4(f, x, dt=DT):
    """SystemModel.rk4_integrator (sys_model.py:138-162); no quaternion renormalisation."""
    k1 = f(x)
    k2 = f(x + dt / 2 * k1)
    k3 = f(x + dt / 2 * k2)
    k4 = f(x + dt * k3)
    return x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)


def normalize_quaternion_plant(x):
    """sys_model.py:164-175."""
    x = np.array(x, dtype=float)
    x[6:10] /= np.linalg.norm(x[6:10])
    return x


def spiral_r(mass=MASS):
    """spiral_parameters.py:39 -> r = |f_virt|/(m |omega_des|^2) * y_hat."""
    return np.linalg.norm(F_VIRT) / (mass * np.linalg.norm(OMEGA_DES) ** 2) * np.array([0.0, 1.0, 0.0])


def compensation_force(D, stuck):
    """spiral_parameters.py:37: [f_virt;0] - D f_fault."""
    return np.concatenate([F_VIRT, np.zeros(3)]) - D @ stuck


def robot_to_center(x, r):
    """spiral_model.py:91-109 (also reorders [q,omega] -> [omega,q])."""
    x = np.asarray(x, dtype=float).reshape(13)
    q, w = x[6:10], x[10:13]
    Rt = rot(q).T
    return np.concatenate([x[0:3] + Rt @ r, x[3:6] + Rt @ np.cross(w, r), w, q])


# --------------------------------------------------------------------------------------
# analytic Jacobians of the centre model (the reference gets these by CasADi AD,
# spiraling_mpc.py:217-230); SURVEY.md Appendix A
# --------------------------------------------------------------------------------------
def drotT_a_dq(q, a):
    """d(Rot(q)^T a)/dq, 3x4, entries linear in q (differentiate utils.py:15-19)."""
    x, y, z, w = q
    a0, a1, a2 = a
    # Rot(q)^T a = sum_j a_j * row_j(Rot(q))   (row j of Rot is column j of Rot^T)
    # d row0 / dq, d row1 / dq, d row2 / dq are 3x4 each
    d0 = 2 * np.array([[x, -y, -z, w], [y, x, w, z], [z, -w, x, -y]])
    d1 = 2 * np.array([[y, x, -w, -z], [-x, y, -z, w], [w, z, y, x]])
    d2 = 2 * np.array([[z, w, x, y], [-w, z, y, -x], [-x, -y, z, w]])
    return a0 * d0 + a1 * d1 + a2 * d2


def centre_jac(c, gen, r, mass=MASS, J=INERTIA):
    """(f_c 13x13, f_g 13x6) of centre_dx_dt at (c, gen)."""
    w, q = c[6:9], c[9:13]
    F, tau = gen[0:3], gen[3:6]
    Jinv = np.linalg.inv(J)
    dw = Jinv @ (tau - np.cross(w, J @ w))
    a_b = F / mass + np.cross(dw, r) + np.cross(w, np.cross(w, r))
    Rt = rot(q).T
    dwdw = -Jinv @ (skew(w) @ J - skew(J @ w))
    dab_dw = -skew(r) @ dwdw - skew(np.cross(w, r)) - skew(w) @ skew(r)
    xi = 0.5 * np.array([[q[3], -q[2], q[1]], [q[2], q[3], -q[0]], [-q[1], q[0], q[3]],
                         [-q[0], -q[1], -q[2]]])
    fc = np.zeros((13, 13))
    fg = np.zeros((13, 6))
    fc[0:3, 3:6] = np.eye(3)
    fc[3:6, 6:9] = Rt @ dab_dw
    fc[3:6, 9:13] = drotT_a_dq(q, a_b)
    fc[6:9, 6:9] = dwdw
    fc[9:13, 6:9] = xi
    fc[9:13, 9:13] = 0.5 * omega_op(w)
    fg[3:6, 0:3] = Rt / mass
    fg[3:6, 3:6] = Rt @ (-skew(r) @ Jinv)
    fg[6:9, 3:6] = Jinv
    return fc, fg


def rk4_with_jac(c, gen, r, dt=DT, mass=MASS, J=INERTIA):
    """One RK4 step of the centre model and its Jacobians (A 13x13, Bg 13x6)."""
    f = lambda s: centre_dx_dt(s, gen, r, mass, J)
    I = np.eye(13)
    k1 = f(c)
    c2 = c + dt / 2 * k1
    k2 = f(c2)
    c3 = c + dt / 2 * k2
    k3 = f(c3)
    c4 = c + dt * k3
    k4 = f(c4)
    nxt = c + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    fc1, fg1 = centre_jac(c, gen, r, mass, J)
    fc2, fg2 = centre_jac(c2, gen, r, mass, J)
    fc3, fg3 = centre_jac(c3, gen, r, mass, J)
    fc4, fg4 = centre_jac(c4, gen, r, mass, J)
    K1 = fc1
    K2 = fc2 @ (I + dt / 2 * K1)
    K3 = fc3 @ (I + dt / 2 * K2)
    K4 = fc4 @ (I + dt * K3)
    G1 = fg1
    G2 = fg2 + fc2 @ (dt / 2 * G1)
    G3 = fg3 + fc3 @ (dt / 2 * G2)
    G4 = fg4 + fc4 @ (dt * G3)
    A = I + dt / 6 * (K1 + 2 * K2 + 2 * K3 + K4)
    Bg = dt / 6 * (G1 + 2 * G2 + 2 * G3 + G4)
    return nxt, A, Bg


# --------------------------------------------------------------------------------------
# reference trajectory (util/get_trajectory.py:43-184, spiraling_mpc.py:255-286,356-365)
# --------------------------------------------------------------------------------------
def hover_trajectory(dt, duration, position=(0.0, 0.0, 0.0)):
    """`hover` / `hover_x_y_z`: 13 x (10*duration/dt) constant reference, identity
    quaternion [0,0,0,1] (get_trajectory.py:109-124)."""
    T = np.arange(0, 10 * duration, dt).size
    x = np.zeros((13, T))
    x[0:3, :] = np.asarray(position, dtype=float).reshape(3, 1)
    x[9, :] = 1.0
    return x


def circle_trajectory(dt, duration, radius=2.0, s_per_circle=30.0):
    """get_trajectory.py:125-141."""
    t = np.arange(0, 10 * duration, dt)
    om = 2 * np.pi / s_per_circle
    x = np.zeros((13, t.size))
    x[0] = radius * np.cos(om * t) - radius
    x[1] = radius * np.sin(om * t)
    x[3] = -radius * om * np.sin(om * t)
    x[4] = radius * om * np.cos(om * t)
    x[9] = 1.0
    return x


def assign_trajectory(traj, horizon, dt=DT, mass=MASS):
    """spiraling_mpc.py:255-286 -> (trajectory 9xT', nominal_input 6xT')."""
    ext = np.hstack([traj, np.tile(traj[:, -1:], (1, horizon))])
    xr = np.vstack([ext[0:6], np.tile(OMEGA_DES.reshape(3, 1), (1, ext.shape[1]))])
    acc = np.gradient(np.gradient(xr[0:3], axis=1), axis=1) / dt ** 2
    ur = np.vstack([acc * mass, np.zeros_like(acc)])
    return xr, ur


def trajectory_window(xr, ur, t, horizon, dt=DT):
    """spiraling_mpc.py:356-365."""
    s = int(t / dt)
    return xr[:, s:s + horizon + 1], ur[:, s:s + horizon + 1]


# --------------------------------------------------------------------------------------
# terminal ingredients (controllers/tools/terminal_ingredients.py:451-474 +
# config/terminal.yaml) -- parsed WITHOUT eval: the cost string is a sympy.lambdify call
# whose body is a polynomial with Float('..', precision=53) literals.
# --------------------------------------------------------------------------------------
def parse_terminal_yaml(text: str):
    """Returns (cost_callable(e9)->float, P 9x9 quadratic part, A 72x9, b 72)."""
    import sympy as sp
    import yaml

    doc = yaml.safe_load(text)
    src = doc["cost"]
    m = re.match(r"\s*sp\.lambdify\(\((.*?)\),\s*(.*),\s*modules=.*\)\s*$", src, re.S)
    if m is None:
        m = re.match(r"\s*sp\.lambdify\(\((.*?)\),\s*(.*)\)\s*$", src, re.S)
    names = [s.strip() for s in m.group(1).split(",")]
    body = m.group(2)
    # the lambdify call may carry a trailing `modules=` kwarg; cut at the last top-level comma
    depth = 0
    cut = None
    for i, ch in enumerate(body):
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "," and depth == 0:
            cut = i
    if cut is not None and "modules" in body[cut:]:
        body = body[:cut]
    syms = sp.symbols(names)
    expr = sp.sympify(body, locals={**{n: s for n, s in zip(names, syms)}, "Float": sp.Float,
                                    "Abs": sp.Abs, "tanh": sp.tanh, "sqrt": sp.sqrt})
    fn = sp.lambdify(syms, expr, "math")
    # quadratic part: exact second-order polynomial coefficients (the (.+1e-6)^0.25 terms
    # are not polynomial and are excluded by construction)
    poly_terms = [t for t in sp.Add.make_args(sp.expand(expr)) if t.is_polynomial(*syms)]
    P = np.zeros((9, 9))
    for t in poly_terms:
        pt = sp.Poly(t, *syms)
        (mon, coeff), = pt.terms()
        if sum(mon) != 2:
            continue
        idx = [i for i, e in enumerate(mon) for _ in range(e)]
        i, j = idx
        if i == j:
            P[i, i] += float(coeff)
        else:
            P[i, j] += float(coeff) / 2
            P[j, i] += float(coeff) / 2
    ts = json.loads(doc["term_set"])
    A = np.array(ts["A"], dtype=float)
    b = np.array(ts["b"], dtype=float).reshape(-1)
    return (lambda e: float(fn(*[float(v) for v in e]))), P, A, b


def terminal_P_quadratic():
    """The quadratic part of terminal.yaml's cost as literal numbers (SURVEY.md section 8
    'Pinned physics'); tests check parse_terminal_yaml(reference file) reproduces it."""
    P = np.zeros((9, 9))
    pp, pv, vv = 19.574136382485836, 28.1433488118291, 98.382994426126402
    for a in range(3):
        P[a, a] = pp
        P[a, 3 + a] = P[3 + a, a] = pv
        P[3 + a, 3 + a] = vv
    P[6, 6], P[7, 7], P[8, 8] = 645.23036107820451, 645.43124119008723, 645.70261416462426
    return P


# --------------------------------------------------------------------------------------
# input hull (controllers/tools/input_bounds.py:43-76) -- only used to verify that the
# thruster-space box is equivalent to the reference's generalized-force hull
# --------------------------------------------------------------------------------------
def input_hull(D, stuck, ub):
    from itertools import product
    from scipy.spatial import ConvexHull

    lohi = [([s, s] if b <= 0 else [0.0, b]) for s, b in zip(stuck, ub)]
    V = np.unique(np.array([D @ np.array(c) for c in product(*lohi)]), axis=0)
    eq = np.unique(ConvexHull(V).equations, axis=0)
    return eq[:, :-1], -eq[:, -1], V
