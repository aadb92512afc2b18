"""Host-side (numpy) mirror of ft_mpc/models/sys_model.py: the 16-thruster plant that the
simulation loop integrates (reference: SystemModel, sys_model.py:31-243).  The MPC solve does
not run here -- it runs in the HIP library -- this is the plant side of the controller seam
(SimulationEnvironment calls model.dynamics(x, u), ft_mpc/simulation/sim_env.py:85).
"""
from __future__ import annotations

import numpy as np


def allocation_matrix_16() -> np.ndarray:
    """6x16 thruster allocation matrix (sys_model.py:73-123), built from the geometry:
    thrusters 0-7 push along -/+x, 8-11 along -/+y, 12-15 along -/+z, with lever arms
    d1=.12, d2=.09, d3=.05."""
    d1, d2, d3 = 0.12, 0.09, 0.05
    D = np.zeros((6, 16))
    D[0, 0:8] = [-1, -1, 1, 1, -1, -1, 1, 1]
    D[4, 0:8] = d3 * np.array([-1, 1, 1, -1, -1, 1, 1, -1.0])
    D[5, 0:8] = d1 * np.array([1, 1, -1, -1, -1, -1, 1, 1.0])
    D[1, 8:12] = [-1, -1, 1, 1]
    D[5, 8:12] = d2 * np.array([-1, 1, 1, -1.0])
    D[2, 12:16] = [-1, 1, -1, 1]
    D[3, 12:16] = d1 * np.array([-1, 1, 1, -1.0])
    return D


def allocation_matrix_8() -> np.ndarray:
    """SYNTHETIC 6x8 matrix of the NT=8 benchmark configs (the reference has
    no 6-DoF 8-thruster vehicle): eight canted unit-thrust thrusters on the body corners
    (+-.15, +-.12, +-.05) m; positively spans R^6."""
    D = np.zeros((6, 8))
    i = 0
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (1, -1):
                p = np.array([sx * 0.15, sy * 0.12, sz * 0.05])
                f1 = 1.0 if (i % 2 == 0) else -1.0
                f2 = 1.0 if ((i // 2) % 2 == 0) else -1.0
                f3 = 1.0 if ((i // 4) % 2 == 0) else -1.0
                f = np.array([-sx * 0.6 * f2, -sy * 0.64 * f1, -sz * 0.48 * f3])
                D[0:3, i] = f
                D[3:6, i] = np.cross(p, f)
                i += 1
    return D


def _omega_q(w, q):
    """1/2 Omega(w) q  (sys_model.py:8-29)."""
    wx, wy, wz = w
    x, y, z, s = q
    return 0.5 * np.array([wz * y - wy * z + wx * s, -wz * x + wx * z + wy * s, wy * x - wx * y + wz * s,
                           -wx * x - wy * y - wz * z])


class SystemModel:
    """The 3-D rigid body with 16 thrusters (sys_model.py:31-247): state [p, v, q(xyzw), omega],
    `dynamics(x, u)` = one RK4 step of `dx_dt`, `set_fault(BrokenThruster)` bookkeeping."""

    def __init__(self, dt):
        self.mass = 16.8
        self.inertia = np.diag([0.2, 0.3, 0.25])
        self.inertia_inv = np.linalg.inv(self.inertia)
        self.max_thrust = 3.4
        self.Nx, self.Nu_simplified, self.Nu_full = 13, 6, 16
        self.dt = dt
        self.D = allocation_matrix_16()
        self.broken_thrusters = []
        self.faulty_force = np.zeros((1, self.Nu_full))
        self.faulty_force_generalized = self.D @ self.faulty_force.flatten()
        self.u_ub_physical = np.full(self.Nu_full, self.max_thrust)

    @property
    def Nu(self):
        return self.Nu_full

    def set_fault(self, broken_thruster):
        self.broken_thrusters.append(broken_thruster)
        self.faulty_force = np.zeros(self.Nu_full)
        self.u_ub_physical = np.full(self.Nu_full, self.max_thrust)
        for bt in self.broken_thrusters:
            self.faulty_force[bt.index] = bt.intensity * self.max_thrust
            self.u_ub_physical[bt.index] = 0.0
        self.faulty_force_generalized = self.D @ self.faulty_force.flatten()

    def dx_dt(self, x, u):
        from ..util.utils import RotInv
        x = np.asarray(x, float).reshape(-1)
        u = np.array(u, float).reshape(-1)
        for bt in self.broken_thrusters:
            u[bt.index] = 0.0
        gen = self.D @ (u + self.faulty_force.reshape(-1))
        v, q, w = x[3:6], x[6:10], x[10:13]
        dw = self.inertia_inv @ (gen[3:6] - np.cross(w, self.inertia @ w))
        return np.concatenate([v, RotInv(q) @ gen[0:3] / self.mass, _omega_q(w, q), dw])

    def dynamics(self, x, u):
        x = np.asarray(x, float).reshape(-1)
        h = self.dt
        k1 = self.dx_dt(x, u)
        k2 = self.dx_dt(x + h / 2 * k1, u)
        k3 = self.dx_dt(x + h / 2 * k2, u)
        k4 = self.dx_dt(x + h * k3, u)
        return x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)

    def normalize_quaternion(self, state):
        state = np.array(state, float).reshape(-1)
        state[6:10] /= np.linalg.norm(state[6:10])
        return state
