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
