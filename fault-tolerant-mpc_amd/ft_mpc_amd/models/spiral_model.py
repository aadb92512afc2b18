"""Mirror of ft_mpc/models/spiral_model.py: controller-side model of the orbit centre
c = [p_c, v_c, omega, q] with the robot<->centre transforms (spiral_model.py:91-125).  The
centre dynamics themselves run inside the HIP kernels (csrc/ftmpc_linearize.hip)."""
import numpy as np

from .sys_model import SystemModel
from ..controllers.tools.spiral_parameters import SpiralParameters
from ..util.utils import RotInv


class SpiralModel(SystemModel):
    def __init__(self, dt, spiral_params):
        self.r = spiral_params.r
        self.spiral_params = spiral_params
        super().__init__(dt)

    @classmethod
    def from_system_model(cls, sys_model):
        new = cls(sys_model.dt, SpiralParameters(sys_model))
        for bt in sys_model.broken_thrusters:
            new.set_fault(bt)
        return new

    @property
    def Nu(self):
        return self.Nu_simplified

    def robot_to_center(self, x):
        x = np.asarray(x, float).flatten()
        q, w = x[6:10], x[10:13]
        Rt = RotInv(q)
        return np.concatenate([x[0:3] + Rt @ self.r, x[3:6] + Rt @ np.cross(w, self.r), w, q])

    def center_to_robot(self, c):
        """Inverse of robot_to_center (the reference's version, spiral_model.py:111-125, reads an
        undefined variable; this is the transform it documents)."""
        c = np.asarray(c, float).flatten()
        w, q = c[6:9], c[9:13]
        Rt = RotInv(q)
        return np.concatenate([c[0:3] - Rt @ self.r, c[3:6] - Rt @ np.cross(w, self.r), q, w])

    def normalize_quaternion(self, state):
        state = np.array(state, float).reshape(-1)
        state[9:13] /= np.linalg.norm(state[9:13])
        return state
