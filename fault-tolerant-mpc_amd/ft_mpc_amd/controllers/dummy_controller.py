"""Mirror of ft_mpc/controllers/dummy_controller.py: constant-input controller documenting the
duck-typed controller seam (`get_control(state, time) -> ndarray[Nu]`)."""
import numpy as np

from ..util.controller_debug import DebugVal


class Controller:
    def __init__(self, model, history):
        self.model, self.Nx, self.Nu, self.history = model, model.Nx, model.Nu, history

    def get_control(self, state, time):
        u = np.zeros(self.Nu)
        u[12] = 1.0
        d = DebugVal(self, time)
        d.set_state(state)
        d.set_input(u, self.model)
        d.set_desired_state(np.zeros(self.Nx))
        d.calculate_errors()
        self.history.add_debug_val(d)
        return u
