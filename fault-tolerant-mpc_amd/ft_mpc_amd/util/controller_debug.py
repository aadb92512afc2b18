"""Thin mirror of ft_mpc/util/controller_debug.py: per-step records and the print logger.
Plotting, animation and the 67-column CSV export are out of scope."""
import numpy as np


class Logger:
    def __init__(self, quiet=False):
        self.quiet = quiet

    def info(self, msg):
        if not self.quiet:
            print(msg)

    def warn(self, msg):
        print(f"WARNING: {msg}")


class DebugVal:
    """One controller step (controller_debug.py:9-79): state, centre state, input, D u, errors."""

    def __init__(self, controller, t):
        self.t = t
        self.faulty_force = np.array(controller.model.faulty_force, float).reshape(-1).copy()
        self.state = self.circle_state = self.input = self.gen_input = self.desired = None
        self.pos_err = self.vel_err = self.omega_err = None

    def set_state(self, x):
        self.state = np.asarray(x, float).reshape(-1).copy()

    def set_circle_state(self, c):
        self.circle_state = np.asarray(c, float).reshape(-1).copy()

    def set_input(self, u, model):
        self.input = np.asarray(u, float).reshape(-1).copy()
        self.gen_input = model.D @ (self.input + self.faulty_force)

    def set_desired_state(self, xd):
        self.desired = np.asarray(xd, float).reshape(-1).copy()

    def calculate_errors(self):
        if self.circle_state is not None and self.desired is not None and self.desired.size >= 9:
            self.pos_err = self.circle_state[0:3] - self.desired[0:3]
            self.vel_err = self.circle_state[3:6] - self.desired[3:6]
            self.omega_err = self.circle_state[6:9] - self.desired[6:9]


class ControllerDebug:
    def __init__(self):
        self.history = []

    def add_debug_val(self, val):
        self.history.append(val)

    def states(self):
        return np.array([h.state for h in self.history])

    def inputs(self):
        return np.array([h.input for h in self.history])
