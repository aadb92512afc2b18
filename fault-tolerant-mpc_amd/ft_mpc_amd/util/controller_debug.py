"""Mirror of ft_mpc/util/controller_debug.py: per-step records (DebugVal), the history container with
the reference's on-disk format (ControllerDebug.export: 67-column ';'-separated CSV,
controller_debug.py:216-260) and the print logger.  Plotting/animation are out of scope."""
from pathlib import Path

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
    """One controller step (controller_debug.py:9-79).  Errors are desired - actual; force/torque
    are D @ u of the commanded inputs (the stuck-thruster force is not included), as in the reference."""

    def __init__(self, controller, t):
        self.controller = str(controller)
        self.faulty_force = np.array(controller.model.faulty_force, float).copy()
        self.time = t
        for name in ("position", "velocity", "orientation", "angular_velocity", "input", "force", "torque",
                     "circle_position", "circle_velocity", "circle_angular_velocity",
                     "position_error", "velocity_error", "orientation_error", "angular_velocity_error",
                     "circle_position_error", "circle_velocity_error", "circle_angular_velocity_error"):
            setattr(self, name, None)

    def set_state(self, x):
        x = np.asarray(x, float).flatten()
        self.position, self.velocity, self.orientation, self.angular_velocity = x[0:3], x[3:6], x[6:10], x[10:13]

    def set_circle_state(self, c):
        c = np.asarray(c, float).flatten()
        self.circle_position, self.circle_velocity, self.circle_angular_velocity = c[0:3], c[3:6], c[6:9]

    def set_input(self, u, model):
        u = np.asarray(u, float).flatten()
        self.input = u
        gen = model.D @ u
        self.force, self.torque = gen[0:3], gen[3:6]

    def set_desired_state(self, x):
        x = np.asarray(x, float).flatten()
        self.desired_position, self.desired_velocity = x[0:3], x[3:6]
        if x.size == 9:
            self.desired_angular_velocity, self.desired_orientation = x[6:9], np.zeros(4)
        else:
            self.desired_orientation, self.desired_angular_velocity = x[6:10], x[10:13]

    def calculate_errors(self):
        if self.position is not None:
            self.position_error = self.desired_position - self.position
            self.velocity_error = self.desired_velocity - self.velocity
            self.orientation_error = self.desired_orientation - self.orientation
            self.angular_velocity_error = self.desired_angular_velocity - self.angular_velocity
        if self.circle_position is not None:
            self.circle_position_error = self.desired_position - self.circle_position
            self.circle_velocity_error = self.desired_velocity - self.circle_velocity
            self.circle_angular_velocity_error = self.desired_angular_velocity - self.circle_angular_velocity


# column groups of the export, in file order: (attribute, component suffixes)
_XYZ, _XYZW = ("x", "y", "z"), ("x", "y", "z", "w")
_GROUPS = [("position", _XYZ), ("velocity", _XYZ), ("orientation", _XYZW), ("angular_velocity", _XYZ),
           ("input", tuple(str(i) for i in range(16))), ("force", _XYZ), ("torque", _XYZ),
           ("circle_position", _XYZ), ("circle_velocity", _XYZ), ("circle_angular_velocity", _XYZ),
           ("position_error", _XYZ), ("velocity_error", _XYZ), ("orientation_error", _XYZW),
           ("angular_velocity_error", _XYZ), ("circle_position_error", _XYZ), ("circle_velocity_error", _XYZ),
           ("circle_angular_velocity_error", _XYZ)]


class ControllerDebug:
    def __init__(self):
        self.history = []

    def add_debug_val(self, val):
        self.history.append(val)

    def get_time(self):
        return np.array([h.time for h in self.history])

    def states(self):
        return np.array([np.concatenate([h.position, h.velocity, h.orientation, h.angular_velocity]) for h in self.history])

    def inputs(self):
        return np.array([h.input for h in self.history])

    @staticmethod
    def header():
        return ["time"] + [f"{name}_{c}" for name, comps in _GROUPS for c in comps]

    def export(self, file_path=None):
        """Writes `<file_path>.csv` in the reference's format (one row per controller step)."""
        if file_path is None:
            file_path = str(Path.cwd() / "debug_data")
        rows = [np.concatenate([[h.time]] + [np.asarray(getattr(h, name), float).flatten() for name, _ in _GROUPS])
                for h in self.history]
        data = np.array(rows) if rows else np.zeros((0, len(self.header())))
        np.savetxt(file_path + ".csv", data, delimiter=";", header=";".join(self.header()))
        return file_path + ".csv"
