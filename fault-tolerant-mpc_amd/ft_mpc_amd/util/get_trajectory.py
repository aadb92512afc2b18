"""Mirror of ft_mpc/util/get_trajectory.py:43-184: reference generator returning a
13 x (10*duration/dt) array [p, v, q(x,y,z,w), omega].  Same command strings and the same
ValueErrors for malformed commands (get_trajectory.py:140,152-182)."""
import numpy as np
import yaml


def _grid(duration, dt):
    return np.arange(0, 10 * duration, dt)


def _static(t, position=(0.0, 0.0, 0.0)):
    x = np.zeros((13, t.size))
    x[0:3] = np.asarray(position, float).reshape(3, 1)
    x[9] = 1.0                               # identity quaternion [0,0,0,1]
    return x


def _line(t):
    x = _static(t)
    x[0] = t
    x[3] = 1.0
    return x


def _circle(t, radius=2.0, s_per_circle=30.0):
    om = 2 * np.pi / s_per_circle
    x = _static(t)
    x[0] = radius * np.cos(om * t) - radius
    x[1] = radius * np.sin(om * t)
    x[3] = -radius * om * np.sin(om * t)
    x[4] = radius * om * np.cos(om * t)
    return x


def _sin(t, dt):
    from scipy.spatial.transform import Rotation
    x = np.zeros((13, t.size))
    x[0], x[1], x[3], x[4] = 0.1 * np.sin(t), t, 0.1 * np.cos(t), 1.0
    x[6:10] = Rotation.from_euler("xyz", [np.pi / 2, 0, 0]).as_quat().reshape(4, 1)
    return x                                  # constant attitude: zero angular velocity


def load_trajectory(action, dt, duration=100, file_path=None):
    t = _grid(duration, dt)
    if action == "generate_sin":
        return _sin(t, dt)
    if action == "generate_line":
        return _line(t)
    if action in ("generate_point_stabilizing", "hover"):
        return _static(t)
    if "hover" in action:
        name, *par = action.split("_")
        if name != "hover":
            raise ValueError(f"Invalid action '{action}'.")
        if len(par) != 3:
            raise ValueError(f"Invalid number of parameters for action '{action}'. Use 'hover' or 'hover_<x>_<y>_<alpha>'")
        return _static(t, [float(p) for p in par])
    if action == "generate_circle":
        return _circle(t)
    if "circle" in action:
        name, *par = action.split("_")
        if name != "circle":
            raise ValueError(f"Invalid action '{action}'.")
        if len(par) != 4 or par[0] != "r" or par[2] != "sPerFullCircle":
            raise ValueError(f"Invalid parameters for action '{action}'. Use 'circle_r_<radius>_sPerFullCircle_<speed>'")
        return _circle(t, float(par[1]), float(par[3]))
    if action == "load":
        if file_path is None:
            raise ValueError(f"Invalid parameters for action '{action}'. Use with 'file_path=<path>'.")
        with open(file_path) as f:
            data = yaml.safe_load(f)
        if data["dt"] != dt:
            raise ValueError(f"Trajectory ({data['dt']}s) and controller ({dt}s) have different time steps.")
        traj = np.array(data["x"], float).T
        if duration is not None and traj.shape[1] < duration / dt:
            raise ValueError(f"Trajectory is too short: {traj.shape[1] * dt}s, but {duration}s needed.")
        return traj
    raise ValueError(f"Invalid action '{action}'.")
