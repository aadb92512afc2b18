"""Counterpart of the reference's examples/sim.py (its only entry point, BASELINE config 1):
300 closed-loop steps of the 16-thruster vehicle, hover reference, faults from reactive.yaml,
initial condition of examples/sim.py:49-54 -- with the MPC step on the MI355X.

    python examples/sim.py [--nominal] [--steps 300] [--seed 0]
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import yaml
from scipy.spatial.transform import Rotation as R

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fault-tolerant-mpc_amd"))

from ft_mpc_amd.controllers.spiraling_mpc import SpiralingController  # noqa: E402
from ft_mpc_amd.models.spiral_model import SpiralModel  # noqa: E402
from ft_mpc_amd.models.sys_model import SystemModel  # noqa: E402
from ft_mpc_amd.simulation.sim_env import SimulationEnvironment  # noqa: E402
from ft_mpc_amd.util.broken_thruster import BrokenThruster  # noqa: E402
from ft_mpc_amd.util.controller_debug import ControllerDebug  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nominal", action="store_true", help="no actuator failure (BASELINE config 1)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--formulation", default="thruster", choices=["thruster", "wrench"],
                    help="wrench: the reference's own two-stage structure (6-D MPC with the input hull, then allocation)")
    ap.add_argument("--terminal-set", action="store_true", help="add the 72-row terminal set of config/terminal.yaml")
    args = ap.parse_args()
    params = yaml.safe_load(open(ROOT / "fault-tolerant-mpc_amd" / "ft_mpc_amd" / "config" / "reactive.yaml"))
    dt, duration = params["time_step"], params["traj_duration"]
    history = ControllerDebug()
    model = SystemModel(dt)
    if not args.nominal:
        for f in params["actuator_failures"]:
            if f["start_time"] != 0:
                print("WARNING: Actuator failures are not supported yet at times other than 0. Skipping.")
                continue
            model.set_fault(BrokenThruster(f["act_id"], f["intensity"]))
    spiral_model = SpiralModel.from_system_model(model)
    tuning = dict(params["tuning"]["spiraling"], formulation=args.formulation, terminal_set=args.terminal_set)
    controller = SpiralingController(spiral_model, tuning, history, quiet=True)
    controller.load_trajectory(params["traj_shape"], duration)
    env = SimulationEnvironment(model, controller, seed=args.seed)
    env.set_initial_state(position=[1, 0, 1], velocity=[1, 0.5, 0],
                          orientation=R.from_euler("zyx", [50, 30, -10], degrees=True).as_quat(),
                          angular_velocity=[0.3, 0.8, -0.1])
    n = args.steps if args.steps is not None else int(duration / dt)
    for i in range(n):
        env.step()
        if (i + 1) % 50 == 0:
            c = spiral_model.robot_to_center(env.state)
            print(f"step {i + 1:4d}  |p_c| = {np.linalg.norm(c[0:3]):.4f}  |v_c| = {np.linalg.norm(c[3:6]):.4f}  "
                  f"omega = {np.round(c[6:9], 3)}")


if __name__ == "__main__":
    main()
