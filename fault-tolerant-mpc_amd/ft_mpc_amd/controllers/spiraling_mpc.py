"""Mirror of ft_mpc/controllers/spiraling_mpc.py: `SpiralingController` with the reference's
constructor and method signatures, whose per-step solve runs on the MI355X through the C-ABI
(ft_mpc_amd.batch.BatchedMPC -> libftmpc_hip.so) instead of CasADi/IPOPT + cvxpy.

Differences a caller can observe (DESIGN.md QP-spec): the solve is one condensed thruster-space
QP per step (linearised about the shifted previous solution), so `solve_mpc` returns 16-D thruster
force stages instead of 6-D generalized deviation inputs, and `status` is an IPM status string.

`params["formulation"] = "wrench"` switches to the reference's own two-stage structure: a 6-D generalized-force QP
with the input hull (tools/input_bounds.py, built once per fault set like the reference's InputBounds) followed by the
min-norm allocation; `params["terminal_set"] = True` adds the 72-row terminal set of config/terminal.yaml;
`params["xub"]` / `params["xlb"]` (13 values, the reference's own optional keys) bound the orbit-centre state of the stages.
"""
import copy
import time

import numpy as np

from ..batch import BatchedMPC, MPCConfig
from ..util.controller_debug import DebugVal, Logger
from ..util.get_trajectory import load_trajectory
from .tools.spiral_parameters import SpiralParameters

_STATUS = {0: "Solve_Succeeded", 1: "Maximum_Iterations_Exceeded", 2: "Numerical_Failure"}


class SpiralingController:
    def __init__(self, model, params, debug, device_id=0, dtype="f64", quiet=False):
        self.params = params
        self.debug = debug
        self.logger = Logger(quiet)
        self.spiral_params = SpiralParameters(model)
        self.model = copy.deepcopy(model)          # spiraling_mpc.py:47
        self.mass, self.J, self.dt = model.mass, model.inertia, model.dt
        self.Nx, self.Nu, self.Nopt = model.Nx, model.Nu, 9
        self.Nt = int(params["horizon"])
        pset = params[params["param_set"]]
        self.Q, self.R = np.diag(pset["Q"]).astype(float), np.diag(pset["R"]).astype(float)
        self.u_comp = self.spiral_params.compensation_force
        self.formulation = str(params.get("formulation", "thruster"))
        if self.formulation not in ("thruster", "wrench"):
            raise ValueError("params['formulation'] must be 'thruster' or 'wrench'")
        t0 = time.time()
        self.mpc = BatchedMPC(MPCConfig(N=self.Nt, NT=self.model.Nu_full, dt=self.dt, mass=self.mass, J=self.J,
                                        D=self.model.D, Q=np.array(pset["Q"], float), R=np.array(pset["R"], float),
                                        r=self.spiral_params.r, f_virt=self.spiral_params.f_virt,
                                        rho=float(params.get("rho", 0.05)), device_id=device_id, dtype=dtype,
                                        max_iters=int(params.get("max_iters", 0)),
                                        terminal_set=True if params.get("terminal_set") else None,
                                        # Bounds on x, default is None (spiraling_mpc.py:129-130; rows :179-185)
                                        xub=params.get("xub", None), xlb=params.get("xlb", None)))
        if (params.get("xub") is not None or params.get("xlb") is not None) and self.formulation == "wrench":
            raise ValueError("state bounds (params 'xub' / 'xlb') are built for formulation='thruster'")
        self.hull = None
        if self.formulation == "wrench":           # spiraling_mpc.py:49: self.bounds = InputBounds(self.model), once
            from .tools.input_bounds import hull_tables
            self.hull = hull_tables(self.model.D, np.asarray(self.model.u_ub_physical, float).reshape(1, -1),
                                    np.asarray(self.model.faulty_force, float).reshape(1, -1))
            if self.hull["degenerate"][0]:
                raise ValueError("the healthy thrusters do not span R^6: no input hull (use formulation='thruster')")
        self.optimal_wrench = None                 # warm start of the wrench formulation: previous tau*, [N, 6]
        self.logger.info(f"# Time to build mpc solver: {time.time() - t0} sec")
        self.logger.info(f"# Number of variables: {self.Nt * self.model.Nu_full} (thruster space, condensed)")
        self.trajectory = None
        self.nominal_input = None
        self.optimal_solution = None               # warm start: previous U*, [N, NT]
        self.x_sp = self.u_sp = None

    # -- reference trajectory (spiraling_mpc.py:240-286, 356-365) --------------------------
    def load_trajectory(self, cmd, duration, fpath=None):
        self.assign_trajectory(load_trajectory(cmd, self.dt, duration, file_path=fpath))

    def assign_trajectory(self, traj):
        ext = np.hstack([traj, np.tile(traj[:, -1:], (1, self.Nt))])
        om = np.tile(self.spiral_params.omega_des.reshape(3, 1), (1, ext.shape[1]))
        self.trajectory = np.vstack([ext[0:6], om])
        acc = np.gradient(np.gradient(self.trajectory[0:3], axis=1), axis=1) / self.dt ** 2
        self.nominal_input = np.vstack([acc * self.mass, np.zeros_like(acc)])

    def get_next_trajectory_part(self, t):
        s = int(t / self.dt)
        return self.trajectory[:, s:s + self.Nt + 1], self.nominal_input[:, s:s + self.Nt + 1]

    # -- one MPC step ------------------------------------------------------------------------
    def _solve(self, x0):
        ub = np.asarray(self.model.u_ub_physical, float).reshape(1, -1)
        stuck = np.asarray(self.model.faulty_force, float).reshape(1, -1)
        if self.formulation == "wrench":
            warm = None
            if self.optimal_wrench is not None:    # shifted like the reference's warm start; the last stage repeats
                warm = np.vstack([self.optimal_wrench[1:], self.optimal_wrench[-1:]])[None].copy()
            out = self.mpc.solve_wrench(np.asarray(x0, float).reshape(1, 13), ub, stuck, self.x_sp.reshape(-1),
                                        uref=self.u_sp.reshape(-1), warmG=warm, return_G=True, hull=self.hull)
            self.optimal_wrench = out["G"][0]
            out["U"] = None
            return out
        warm = None
        if self.optimal_solution is not None:      # shift by one stage (spiraling_mpc.py:324-334)
            warm = np.vstack([self.optimal_solution[1:], np.zeros((1, self.model.Nu_full))])[None].copy()
        out = self.mpc.solve(np.asarray(x0, float).reshape(1, 13), ub, stuck, self.x_sp.reshape(-1),
                             uref=self.u_sp.reshape(-1), warmU=warm, return_U=True,
                             relinearize=int(self.params.get("sqp_iters", 1)) - 1)
        self.optimal_solution = out["U"][0]
        return out

    def get_control(self, x0, t):
        x0 = np.asarray(x0, float).flatten()
        c0 = self.model.robot_to_center(x0) if hasattr(self.model, "robot_to_center") else None
        x_ref, u_ref = self.get_next_trajectory_part(t)
        self.x_sp = x_ref.reshape(-1, 1, order="F")
        self.u_sp = u_ref.reshape(-1, 1, order="F")
        t0 = time.time()
        out = self._solve(x0)
        self.logger.info(f"MPC - GPU time: {time.time() - t0:,.7f} seconds  |  Horizon length: {self.Nt}  |  "
                         f"{_STATUS[int(out['status'][0])]} ({int(out['iters'][0])} it)")   # never raised, :347-352
        u_phys = out["u0"][0].copy()
        if self.debug is not None:
            dbg = DebugVal(self, t)
            dbg.set_state(x0)
            if c0 is not None:
                dbg.set_circle_state(c0)
            dbg.set_input(u_phys, self.model)
            dbg.set_desired_state(self.x_sp[0:self.Nopt, 0])
            dbg.calculate_errors()
            self.debug.add_debug_val(dbg)
        return u_phys

    def solve_mpc(self, c0):
        """Reference signature (spiraling_mpc.py:319-354): centre state in,
        (x_list, u_list, solve_time, cost, status) out.  u_list holds thruster-force stages."""
        if self.x_sp is None:
            x_ref, u_ref = self.get_next_trajectory_part(0.0)
            self.x_sp = x_ref.reshape(-1, 1, order="F")
            self.u_sp = u_ref.reshape(-1, 1, order="F")
        t0 = time.time()
        x0 = self.model.center_to_robot(np.asarray(c0, float).flatten())
        out = self._solve(x0)
        if self.formulation == "wrench":
            raise NotImplementedError("solve_mpc returns thruster stages; with formulation='wrench' use get_control / mpc.solve_wrench")
        U = out["U"][0]
        xs = [np.asarray(c0, float).flatten()]
        x = x0
        for k in range(self.Nt):                   # predicted centre states under U* (plant model)
            x = self.model.dynamics(x, U[k])
            xs.append(self.model.robot_to_center(x))
        xr = self.x_sp.reshape(self.Nopt, -1, order="F")
        cost = 0.0
        fv = np.concatenate([self.spiral_params.f_virt, np.zeros(3)])
        for k in range(self.Nt):
            e = xs[k][0:9] - xr[:, k]
            ut = self.model.D @ (U[k] + self.model.faulty_force.reshape(-1)) - fv
            cost += float(e @ self.Q @ e + ut @ self.R @ ut)
        return xs, [U[k] for k in range(self.Nt)], time.time() - t0, cost, _STATUS[int(out["status"][0])]
