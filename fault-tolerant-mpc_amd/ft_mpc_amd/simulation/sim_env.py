"""Mirror of ft_mpc/simulation/sim_env.py: the closed loop around the controller seam
(`u = controller.get_control(state, t)`, sim_env.py:82).  Identical behaviour except that the
measurement noise can be seeded (the reference draws from the global numpy RNG, unseeded)."""
import numpy as np


class SimulationEnvironment:
    def __init__(self, model, controller, seed=None):
        self.model = model
        self.set_controller(controller)
        self.dt = model.dt
        self.noise = {"position": 0.001, "velocity": 0.001, "orientation": 0.001, "angular_velocity": 0.001}
        self.state = np.zeros(model.Nx)
        self.cur_time = 0.0
        self.rng = np.random.default_rng(seed) if seed is not None else np.random

    def set_controller(self, controller):
        self.controller = controller

    def set_initial_state(self, position=None, velocity=None, orientation=None, angular_velocity=None):
        if position is not None:
            self.state[0:3] = np.array(position)
        if velocity is not None:
            self.state[3:6] = np.array(velocity)
        if orientation is not None:
            self.state[6:10] = orientation
        if angular_velocity is not None:
            self.state[10:] = np.array(angular_velocity)

    def step(self):
        u = self.controller.get_control(self.state, self.cur_time)
        x = np.array(self.model.dynamics(self.state, u), float).reshape(-1)
        x[0:3] += self.rng.uniform(0, self.noise["position"], size=3)        # one-sided U(0, 1e-3), as the reference
        x[3:6] += self.rng.uniform(0, self.noise["velocity"], size=3)
        x[6:10] += self.rng.uniform(0, self.noise["orientation"], size=4)
        x[10:] += self.rng.uniform(0, self.noise["angular_velocity"], size=3)
        self.state = np.array(self.model.normalize_quaternion(x))
        self.cur_time += self.dt

    def run_simulation(self, duration):
        for _ in range(int(duration / self.dt)):
            self.step()
