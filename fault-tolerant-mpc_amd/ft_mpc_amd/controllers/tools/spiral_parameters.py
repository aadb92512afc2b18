"""Mirror of ft_mpc/controllers/tools/spiral_parameters.py:7-57 (orbit constants from the fault state)."""
import numpy as np


class SpiralParameters:
    def __init__(self, model):
        self.model = model
        self.mass = model.mass
        self.inertia = model.inertia
        self.faulty_force = np.asarray(model.faulty_force, float).flatten()
        self.faulty_force_generalized = np.asarray(model.faulty_force_generalized, float).flatten()
        self.D = model.D
        self.beta = np.array([0.0, 0.0, 0.0, 1.0])     # robot-local == force-aligned frame
        self.calculate_optimal_parameters()

    def calculate_optimal_parameters(self):
        self.omega_des = np.array([0.0, 0.0, 0.6])
        r_dir = np.array([0.0, 1.0, 0.0])
        self.f_virt = 3.5 * r_dir
        self.compensation_force = np.concatenate([self.f_virt, np.zeros(3)]) - self.faulty_force_generalized
        self.r = np.linalg.norm(self.f_virt) / (self.mass * np.linalg.norm(self.omega_des) ** 2) * r_dir
        rr = np.linalg.norm(self.r)
        helper = np.zeros((3, 3))
        helper[0, 2] = -rr / self.inertia[2, 2]
        helper[2, 0] = rr / self.inertia[0, 0]
        self.M = np.block([[np.eye(3) / self.mass, helper], [np.zeros((3, 3)), np.linalg.inv(self.inertia)]])
