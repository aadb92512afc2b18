"""CPU: the ft_mpc-shaped host mirror (models, reference generator, parameters, simulation loop)
against the oracle restatement and the reference-derived golden pins."""
from pathlib import Path

import numpy as np
import pytest

from ft_mpc_amd.controllers.dummy_controller import Controller
from ft_mpc_amd.controllers.tools.spiral_parameters import SpiralParameters
from ft_mpc_amd.models.spiral_model import SpiralModel
from ft_mpc_amd.models.sys_model import SystemModel, allocation_matrix_8, allocation_matrix_16
from ft_mpc_amd.simulation.sim_env import SimulationEnvironment
from ft_mpc_amd.util.broken_thruster import BrokenThruster
from ft_mpc_amd.util.controller_debug import ControllerDebug
from ft_mpc_amd.util.get_trajectory import load_trajectory
from ft_mpc_amd.util.utils import Rot, RotFull, RotFullInv, RotInv
from oracle import qp_oracle as qo
from oracle import refmath as rm

G = np.load(Path(__file__).parent / "golden" / "reference_pins.npz")


def test_matrices_and_batch_generator_are_bit_identical_to_the_oracle():
    import ft_mpc_amd
    assert np.array_equal(allocation_matrix_16(), rm.allocation_matrix_16())
    assert np.array_equal(allocation_matrix_8(), rm.allocation_matrix_8())
    a = ft_mpc_amd.make_synthetic_batch(32, 20, 8, 2, 1003)
    b = qo.make_batch(32, 20, 8, 2, 1003)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_system_model_fault_bookkeeping_and_dynamics():
    m = SystemModel(0.1)
    assert (m.mass, m.max_thrust, m.Nx, m.Nu) == (16.8, 3.4, 13, 16)
    m.set_fault(BrokenThruster(10, 1.0))
    m.set_fault(BrokenThruster(11, 0.3))
    assert m.faulty_force[10] == 3.4 and m.faulty_force[11] == pytest.approx(1.02) and m.u_ub_physical[10] == 0
    assert np.allclose(m.faulty_force_generalized, m.D @ m.faulty_force)
    rng = np.random.default_rng(0)
    x, u = rng.standard_normal(13), rng.uniform(0, 3.4, 16)
    fs = rm.FaultState(16).set_fault(10, 1.0).set_fault(11, 0.3)
    ref = rm.rk4(lambda s: rm.plant_dx_dt(s, u, rm.allocation_matrix_16(), fs.stuck, fs.ub), x)
    assert np.allclose(m.dynamics(x, u), ref, atol=1e-14)
    assert np.linalg.norm(m.normalize_quaternion(x)[6:10]) == pytest.approx(1.0)


def test_spiral_model_and_parameters():
    m = SystemModel(0.1)
    m.set_fault(BrokenThruster(10, 1.0))
    sm = SpiralModel.from_system_model(m)
    sp = SpiralParameters(m)
    assert sp.r[1] == pytest.approx(0.5787037037037037) and np.array_equal(sp.omega_des, [0, 0, 0.6])
    assert np.allclose(sp.compensation_force, rm.compensation_force(m.D, m.faulty_force))
    assert sm.Nu == 6 and len(sm.broken_thrusters) == 1
    x = np.random.default_rng(1).standard_normal(13)
    assert np.allclose(sm.robot_to_center(x), rm.robot_to_center(x, rm.spiral_r()), atol=1e-15)
    assert np.allclose(sm.center_to_robot(sm.robot_to_center(x)), x, atol=1e-14)
    q = G["ic_quat"]
    assert np.allclose(Rot(q), rm.rot(q)) and np.allclose(RotInv(q), rm.rot(q).T)
    assert np.allclose(RotFull(q) @ RotFullInv(q), np.eye(6), atol=1e-12)


def test_trajectory_generator_against_reference_pins():
    assert np.array_equal(load_trajectory("hover", 0.1, 30)[:, [0, 1, 1500, 2999]], G["hover_cols"])
    assert np.array_equal(load_trajectory("hover_1_2_3", 0.1, 5)[:, 7], G["hover123_col"])
    c = load_trajectory("circle_r_0.65_sPerFullCircle_40", 0.1, 30)
    assert c.shape == tuple(G["circle_shape"]) and np.allclose(c[:, :64], G["circle_cols"], atol=1e-14)
    for bad in ("hoverx", "hover_1_2", "circle_r_2", "circle_x_2_sPerFullCircle_3", "nonsense", "load"):
        with pytest.raises(ValueError):
            load_trajectory(bad, 0.1, 5)


def test_simulation_loop_with_dummy_controller_matches_reference_sim_env():
    """The mirror's loop against the committed capture of the reference's own SimulationEnvironment driving
    the same duck-typed model/controller (tests/golden/sim_env_pin.npz, made by oracle/gen_golden.py:sim_env_fixture
    in the build container; checked on every box, /root/reference is not read here)."""
    pin = np.load(Path(__file__).parent / "golden" / "sim_env_pin.npz")

    def run(dur):
        m = SystemModel(0.1)
        env = SimulationEnvironment(m, Controller(m, ControllerDebug()), seed=0)
        env.set_initial_state(position=[1, 0, 1], velocity=[1, .5, 0], orientation=G["ic_quat"], angular_velocity=[.3, .8, -.1])
        x_init = np.array(env.state, float).reshape(-1).copy()
        for k in env.noise:
            env.noise[k] = 0.0
        env.run_simulation(dur)
        return x_init, np.array(env.state, float).reshape(-1)

    for name, dur in (("1s", 1.0), ("2p5s", 2.5)):
        x_init, mine = run(dur)
        assert np.allclose(x_init, pin["init_" + name], atol=1e-15)
        assert abs(np.linalg.norm(mine[6:10]) - 1) < 1e-12 and mine[5] != 0   # thruster 12 pushes along -z
        assert np.allclose(mine, pin["state_" + name], atol=1e-13)
