"""CPU: the oracle's physics restatement against the reference-derived golden pins
(tests/golden/reference_pins.npz, made by oracle/gen_golden.py from the reference's importable
modules and data files) and against first-principles properties."""
from pathlib import Path

import numpy as np
import pytest

from oracle import refmath as rm

G = np.load(Path(__file__).parent / "golden" / "reference_pins.npz")


def test_allocation_matrix_three_ways():
    D = rm.allocation_matrix_16()
    assert np.array_equal(D, G["D_md"])          # data/InertialProperties.md:30-41
    assert np.allclose(D, G["D_geom"], atol=1e-15)  # r x F from util/animate.py:66-110


def test_yaml_constants():
    assert G["yaml_dt"][0] == rm.DT
    assert np.array_equal(G["yaml_Q"], rm.Q_DIAG)
    assert np.array_equal(G["yaml_R"], rm.R_DIAG)
    assert G["yaml_horizon"][0] == 15
    assert np.array_equal(G["yaml_faults"], [[10, 1.0, 0], [11, 1.0, 0]])


def test_spiral_constants():
    r = rm.spiral_r()
    assert r[1] == pytest.approx(0.5787037037037037, abs=1e-16) and r[0] == 0 and r[2] == 0
    fs = rm.FaultState(16).set_fault(10, 1.0).set_fault(11, 1.0)
    assert fs.stuck[10] == 3.4 and fs.ub[10] == 0 and fs.ub[0] == 3.4
    comp = rm.compensation_force(rm.allocation_matrix_16(), fs.stuck)
    assert np.allclose(comp, [0, 3.5 - 2 * 3.4, 0, 0, 0, 0], atol=1e-12)  # thrusters 10,11 push +y, torques cancel


def test_trajectories_match_reference_loader():
    hov = rm.hover_trajectory(0.1, 30)
    assert tuple(G["hover_shape"]) == hov.shape
    assert np.array_equal(hov[:, [0, 1, 1500, 2999]], G["hover_cols"])
    h = rm.hover_trajectory(0.1, 5, (1, 2, 3))
    assert tuple(G["hover123_shape"]) == h.shape and np.array_equal(h[:, 7], G["hover123_col"])
    c = rm.circle_trajectory(0.1, 30, 0.65, 40.0)
    assert tuple(G["circle_shape"]) == c.shape
    assert np.allclose(c[:, :64], G["circle_cols"], atol=1e-14)
    assert np.allclose([c.sum(), np.abs(c).sum()], G["circle_sum"], rtol=1e-13)


def test_terminal_ingredients():
    assert np.array_equal(rm.terminal_P_quadratic(), G["term_P"])
    assert G["term_A"].shape == (72, 9) and G["term_b"].shape == (72,)
    assert np.allclose(G["term_b"][-6:], 0.18290065200822866)
    assert abs(G["term_cost"][0]) < 1e-9
    assert G["term_cost"][1] == pytest.approx(82.58493648884095, rel=1e-12)
    assert G["term_cost"][2] == pytest.approx(63.3621948305597, rel=1e-12)
    # the quadratic part accounts for the polynomial second-order terms of the cost
    P = G["term_P"]
    assert np.allclose(P, P.T) and np.linalg.eigvalsh(P).min() > 0


def test_initial_quaternion_and_rotation_convention():
    q = G["ic_quat"]
    assert np.allclose(q, [0.03266701292872763, 0.26925564114813405, 0.3862204035220014, 0.8816280768439285])
    from scipy.spatial.transform import Rotation
    # utils.py:15-19 is the world->body matrix: transpose of scipy's body->world
    assert np.allclose(rm.rot(q), Rotation.from_quat(q).as_matrix().T, atol=1e-14)
    assert np.allclose(rm.rot_full(q)[:3, :3], rm.rot(q)) and np.allclose(rm.rot_full(q)[3:, 3:], np.eye(3))


def test_omega_operator_is_quaternion_kinematics():
    rng = np.random.default_rng(0)
    w, q = rng.standard_normal(3), rng.standard_normal(4)
    O = rm.omega_op(w)
    assert np.allclose(O, -O.T)
    # q (x) [w,0] in [x,y,z,w] order
    x, y, z, s = q
    prod = np.array([s * w[0] + y * w[2] - z * w[1], s * w[1] + z * w[0] - x * w[2], s * w[2] + x * w[1] - y * w[0],
                     -(x * w[0] + y * w[1] + z * w[2])])
    assert np.allclose(O @ q, prod)


def test_rk4_invariants():
    r = rm.spiral_r()
    D = rm.allocation_matrix_16()
    x = np.array([1, 0, 1, 1, .5, 0, *G["ic_quat"], .3, .8, -.1])
    xn = rm.rk4(lambda s: rm.plant_dx_dt(s, np.zeros(16), D, np.zeros(16), np.full(16, 3.4)), x)
    assert np.allclose(xn[3:6], x[3:6])                  # zero force: constant velocity
    assert np.allclose(xn[0:3], x[0:3] + 0.1 * x[3:6])
    assert abs(np.linalg.norm(xn[6:10]) - 1) < 1e-7     # O(dt^5) norm drift
    # centre model == plant observed at the orbit centre (same wrench)
    u = np.random.default_rng(1).uniform(0, 3.4, 16)
    gen = D @ u
    cn = rm.rk4(lambda s: rm.centre_dx_dt(s, gen, r), rm.robot_to_center(x, r))
    xn = rm.rk4(lambda s: rm.plant_dx_dt(s, u, D, np.zeros(16), np.full(16, 3.4)), x)
    assert np.allclose(cn, rm.robot_to_center(xn, r), atol=2e-6)  # both are O(dt^5) accurate


def test_jacobians_against_finite_differences():
    rng = np.random.default_rng(3)
    r = rm.spiral_r()
    for _ in range(3):
        c = rng.standard_normal(13)
        gen = rng.standard_normal(6)
        nxt, A, Bg = rm.rk4_with_jac(c, gen, r)
        f = lambda s, g: rm.rk4(lambda z: rm.centre_dx_dt(z, g, r), s)
        h = 1e-6
        Afd = np.column_stack([(f(c + h * e, gen) - f(c - h * e, gen)) / (2 * h) for e in np.eye(13)])
        Bfd = np.column_stack([(f(c, gen + h * e) - f(c, gen - h * e)) / (2 * h) for e in np.eye(6)])
        assert np.abs(A - Afd).max() < 1e-8 and np.abs(Bg - Bfd).max() < 1e-8
        # block structure the kernels rely on (csrc/ftmpc_common.h)
        assert np.allclose(A[:, 0:3], np.eye(13)[:, 0:3]) and np.allclose(A[0:3, 3:6], 0.1 * np.eye(3))
        assert np.allclose(A[3:6, 3:6], np.eye(3)) and np.allclose(A[6:13, 3:6], 0) and np.allclose(A[6:9, 9:13], 0)
        assert np.allclose(Bg[6:13, 0:3], 0)


def test_synthetic_nt8_matrix_positively_spans():
    from scipy.optimize import linprog
    D = rm.allocation_matrix_8()
    assert np.linalg.matrix_rank(D) == 6
    res = linprog(np.ones(8), A_eq=D, b_eq=np.zeros(6), bounds=[(1, None)] * 8)
    assert res.status == 0
    assert np.allclose(np.linalg.norm(D[:3], axis=0), 1.0)


@pytest.mark.parametrize("faults", [[], [(0, 1.0)], [(10, 1.0), (11, 1.0)]])
def test_hull_has_26_facets_and_box_equivalence(faults):
    """input_bounds.py:43-76: the generalized-force hull the reference constrains u to is the image
    of the thruster box, so box constraints in thruster space are equivalent (QP-spec)."""
    D = rm.allocation_matrix_16()
    fs = rm.FaultState(16)
    for i, a in faults:
        fs.set_fault(i, a)
    A, b, V = rm.input_hull(D, fs.stuck, fs.ub)
    assert A.shape == (26, 6)
    u = np.random.default_rng(5).uniform(0, 1, (200, 16)) * fs.ub + fs.stuck
    assert (A @ (D @ u.T) <= b[:, None] + 1e-9).all()
