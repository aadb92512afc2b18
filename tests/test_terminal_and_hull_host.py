"""CPU: the product-side loaders of the generalized-force / terminal-set path and the oracle that checks their kernels.
  * controllers/tools/terminal_ingredients.py (no eval, no sympy) against the reference-derived pins of
    tests/golden/reference_pins.npz (made by oracle/gen_golden.py with sympy on the reference's terminal.yaml)
  * controllers/tools/input_bounds.py (zonotope H-rep) against the reference's recipe restated in
    oracle/refmath.py:input_hull (corner enumeration + Qhull, input_bounds.py:43-76)
  * oracle/qp_oracle.py:ipm_general against an LP feasibility certificate, SLSQP and the thruster-space BVLS solution
"""
from pathlib import Path

import numpy as np
import pytest

from ft_mpc_amd.controllers.tools import terminal_ingredients as ti
from ft_mpc_amd.controllers.tools.input_bounds import InputBounds, hull_tables, zonotope_hrep
from ft_mpc_amd.models.sys_model import SystemModel
from ft_mpc_amd.util.broken_thruster import BrokenThruster
from oracle import qp_oracle as qo
from oracle import refmath as rm

G = np.load(Path(__file__).parent / "golden" / "reference_pins.npz")


def _rows(A, b):
    M = np.hstack([A, np.asarray(b).reshape(-1, 1)])
    return M[np.lexsort(np.round(M, 6).T[::-1])]


def test_terminal_yaml_loader_matches_the_reference_pins():
    cost, tset = ti.load_terminal_ingredients()          # same return contract as the reference loader
    t = ti.load_terminal()
    assert np.array_equal(t.P, G["term_P"]) and np.array_equal(tset.A, G["term_A"])
    assert np.array_equal(tset.b.reshape(-1), G["term_b"]) and tset.b.shape == (72, 1) and tset.Nc == 72
    for p, c in zip(G["term_points"], G["term_cost"]):
        assert cost(*p) == pytest.approx(c, abs=1e-11)     # called like the reference's lambdified cost
    assert t.cost(np.zeros(9)) == 0.0 and np.abs(t.grad(np.zeros(9))).max() < 1e-12
    rng = np.random.default_rng(3)
    for _ in range(5):
        e = rng.uniform(-0.3, 0.3, 9)
        fd = np.array([(t.cost(e + 1e-6 * np.eye(9)[i]) - t.cost(e - 1e-6 * np.eye(9)[i])) / 2e-6 for i in range(9)])
        assert np.abs(t.grad(e) - fd).max() <= 1e-8 * max(1.0, np.abs(fd).max())
        assert t.cost(e) == pytest.approx(e @ t.P @ e + t.cost(e, quadratic=False), rel=1e-13)
    tab = t.device_tables()
    assert len(tab["root_coef"]) == 12 and (tab["root_pow"] == 0.25).all() and (tab["root_eps"] == 1e-6).all()
    assert tset.contains(np.zeros(9)) and not tset.contains(np.full(9, 0.5))


@pytest.mark.parametrize("bad", ["sp.lambdify((ep1, ep2, ep3, ev1, ev2, ev3, eo1, eo2, eo3), __import__('os').system('true'))",
                                 "sp.lambdify((ep1, ep2, ep3, ev1, ev2, ev3, eo1, eo2, eo3), ep1**ep2)",
                                 "sp.lambdify((a, b), a*b)", "lambda e: 0"])
def test_terminal_cost_parser_refuses_anything_outside_its_grammar(bad):
    with pytest.raises(ValueError):
        ti.parse_terminal_cost(bad)


@pytest.mark.parametrize("faults", [[], [(0, 1.0)], [(10, 1.0), (11, 1.0)], [(0, 0.5), (1, 0.5)], [(12, 0.0)]])
def test_zonotope_rows_are_the_reference_hull(faults):
    """The five fault patterns SURVEY.md section 8(c) lists: 26 facets each, the same rows as corner enumeration + Qhull."""
    D = rm.allocation_matrix_16()
    ub, stuck = np.full(16, rm.F_MAX), np.zeros(16)
    m = SystemModel(0.1)
    for i, a in faults:
        ub[i], stuck[i] = 0.0, a * rm.F_MAX
        m.set_fault(BrokenThruster(i, a))
    A, b = zonotope_hrep(D, ub, stuck)
    Aq, bq, V = rm.input_hull(D, stuck, ub)
    assert A.shape == (26, 6) and np.abs(_rows(A, b) - _rows(Aq, bq)).max() < 1e-12
    assert (V @ A.T <= b + 1e-9).all()                             # every corner inside
    Am, bm = InputBounds(m).get_conv_hull()                        # the mirror class on the mirror model
    assert np.array_equal(Am, A) and np.allclose(bm, b, atol=1e-13)


def test_hull_tables_for_a_batch():
    D = rm.allocation_matrix_16()
    _, ub, stuck, _ = qo.make_batch(600, 20, 16, 2, 5)
    T = hull_tables(D, ub, stuck)
    assert T["rows"] == 26 and T["A"].shape[1:] == (26, 6) and T["A"].shape[0] <= 118
    flat = T["degenerate"]
    pairs = {tuple(np.flatnonzero(u == 0)) for u in ub[flat]}
    assert pairs <= {(12, 13), (14, 15)}                            # the only flat hulls of the reference vehicle
    for b in np.flatnonzero(~flat)[:20]:
        A, bb = zonotope_hrep(D, ub[b], stuck[b])
        assert np.allclose(T["A"][T["set"][b]], A, atol=1e-13) and np.allclose(T["b"][b], bb, atol=1e-12)
    with pytest.raises(ValueError):
        zonotope_hrep(D, ub[np.flatnonzero(flat)[0]], stuck[np.flatnonzero(flat)[0]])


def test_general_ipm_against_independent_solvers():
    from scipy.optimize import minimize
    cfg = qo.QPConfig(N=6, NT=16)
    x0, ub, stuck, xref = qo.make_batch(6, 6, 16, 2, 9)
    cfg0 = qo.QPConfig(N=6, NT=16, rho=1e-9)
    for b in range(6):
        try:
            tau0, T, st, nit, qp = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref)
        except ValueError:
            continue
        assert st == 0 and max(qo.kkt_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d"], qp["z"])) < 1e-5
        # (1) thruster-space BVLS without allocation weight: same optimal wrenches
        _, U, _ = qo.solve_instance(cfg0, x0[b], ub[b], stuck[b], xref, exact=True)
        assert np.abs(T - (U + stuck[b]) @ cfg.D.T).max() < 3e-5
        # (2) SLSQP on the same 36-variable QP
        f = lambda d: 0.5 * d @ qp["H"] @ d + qp["g"] @ d
        r = minimize(f, qp["d0"], jac=lambda d: qp["H"] @ d + qp["g"], method="SLSQP",
                     constraints=[dict(type="ineq", fun=lambda d: qp["h"] - qp["C"] @ d, jac=lambda d: -qp["C"])],
                     options=dict(maxiter=500, ftol=1e-14))
        assert abs(f(r.x) - f(qp["d"])) < 1e-6 * max(1.0, abs(f(qp["d"])))
