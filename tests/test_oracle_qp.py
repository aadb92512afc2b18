"""CPU: the QP-spec oracle (numpy restatement, C port) against an independent exact solver
(scipy BVLS) and against the committed golden QP fixtures."""
from pathlib import Path

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import qp_oracle as qo
from oracle import refmath as rm

GOLD = Path(__file__).parent / "golden"


def _load(name):
    d = np.load(GOLD / f"qp_{name}.npz")
    ur = d["uref"] if d["uref"].size else None
    W = d["warm"] if d["warm"].size else None
    return d, ur, W


@pytest.mark.parametrize("name", ["cfg2_single_fault", "cfg3_double_fault", "cfg3_warm_uref", "nominal_nt8", "short_horizon",
                                  "refvehicle_n15", "refvehicle_n20", "cfg5_n40_nt16"])
def test_c_oracle_reproduces_golden(name):
    d, ur, W = _load(name)
    cfg = qo.QPConfig(N=int(d["N"]), NT=int(d["NT"]))
    assert np.array_equal(cfg.D, d["D"]) and cfg.rho == float(d["rho"])
    out = co.solve_batch(cfg, d["x0"], d["ub"], d["stuck"], d["xref"], uref=ur, warmU=W, nthreads=4)
    assert (out["status"] == 0).all()
    # BVLS (1e-15 tol) against the port: finished by the active-set polish (ftmpc_oracle.c:polish_box) it is the exact solution on the
    # verified set (5e-11 N measured on the largest shape) ...
    assert np.abs(out["U"] - d["U"]).max() < 1e-9
    assert np.abs(out["u0"] - d["u0"]).max() < 1e-9
    assert (out["U"][d["ub"][:, None, :].repeat(cfg.N, 1) == 0] == 0).all()
    # ... the interior-point iteration alone (mu 1e-13) is not, where late-horizon thrusters are weakly determined (n up to 560); it
    # takes one to two passes more
    ipm = co.solve_batch(cfg, d["x0"], d["ub"], d["stuck"], d["xref"], uref=ur, warmU=W, nthreads=4, polish=False)
    assert (ipm["status"] == 0).all()
    assert np.abs(ipm["U"] - d["U"]).max() < (5e-7 if cfg.NT == 8 else 3e-6) and np.abs(ipm["u0"] - d["u0"]).max() < 5e-7
    assert ipm["iters"].mean() > out["iters"].mean()


def test_box_polish_of_the_port_is_exact_on_degenerate_batches():
    """polish_box on batches with many weakly active bounds (16 thrusters, up to nine of them broken, warm start beyond the bounds): every
    converged instance within 1e-9 N of BVLS, where the iteration alone is up to ~1e-5 N away; an instance the polish does not verify
    falls back to the iteration (still converged)."""
    N, NT, B = 15, 16, 24
    rng = np.random.default_rng(7701)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 1, 7801)
    for b in range(B):
        k = int(rng.integers(0, 9))
        idx = rng.choice(np.flatnonzero(ub[b] > 0), k, replace=False)
        ub[b, idx] = 0.0
        stuck[b, idx] = rng.uniform(0, 1, k) * rm.F_MAX
    W = np.ascontiguousarray(rng.uniform(-0.2, 1.2 * rm.F_MAX, (B, N, NT)))
    pol = co.solve_batch(cfg, x0, ub, stuck, xref, warmU=W.copy(), nthreads=4, max_iters=60)
    ipm = co.solve_batch(cfg, x0, ub, stuck, xref, warmU=W.copy(), nthreads=4, max_iters=60, polish=False)
    assert (pol["status"] == 0).all() and (ipm["status"] == 0).all()
    worst_pol = worst_ipm = 0.0
    for b in range(B):
        _, U, _ = qo.solve_instance(cfg, x0[b], ub[b], stuck[b], xref, warmU=W[b], exact=True)
        worst_pol = max(worst_pol, np.abs(pol["U"][b] - U).max())
        worst_ipm = max(worst_ipm, np.abs(ipm["U"][b] - U).max())
    assert worst_pol < 1e-9 and worst_ipm < 2e-5, (worst_pol, worst_ipm)
    assert worst_ipm > 10 * worst_pol


def test_numpy_and_c_build_agree_and_are_spd():
    cfg = qo.QPConfig(N=20, NT=8)
    x0, ub, stuck, xref = qo.make_batch(4, 20, 8, 2, 77)
    for b in range(4):
        qp = qo.build_qp(cfg, x0[b], ub[b], stuck[b], xref)
        H, g, lo, hi = co.build_qp(cfg, x0[b], ub[b], stuck[b], xref)
        assert np.abs(H - qp["H"]).max() < 1e-10 * np.abs(H).max()
        assert np.abs(g - qp["g"]).max() < 1e-10 * max(1, np.abs(g).max())
        assert np.allclose(H, H.T) and np.linalg.eigvalsh(H).min() >= 2 * cfg.rho * (1 - 1e-9)
        assert np.array_equal(lo, -qp["Ubar"]) and np.array_equal(hi, qp["ub"] - qp["Ubar"])


def test_condensed_model_is_the_linearisation_of_the_rollout():
    """c_hat = c_bar + Bbar (U - Ubar) must match the nonlinear rollout to second order."""
    cfg = qo.QPConfig(N=8, NT=8)
    x0, ub, stuck, xref = qo.make_batch(1, 8, 8, 1, 5)
    rng = np.random.default_rng(0)
    W = rng.uniform(0.5, 2.5, (8, 8))
    cbar, A, Bg, Ubar = qo.linearize(cfg, x0[0], ub[0], stuck[0], W)
    dU = 1e-4 * rng.standard_normal((8, 8)) * (ub[0] > 0)
    c2, *_ = qo.linearize(cfg, x0[0], ub[0], stuck[0], Ubar + dU)
    d = np.zeros(13)
    for k in range(8):
        d = A[k] @ d + Bg[k] @ (cfg.D @ dU[k])
        assert np.abs((c2[k + 1] - cbar[k + 1]) - d).max() < 5e-7


def test_ipm_mirror_fp64_equals_bvls_and_fp32_is_within_tolerance():
    cfg = qo.QPConfig(N=20, NT=8)
    x0, ub, stuck, xref = qo.make_batch(6, 20, 8, 2, 1003)
    for b in range(6):
        qp = qo.build_qp(cfg, x0[b], ub[b], stuck[b], xref)
        lo, hi = -qp["Ubar"], qp["ub"] - qp["Ubar"]
        dx = qo.solve_exact(qp["H"], qp["g"], lo, hi)
        assert qo.kkt_residual(qp["H"], qp["g"], lo, hi, dx) < 1e-9
        d64, _, _, n64 = qo.ipm_box(qp["H"], qp["g"], lo, hi, iters=40)
        assert np.abs(d64 - dx).max() < 1e-6
        d32, _, _, n32 = qo.ipm_box(qp["H"], qp["g"], lo, hi, iters=16, dtype=np.float32)
        na = qp["na"]
        assert np.abs(d32 - dx)[:na].max() / rm.F_MAX < 1e-4   # the kernel's arithmetic, stage-0 command
        assert n32 <= 16


def test_no_active_thruster_and_all_faulted_edge_cases():
    cfg = qo.QPConfig(N=5, NT=8)
    x0, ub, stuck, xref = qo.make_batch(2, 5, 8, 0, 3)
    ub[0, :] = 0.0                      # every thruster broken: nothing to optimise
    stuck[0, :] = 1.0
    out = co.solve_batch(cfg, x0, ub, stuck, xref)
    assert (out["u0"][0] == 0).all() and out["iters"][0] == 0 and out["status"][0] == 0
    assert out["status"][1] == 0 and out["u0"][1].max() <= 3.4 + 1e-12 and out["u0"][1].min() >= 0


def test_plant_step_matches_numpy_restatement():
    cfg = qo.QPConfig(N=5, NT=16)
    rng = np.random.default_rng(9)
    fs = rm.FaultState(16).set_fault(10, 1.0).set_fault(11, 0.3)
    x = rng.standard_normal(13)
    u = rng.uniform(0, 3.4, 16)
    ref = rm.rk4(lambda s: rm.plant_dx_dt(s, u, cfg.D, fs.stuck, fs.ub), x)
    assert np.allclose(co.plant_step(cfg, x, u, fs.ub, fs.stuck), ref, atol=1e-13)


def test_general_constraint_fixtures_are_reproduced_and_certified():
    """tests/golden/qp_wrench_hull_n15.npz and qp_terminal_set_n20.npz (oracle/gen_golden.py:general_constraint_fixtures): the
    NumPy general-constraint IPM reproduces the stored solutions, and each stored solution passes the solver-independent KKT
    certificate on the QP rebuilt from the stored inputs (a subset here: the whole set runs on the GPU side)."""
    d = np.load(GOLD / "qp_wrench_hull_n15.npz")
    cfg = qo.QPConfig(N=int(d["N"]), NT=int(d["NT"]))
    assert np.array_equal(cfg.D, d["D"]) and (d["status"] == 0).all() and d["x0"].shape[0] >= 16
    for b in (0, 7, 15):
        tau0, T, st, _, qp = qo.solve_wrench_instance(cfg, d["x0"][b], d["ub"][b], d["stuck"][b], d["xref"])
        assert st == 0 and np.abs(T - d["G"][b]).max() < 1e-9 and np.abs(tau0 - d["tau0"][b]).max() < 1e-9
        dd = (d["G"][b] - qp["Tbar"]).reshape(-1)
        assert (qp["C"] @ dd <= qp["h"] + 1e-8).all()                          # the stored wrenches respect every hull row
        assert max(qo.kkt_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d"], qp["z"])) < 1e-4
    t = np.load(GOLD / "qp_terminal_set_n20.npz")
    cfg = qo.QPConfig(N=int(t["N"]), NT=int(t["NT"]))
    ok = np.flatnonzero(t["status"] == 0)
    assert ok.size >= 6 and ok.size < t["x0"].shape[0] and t["x0"].shape[0] >= 16
    with np.errstate(all="ignore"):
        for b in list(ok[:2]) + [int(np.flatnonzero(t["status"] != 0)[0])]:
            _, U, st, _, qp = qo.solve_box_terminal_instance(cfg, t["x0"][b], t["ub"][b], t["stuck"][b], t["xref"], (t["term_A"], t["term_b"]), iters=60)
            assert (st == 0) == (t["status"][b] == 0)
            if st == 0:
                assert np.abs(U - t["U"][b]).max() < 1e-9
                assert (t["term_A"] @ (qp["eN"] + qp["GN"] @ qp["d"]) <= t["term_b"] + 1e-7).all()


def _boundary_states(x0, At, bt, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    r = rm.spiral_r()
    for b in range(x0.shape[0]):
        e = rng.standard_normal(9)
        e *= scale / max((At @ e / bt).max(), 1e-9)
        R = rm.rot(x0[b, 6:10])
        w = rm.OMEGA_DES + e[6:9]
        x0[b, 0:3] = e[0:3] - R.T @ r
        x0[b, 3:6] = e[3:6] - R.T @ np.cross(w, r)
        x0[b, 10:13] = w


@pytest.mark.parametrize("with_set", [False, True])
def test_active_set_polish_reaches_the_exact_solution_of_the_general_forms(with_set):
    """The interior-point iterate at mu 1e-10 is up to 7e-5 f_max from the exact solution where rows are weakly active; the
    polish (polish_general: what the kernels run since round 4) closes that: KKT residuals ~1e-12 and agreement with the
    primal-dual active-set certificate (solve_general_exact, an independent iteration) to 1e-9 f_max."""
    N, NT, B = 15, 16, 24
    cfg = qo.QPConfig(N=N, NT=NT)
    d = np.load(GOLD / "qp_wrench_hull_terminal_n15.npz")
    At, bt = d["term_A"], d["term_b"]
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 9300)
    if with_set:
        _boundary_states(x0, At, bt, 9301)
    worst_ipm = worst_pol = 0.0
    solved = 0
    with np.errstate(all="ignore"):
        for b in range(B):
            try:
                qp = qo.build_qp_wrench(cfg, x0[b], ub[b], stuck[b], xref, term_set=(At, bt) if with_set else None)
            except ValueError:
                continue
            d0, s0, z0, _, st0 = qo.ipm_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d0"], qp["nhull"], iters=60, polish=False)
            d1, s1, z1, _, st1 = qo.ipm_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d0"], qp["nhull"], iters=60, polish=True)
            assert st0 == st1
            if st0 != 0:
                continue
            dx, zx = qo.solve_general_exact(qp["H"], qp["g"], qp["C"], qp["h"], d0, z0, s0)
            assert dx is not None and max(qo.kkt_general(qp["H"], qp["g"], qp["C"], qp["h"], dx, zx)) < 1e-7      # (complementarity: multipliers ~1e3 times slacks ~1e-12)
            assert max(qo.kkt_general(qp["H"], qp["g"], qp["C"], qp["h"], d1, z1)) < 1e-7
            worst_ipm = max(worst_ipm, np.abs(d0 - dx).max() / rm.F_MAX)
            worst_pol = max(worst_pol, np.abs(d1 - dx).max() / rm.F_MAX)
            solved += 1
    assert solved >= B // 2
    assert worst_pol <= 1e-9, worst_pol
    assert worst_ipm <= 2e-4      # (what the unpolished iterate is worth: the reason the polish exists)


def test_general_constraint_golden_fixtures_are_reproduced_by_the_oracle():
    """tests/golden/qp_wrench_hull_terminal_n15.npz (hull rows + terminal set) re-solved: same verdicts, same wrenches."""
    d = np.load(GOLD / "qp_wrench_hull_terminal_n15.npz")
    cfg = qo.QPConfig(N=int(d["N"]), NT=int(d["NT"]))
    for b in range(0, d["x0"].shape[0], 4):
        if d["status"][b] == 3:
            continue
        with np.errstate(all="ignore"):
            _, T, st, _, _ = qo.solve_wrench_instance(cfg, d["x0"][b], d["ub"][b], d["stuck"][b], d["xref"], term_set=(d["term_A"], d["term_b"]), iters=60)
        assert st == d["status"][b]
        if st == 0:
            assert np.abs(T - d["G"][b]).max() <= 1e-9


def test_state_bound_rows_of_the_oracle_and_their_golden_fixture():
    """The reference's optional state bounds (spiraling_mpc.py:129-130,179-185) as rows of the oracle's general form: the fixture
    tests/golden/qp_state_bounds_n20.npz re-solved (same verdicts, same sequences), the bounds hold along the linearised
    prediction, and the solution passes the KKT certificate."""
    d = np.load(GOLD / "qp_state_bounds_n20.npz")
    cfg = qo.QPConfig(N=int(d["N"]), NT=int(d["NT"]))
    seen = 0
    with np.errstate(all="ignore"):
        for b in range(0, d["x0"].shape[0], 3):
            _, U, st, _, qp = qo.solve_box_state_instance(cfg, d["x0"][b], d["ub"][b], d["stuck"][b], d["xref"], d["xlb"], d["xub"], iters=60)
            assert st == d["status"][b]
            if st != 0:
                continue
            seen += 1
            assert np.abs(U - d["U"][b]).max() <= 1e-9
            assert max(qo.kkt_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d"], qp["z"])) < 1e-7
            nh = qp["nhull"]
            assert (qp["C"][nh:] @ qp["d"] <= qp["h"][nh:] + 1e-9).all() and len(qp["srow"]) == qp["h"].size - nh
    assert seen >= 3
