"""GPU parity of the generalized-force formulation (6-D QP with the input hull, then allocation) and of the terminal
set, SURVEY.md section 8(f) ranks 2/3 -- the reference's own two-stage structure (spiraling_mpc.py:133-137,175-177,
199-202; input_bounds.py:43-76; control_allocator.py:65-94).
Checkers: oracle/qp_oracle.py:ipm_general (the same algorithm in NumPy, finished by the same active-set polish: its
solutions satisfy the KKT conditions to 1e-12 and agree with the primal-dual active-set certificate solve_general_exact to
1e-9 f_max) and, for the hull form, the THRUSTER-SPACE exact solution (BVLS) with the allocation weight rho -> 0, whose
optimal total wrench must coincide (the hull is the image of the thruster box).  Tolerance: 1e-6 f_max on wrenches /
thruster forces for the float64 kernel, 1e-4 f_max (north_star) for fp32 handles (kernel 11 + hand-over)."""
import numpy as np
import pytest

import ft_mpc_amd
from ft_mpc_amd.controllers.tools.input_bounds import hull_tables, zonotope_hrep
from ft_mpc_amd.controllers.tools.terminal_ingredients import load_terminal
from oracle import alloc_oracle as ao
from oracle import batch as ob
from oracle import qp_oracle as qo
from oracle import refmath as rm

pytestmark = pytest.mark.gpu
F_MAX = rm.F_MAX
TOL = 1e-6


def _near_terminal_set(B, N, NT, nf, seed, At, bt, scale=2.0):
    """Random poses / faults whose orbit-centre tracking error starts on `scale` times the boundary of the terminal set:
    for some of them the set is reachable within the horizon (often with active rows), for others it is not."""
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nf, seed)
    rng = np.random.default_rng(seed + 1)
    r = rm.spiral_r()
    for b in range(B):
        e = rng.standard_normal(9)
        e *= scale / max((At @ e / bt).max(), 1e-9)
        R = rm.rot(x0[b, 6:10])
        w = rm.OMEGA_DES + e[6:9]
        x0[b, 0:3] = e[0:3] - R.T @ r                    # robot_to_center (spiral_model.py:103-109) inverted
        x0[b, 3:6] = e[3:6] - R.T @ np.cross(w, r)
        x0[b, 10:13] = w
    return x0, ub, stuck, xref


def _reachable(qp):
    """Independent certificate (LP phase 1, HiGHS): is there any d with C d <= h at all?"""
    from scipy.optimize import linprog
    n, m, nh = qp["n"], len(qp["h"]), qp["nhull"]
    Aub = np.hstack([qp["C"], np.r_[np.zeros(nh), -np.ones(m - nh)][:, None]])
    res = linprog(np.r_[np.zeros(n), 1.0], A_ub=Aub, b_ub=qp["h"], bounds=[(None, None)] * n + [(0, None)], method="highs")
    return res.status == 0 and res.fun <= 1e-9


@pytest.mark.parametrize("N,NT,nf,B", [(20, 16, 2, 40), (15, 16, 1, 24), (20, 8, 2, 24), (15, 16, 0, 8)])
def test_wrench_step_against_the_oracle(gpu_mpc_factory, N, NT, nf, B):
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nf, 7000 + N + NT)
    out = mpc.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), return_G=True)
    deg = out["status"] == 3
    assert deg.sum() <= B // 4
    for b in range(B):
        if deg[b]:
            with pytest.raises(ValueError):
                zonotope_hrep(cfg.D, ub[b], stuck[b])
            continue
        assert out["status"][b] == 0
        tau0, T, st, nit, qp = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref, mu_polish=1e-7)      # (kernel 13 leaves the iteration there)
        assert st == 0 and max(qo.kkt_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d"], qp["z"])) < 1e-8
        assert np.abs(out["G"][b] - T).max() / F_MAX <= TOL, (b, np.abs(out["G"][b] - T).max())
        assert abs(int(out["iters"][b]) - nit) <= 1
        # second stage: the allocated thruster forces realise tau0 - D stuck with minimum norm
        want = out["tau0"][b] - cfg.D @ stuck[b]
        assert out["alloc_status"][b] == 0
        assert np.abs(cfg.D @ out["u0"][b] - want).max() <= 1e-7 * (1 + np.abs(want).max())
        ref_u = ao.allocate(cfg.D, want, ub[b])[0]
        assert np.abs(out["u0"][b] - ref_u).max() / F_MAX <= 1e-6
        assert (out["u0"][b][ub[b] == 0] == 0).all() and (out["u0"][b] >= -1e-12).all() and (out["u0"][b] <= ub[b] + 1e-9).all()


def test_wrench_solution_is_the_thruster_space_solution_without_allocation_weight(gpu_mpc_factory):
    """Independent pin: BVLS on the thruster-space QP with rho = 1e-9 (no wrench hull anywhere in it) gives the same total wrench."""
    N, NT, B = 20, 16, 12
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 7101)
    out = mpc.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), return_G=True)
    cfg0 = qo.QPConfig(N=N, NT=NT, rho=1e-9)
    for b in np.flatnonzero(out["status"] == 0):
        _, U, _ = qo.solve_instance(cfg0, x0[b], ub[b], stuck[b], xref, exact=True)
        diff = np.abs(out["G"][b] - (U + stuck[b]) @ cfg0.D.T).max()
        assert diff / F_MAX <= 2e-6, (b, diff)      # (rho = 1e-9 there, none here; both exact on their active sets)
    assert (out["status"] == 0).sum() >= B - 2


def test_wrench_with_qhull_rows_warm_start_and_reference_window(gpu_mpc_factory):
    """Hull rows from the reference's own recipe (corner enumeration + Qhull, restated in oracle/refmath.py:input_hull)
    instead of the zonotope shortcut; warm start given as wrenches; circle reference (uref != 0)."""
    N, NT, B = 15, 16, 6
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, _, _, _ = qo.make_batch(B, N, NT, 0, 7201)
    ub = np.full((B, NT), F_MAX); stuck = np.zeros((B, NT))
    ub[:, [10, 11]] = 0.0; stuck[:, 10] = F_MAX; stuck[:, 11] = 0.4 * F_MAX       # reactive.yaml-like: 10 and 11 stuck
    Aq, bq, _ = rm.input_hull(cfg.D, stuck[0], ub[0])
    assert Aq.shape == (26, 6)
    hull = dict(A=Aq[None], set=np.zeros(B, np.int32), b=np.tile(bq, (B, 1)), rows=26, degenerate=np.zeros(B, bool))
    traj = rm.circle_trajectory(0.1, 10, radius=0.65, s_per_circle=40.0)
    xr_all, ur_all = rm.assign_trajectory(traj, N)
    xw, uw = rm.trajectory_window(xr_all, ur_all, 1.0, N)
    rng = np.random.default_rng(5)
    W = np.ascontiguousarray(np.tile(cfg.D @ (ub[0] / 2 + stuck[0]), (B, N, 1)) + rng.uniform(-0.2, 0.2, (B, N, 6)))
    W0 = W.copy()
    out = mpc.solve_wrench(x0, ub, stuck, xw.reshape(-1, order="F"), uref=uw.reshape(-1, order="F"), warmG=W, hull=hull, return_G=True)
    assert (out["status"] == 0).all()
    for b in range(B):
        _, T, st, _, _ = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xw, uref=uw, warmG=W0[b], hull=(Aq, bq))
        assert st == 0 and np.abs(out["G"][b] - T).max() / F_MAX <= TOL
    assert np.array_equal(W, out["G"])       # warm buffer updated in place
    mine = hull_tables(cfg.D, ub, stuck)
    out2 = mpc.solve_wrench(x0, ub, stuck, xw.reshape(-1, order="F"), uref=uw.reshape(-1, order="F"), warmG=W0.copy(), hull=mine)
    assert np.abs(out2["tau0"] - out["tau0"]).max() / F_MAX <= TOL


def test_wrench_persistent_loop(gpu_mpc_factory):
    """B > 2 x resident workgroups (one per CU): every workgroup re-uses its tile / panel slots and LDS tables."""
    N, NT, B = 12, 16, 700
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 7301)
    out = mpc.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"))
    ok = np.flatnonzero(out["status"] != 3)
    assert (out["status"][ok] == 0).all()
    err = 0.0
    for b in ok:
        tau0, _, st, _, _ = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref)
        assert st == 0
        err = max(err, np.abs(out["tau0"][b] - tau0).max())
    assert err / F_MAX <= TOL, err


@pytest.mark.parametrize("form", ["box", "wrench"])
def test_terminal_set_rows(gpu_mpc_factory, form):
    """72-row terminal set of config/terminal.yaml in the IPM (flagged): thruster-space (box + terminal rows) and
    generalized-force (hull + terminal rows) against the oracle.  Where the set is reachable (LP certificate) the
    solutions agree and some have active rows; where it is not, both report it (non-zero status, finite output)."""
    N, NT, B = 20, 8, 24
    term = load_terminal().term_set
    At, bt = term.A, term.b.reshape(-1)
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=60, terminal_set=term)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = _near_terminal_set(B, N, NT, 2, 12, At, bt)
    xr = xref.reshape(-1, order="F")
    out = mpc.solve(x0, ub, stuck, xr, return_U=True) if form == "box" else mpc.solve_wrench(x0, ub, stuck, xr, return_G=True)
    key = "U" if form == "box" else "G"
    solved = active = 0
    with np.errstate(all="ignore"):
        for b in range(B):
            if form == "box":
                _, X, st, nit, qp = qo.solve_box_terminal_instance(cfg, x0[b], ub[b], stuck[b], xref, (At, bt), iters=60)
            else:
                _, X, st, nit, qp = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref, term_set=(At, bt), iters=60)
            assert (st == 0) == _reachable(qp), b
            assert (out["status"][b] == 0) == (st == 0), (b, out["status"][b], st)
            assert np.isfinite(out[key][b]).all()
            if st != 0:
                continue
            solved += 1
            assert max(qo.kkt_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d"], qp["z"])) < 1e-7
            assert np.abs(out[key][b] - X).max() / F_MAX <= TOL, (b, np.abs(out[key][b] - X).max())
            assert (At @ (qp["eN"] + qp["GN"] @ qp["d"]) <= bt + 1e-7).all()
            active += int((qp["z"][qp["nhull"]:] > 1e-6).any())
    assert solved >= 6 and active >= 2 and solved < B      # reachable with active rows, and unreachable, both occur
    free = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=60)
    ref = free.solve(x0, ub, stuck, xr, return_U=True) if form == "box" else free.solve_wrench(x0, ub, stuck, xr, return_G=True)
    ok = out["status"] == 0
    assert np.abs(ref[key][ok] - out[key][ok]).max() / F_MAX > 1e-4     # the rows matter


def test_unreachable_terminal_set_is_reported_not_raised(gpu_mpc_factory):
    """Random far-away states: the terminal set cannot be reached within the horizon.  The reference logs IPOPT's
    failure and carries on (spiraling_mpc.py:347-352); here the instance gets a non-zero status and finite outputs."""
    N, NT, B = 20, 8, 16
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40, terminal_set=True)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 13)
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"))
    assert (out["status"] != 0).sum() >= 3
    assert np.isfinite(out["u0"]).all() and (out["u0"] >= 0).all() and (out["u0"] <= ub + 1e-9).all()


@pytest.mark.parametrize("dtype,N,nf", [("f32", 20, 0), ("f64", 12, 1), ("f32", 33, 2)])
def test_generic_vehicle_with_more_than_32_facets(gpu_mpc_factory, dtype, N, nf):
    """The synthetic 8-thruster benchmark matrix is generic: 2 C(8,5) = 112 facets without a fault (the reference vehicle's
    symmetric layout has 26 for every fault set; input_bounds.py:43-76 has no limit).  Rounds 2-3 refused such hulls; kernel 13
    (float64, Riccati recursion: a stage's rows enter as one 6 x 6 block however many they are) takes up to 128, at BASELINE's
    horizon and beyond, whatever the handle's dtype -- against oracle/qp_oracle.py instance by instance."""
    NT, B = 8, 12
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype=dtype, max_iters=40)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nf, 7900 + N)
    hull = hull_tables(cfg.D, ub, stuck)
    assert hull["rows"] > 32 or nf == 2      # (two of eight broken: six generators, 12 facets -- the same kernel)
    out = mpc.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), hull=hull, return_G=True)
    assert mpc.last_handed_over() == 0
    for b in range(B):
        assert out["status"][b] == 0 and out["alloc_status"][b] == 0, (b, out["status"][b], out["alloc_status"][b])
        tau0, T, st, nit, qp = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref, mu_polish=1e-7)
        assert st == 0 and qp["mh"] == hull["rows"] or qp["mh"] <= hull["rows"]
        assert np.abs(out["G"][b] - T).max() / F_MAX <= TOL, (b, np.abs(out["G"][b] - T).max() / F_MAX)
        assert abs(int(out["iters"][b]) - nit) <= 1
        want = out["tau0"][b] - cfg.D @ stuck[b]
        assert np.abs(cfg.D @ out["u0"][b] - want).max() <= 1e-7 * (1 + np.abs(want).max())


def test_dense_kernel_with_more_than_32_facets_is_refused(gpu_mpc_factory):
    """kernel_select = "dense" keeps the dense float64 kernel, which holds 32 rows per stage: refused, not truncated."""
    from ft_mpc_amd._lib import FtmpcError
    N, NT = 12, 8
    x0, ub, stuck, xref = qo.make_batch(4, N, NT, 0, 1)
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40, kernel_select="dense")
    with pytest.raises(FtmpcError):
        mpc.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), hull=hull_tables(qo.QPConfig(N=N, NT=NT).D, ub, stuck))


def test_terminal_set_on_an_fp32_handle(gpu_mpc_factory):
    """An fp32 handle with the terminal set: the thruster form (box + terminal rows exist in float64 only) runs on the float64
    kernel -- the bits of a float64 handle; the reference's two-stage form (hull rows + terminal set, spiraling_mpc.py:175-202)
    runs on kernel 11 with the rank-9 terminal term: same reachable / unreachable verdicts as the float64 kernel, wrenches
    within 1e-4 f_max where the set is reachable, some of them with active terminal rows."""
    N, NT, B = 15, 16, 96
    term = load_terminal().term_set
    At, bt = term.A, term.b.reshape(-1)
    x0, ub, stuck, xref = _near_terminal_set(B, N, NT, 2, 31, At, bt, scale=1.5)
    xr = xref.reshape(-1, order="F")
    m32 = gpu_mpc_factory(N=N, NT=NT, dtype="f32", max_iters=60, terminal_set=term)
    m64 = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=60, terminal_set=term)
    a, b = m32.solve(x0, ub, stuck, xr, return_U=True), m64.solve(x0, ub, stuck, xr, return_U=True)
    assert np.array_equal(a["U"], b["U"], equal_nan=True) and np.array_equal(a["status"], b["status"])
    w32, w64 = m32.solve_wrench(x0, ub, stuck, xr, return_G=True), m64.solve_wrench(x0, ub, stuck, xr, return_G=True)
    has = w64["status"] != 3
    ok64, ok32 = (w64["status"] == 0) & has, (w32["status"] == 0) & has
    assert ok64.sum() >= 12 and (has & ~ok64).sum() >= 4                     # reachable and unreachable both occur
    assert (ok64 != ok32).sum() <= 1, (ok64.sum(), ok32.sum())               # (a borderline instance may fall either way)
    both = ok64 & ok32
    err = np.abs(w32["G"][both] - w64["G"][both]).max(axis=(1, 2)) / F_MAX
    assert err.max() <= 1e-4, (err.max(), int(err.argmax()))
    assert np.isfinite(w32["G"][has]).all()
    free = gpu_mpc_factory(N=N, NT=NT, dtype="f32", max_iters=60).solve_wrench(x0, ub, stuck, xr, return_G=True)
    assert (np.abs(free["G"][both] - w32["G"][both]).max(axis=(1, 2)) / F_MAX > 1e-3).sum() >= 4      # the rows matter


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_reference_formulation_at_the_baseline_horizon(gpu_mpc_factory, dtype):
    """Hull rows AND the terminal set at N = 20 with 16 thrusters -- the reference's NLP structure (spiraling_mpc.py:175-177,198-202)
    at every BASELINE config's horizon.  Round 3: the dense float64 kernel only (n = 120 tiles in a global slot).  Now kernel 13
    (float64, Riccati recursion) whatever the handle's dtype; tracking error on the boundary of the set, against the oracle."""
    N, NT, B = 20, 16, 192
    term = load_terminal().term_set
    At, bt = term.A, term.b.reshape(-1)
    x0, ub, stuck, xref = _near_terminal_set(B, N, NT, 2, 9400, At, bt, scale=1.0)
    ref = ob.solve_wrench_batch(N, NT, x0, ub, stuck, xref, term_set=(At, bt), iters=60, mu_polish=1e-7)
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype=dtype, max_iters=60, terminal_set=term)
    out = mpc.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), return_G=True)
    assert mpc.last_handed_over() == 0
    assert np.array_equal(ref["status"] == 3, out["status"] == 3)
    ok_ref, ok = ref["status"] == 0, out["status"] == 0
    assert ok_ref.sum() >= B // 2 and (ref["active"][ok_ref] > 0).all(axis=1).sum() >= B // 10
    assert (ok_ref != ok).sum() <= 1, np.flatnonzero(ok_ref != ok)
    both = ok_ref & ok
    err = np.abs(out["G"][both] - ref["G"][both]).max(axis=(1, 2)) / F_MAX
    assert err.max() <= TOL, (err.max(), int(np.flatnonzero(both)[err.argmax()]))
    assert np.abs(out["iters"][both].astype(int) - ref["iters"][both]).max() <= 1
    assert out["alloc_status"][both].max() == 0


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-4), ("f64", 1e-6)])
def test_two_stage_form_with_the_terminal_set_on_the_boundary_batch_against_the_oracle(gpu_mpc_factory, dtype, tol):
    """The reference's NLP always carries hull rows AND the terminal set (spiraling_mpc.py:175-177,198-202).  2 048 vehicles with
    the tracking error ON the boundary of the set (hull and terminal rows active together in ~40 % of them, weakly active rows
    in many) against oracle/qp_oracle.py instance by instance: fp32 handle = ftmpc_solve_hull32_kernel<6, true> with its
    hand-over to the float64 kernel, float64 handle = the float64 kernel (MODE 3), both finished by the active-set polish.
    EVERY instance the oracle solves must be solved and lie within the tolerance (north_star: 1e-4 f_max for fp32), whole
    horizon; reachable / unreachable verdicts may differ on borderline instances only (<= 0.1 %)."""
    N, NT, B = 15, 16, 2048
    term = load_terminal().term_set
    At, bt = term.A, term.b.reshape(-1)
    x0, ub, stuck, xref = _near_terminal_set(B, N, NT, 2, 9100, At, bt, scale=1.0)
    ref = ob.solve_wrench_batch(N, NT, x0, ub, stuck, xref, term_set=(At, bt), iters=60)
    out = gpu_mpc_factory(N=N, NT=NT, dtype=dtype, max_iters=60, terminal_set=term).solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), return_G=True)
    assert np.array_equal(ref["status"] == 3, out["status"] == 3)
    ok_ref, ok = ref["status"] == 0, out["status"] == 0
    assert ok_ref.sum() >= B // 2 and (ref["active"][ok_ref] > 0).all(axis=1).sum() >= B // 8
    assert (ok_ref != ok).sum() <= B // 1000, ((ok_ref != ok).sum(), np.flatnonzero(ok_ref != ok)[:8])
    both = ok_ref & ok
    err = np.abs(out["G"][both] - ref["G"][both]).max(axis=(1, 2)) / F_MAX
    assert err.max() <= tol, (err.max(), int(np.flatnonzero(both)[err.argmax()]), int((err > tol).sum()))
    assert out["alloc_status"][both].max() == 0
    assert np.isfinite(out["G"][out["status"] != 3]).all()


def test_general_constraint_forms_against_golden(gpu_mpc_factory):
    """The committed fixtures of the reference's own formulation (tests/golden/qp_wrench_hull_n15.npz: hull rows per stage;
    qp_terminal_set_n20.npz: 72-row terminal set, reachable and unreachable instances; qp_wrench_hull_terminal_n15.npz: both
    together) against the HIP path."""
    from pathlib import Path
    g = Path(__file__).parent / "golden"
    d = np.load(g / "qp_wrench_hull_n15.npz")
    mpc = gpu_mpc_factory(N=int(d["N"]), NT=int(d["NT"]), dtype="f64", max_iters=40)
    out = mpc.solve_wrench(d["x0"], d["ub"], d["stuck"], d["xref"].reshape(-1, order="F"), return_G=True)
    assert (out["status"] == 0).all() and d["x0"].shape[0] >= 16
    assert np.abs(out["G"] - d["G"]).max() / F_MAX <= TOL and np.abs(out["tau0"] - d["tau0"]).max() / F_MAX <= TOL
    t = np.load(g / "qp_terminal_set_n20.npz")
    mt = gpu_mpc_factory(N=int(t["N"]), NT=int(t["NT"]), dtype="f64", max_iters=60, terminal_set=(t["term_A"], t["term_b"]))
    o2 = mt.solve(t["x0"], t["ub"], t["stuck"], t["xref"].reshape(-1, order="F"), return_U=True)
    ok = t["status"] == 0
    assert ((o2["status"] == 0) == ok).all() and ok.sum() >= 6
    assert np.abs(o2["U"][ok] - t["U"][ok]).max() / F_MAX <= TOL and np.isfinite(o2["U"]).all()
    w = np.load(g / "qp_wrench_hull_terminal_n15.npz")
    ok = w["status"] == 0
    assert ok.sum() >= 16 and ((w["active_hull"] > 0) & (w["active_term"] > 0)).sum() >= 6
    for dtype, tol in (("f64", TOL), ("f32", 1e-4)):
        mw = gpu_mpc_factory(N=int(w["N"]), NT=int(w["NT"]), dtype=dtype, max_iters=60, terminal_set=(w["term_A"], w["term_b"]))
        o3 = mw.solve_wrench(w["x0"], w["ub"], w["stuck"], w["xref"].reshape(-1, order="F"), return_G=True)
        assert np.array_equal(o3["status"] == 0, ok) and np.array_equal(o3["status"] == 3, w["status"] == 3), (dtype, o3["status"], w["status"])
        assert np.abs(o3["G"][ok] - w["G"][ok]).max() / F_MAX <= tol, (dtype, np.abs(o3["G"][ok] - w["G"][ok]).max() / F_MAX)


# ---- kernel 11: the same formulation in fp32 on one wave per instance (N <= 16, up to 32 hull rows, no terminal set) ----
TOL32 = 1e-4      # the fp32 kernels' specification: u0 / tau within 1e-4 f_max of the exact solution


@pytest.mark.parametrize("N,NT,nf,B", [(15, 16, 2, 48), (15, 16, 0, 8), (16, 16, 1, 16), (10, 16, 3, 16)])
def test_fp32_wrench_step_against_the_oracle(gpu_mpc_factory, N, NT, nf, B):
    """ftmpc_solve_hull32_kernel against oracle/qp_oracle.py:ipm_general, instance by instance: whole-horizon wrenches within
    1e-4 f_max, every instance converged, allocation exact for the wrench the kernel hands over."""
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f32", max_iters=40)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nf, 7600 + N + NT)
    out = mpc.solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), return_G=True)
    deg = out["status"] == 3
    assert deg.sum() <= B // 4
    worst = 0.0
    for b in np.flatnonzero(~deg):
        assert out["status"][b] == 0, (b, out["status"][b], out["iters"][b])
        tau0, T, st, nit, qp = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref)
        assert st == 0
        worst = max(worst, np.abs(out["G"][b] - T).max() / F_MAX)
        assert abs(int(out["iters"][b]) - nit) <= 5      # (kernel 11 leaves the interior-point iteration at mu 1e-7 for its polish, the oracle at 1e-10)
        want = out["tau0"][b] - cfg.D @ stuck[b]
        assert out["alloc_status"][b] == 0
        assert np.abs(cfg.D @ out["u0"][b] - want).max() <= 1e-6 * (1 + np.abs(want).max())
        assert (out["u0"][b][ub[b] == 0] == 0).all() and (out["u0"][b] >= -1e-12).all() and (out["u0"][b] <= ub[b] + 1e-9).all()
    assert worst <= TOL32, worst


def test_fp32_wrench_warm_start_reference_window_and_persistent_grid(gpu_mpc_factory):
    """Large batch (several instances per resident wave), wrench warm start, circle reference with uref != 0, mixed fault
    sets (several hull tables): kernel 11 against the float64 kernel on the same inputs."""
    N, NT, B = 15, 16, 6000
    x0, ub, stuck, _ = qo.make_batch(B, N, NT, 2, 7700)
    cfg = qo.QPConfig(N=N, NT=NT)
    traj = rm.circle_trajectory(0.1, 10, radius=0.65, s_per_circle=40.0)
    xr_all, ur_all = rm.assign_trajectory(traj, N)
    xw, uw = rm.trajectory_window(xr_all, ur_all, 1.0, N)
    hull = hull_tables(cfg.D, ub, stuck)
    rng = np.random.default_rng(6)
    ctr = (ub / 2 + stuck) @ cfg.D.T
    W = np.ascontiguousarray(ctr[:, None, :] + rng.uniform(-0.05, 0.05, (B, N, 6)))
    ref = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40).solve_wrench(
        x0, ub, stuck, xw.reshape(-1, order="F"), uref=uw.reshape(-1, order="F"), warmG=W.copy(), hull=hull, return_G=True)
    W32 = W.copy()
    out = gpu_mpc_factory(N=N, NT=NT, dtype="f32", max_iters=40).solve_wrench(
        x0, ub, stuck, xw.reshape(-1, order="F"), uref=uw.reshape(-1, order="F"), warmG=W32, hull=hull, return_G=True)
    ok = ref["status"] == 0
    assert ok.sum() >= B - B // 8 and (out["status"][ok] == 0).all()
    assert np.array_equal(ref["status"] == 3, out["status"] == 3)
    err = np.abs(out["G"][ok] - ref["G"][ok]).max(axis=(1, 2)) / F_MAX
    assert err.max() <= TOL32, (err.max(), int(err.argmax()))
    assert np.abs(out["u0"][ok] - ref["u0"][ok]).max() / F_MAX <= TOL32      # the thrust command AFTER allocation: north_star's 1e-4
    assert np.array_equal(W32[ok], out["G"][ok])       # warm buffer updated in place
    assert np.abs(out["iters"][ok].astype(int) - ref["iters"][ok]).max() <= 5      # (up to three polish rounds on either side)


def test_fp32_wrench_kernel_select_dense_keeps_the_float64_kernel(gpu_mpc_factory):
    """kernel_select = "dense" keeps the dense float64 kernel whatever the handle's dtype (same bits from an fp32 and a float64
    handle); the default float64 route is kernel 13 (Riccati recursion): the same polished solution to 1e-9 f_max."""
    N, NT, B = 15, 16, 16
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 1, 7800)
    a = gpu_mpc_factory(N=N, NT=NT, dtype="f32", max_iters=40, kernel_select="dense").solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), return_G=True)
    b = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40, kernel_select="dense").solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), return_G=True)
    assert np.array_equal(a["G"], b["G"], equal_nan=True) and np.array_equal(a["status"], b["status"])
    c = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40).solve_wrench(x0, ub, stuck, xref.reshape(-1, order="F"), return_G=True)
    assert np.array_equal(c["status"], b["status"]) and np.abs(c["G"] - b["G"]).max() / F_MAX <= 1e-9


def test_fp32_wrench_step_against_golden(gpu_mpc_factory):
    """Kernel 11 against the committed fixture of the reference's own formulation (tests/golden/qp_wrench_hull_n15.npz)."""
    from pathlib import Path
    d = np.load(Path(__file__).parent / "golden" / "qp_wrench_hull_n15.npz")
    mpc = gpu_mpc_factory(N=int(d["N"]), NT=int(d["NT"]), dtype="f32", max_iters=40)
    out = mpc.solve_wrench(d["x0"], d["ub"], d["stuck"], d["xref"].reshape(-1, order="F"), return_G=True)
    assert (out["status"] == 0).all() and (out["alloc_status"] == 0).all()
    assert np.abs(out["G"] - d["G"]).max() / F_MAX <= TOL32 and np.abs(out["tau0"] - d["tau0"]).max() / F_MAX <= TOL32
