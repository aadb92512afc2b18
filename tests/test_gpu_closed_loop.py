"""GPU: BASELINE config 1 plumbing -- the examples/sim.py closed loop (16 thrusters, N=15, hover,
IC of the reference's examples/sim.py:49-54) with the MPC step on the MI355X, against the same loop
driven by the float64 C oracle.  float64 kernel: states agree to 1e-6 over 40 steps."""
import numpy as np
import pytest

from ft_mpc_amd.controllers.spiraling_mpc import SpiralingController
from ft_mpc_amd.models.spiral_model import SpiralModel
from ft_mpc_amd.models.sys_model import SystemModel
from ft_mpc_amd.simulation.sim_env import SimulationEnvironment
from ft_mpc_amd.util.broken_thruster import BrokenThruster
from ft_mpc_amd.util.controller_debug import ControllerDebug
from oracle import c_oracle as co
from oracle import qp_oracle as qo

pytestmark = pytest.mark.gpu
PARAMS = {"horizon": 15, "param_set": "P1", "P1": {"Q": [1, 1, 1, 1, 1, 1, 2, 2, 2], "R": [0.1, 0.1, 0.1, 0.01, 0.01, 0.01]},
          "max_iters": 40}
IC = dict(position=[1, 0, 1], velocity=[1, 0.5, 0],
          orientation=[0.03266701292872763, 0.26925564114813405, 0.3862204035220014, 0.8816280768439285],
          angular_velocity=[0.3, 0.8, -0.1])


class OracleController:
    """Same seam, same warm-start policy, solve by oracle/ftmpc_oracle.c."""

    def __init__(self, model, N):
        self.model, self.N, self.prev = model, N, None
        self.cfg = qo.QPConfig(N=N, NT=16)
        self.xref = np.zeros((9, N + 1))
        self.xref[8] = 0.6

    def get_control(self, x, t):
        warm = None if self.prev is None else np.vstack([self.prev[1:], np.zeros((1, 16))])[None]
        out = co.solve_batch(self.cfg, x[None], self.model.u_ub_physical[None], self.model.faulty_force.reshape(1, -1),
                             self.xref, uref=np.zeros((6, self.N + 1)), warmU=warm, max_iters=60)
        self.prev = out["U"][0]
        return out["u0"][0]


@pytest.mark.parametrize("faults", [[], [(10, 1.0), (11, 1.0)]])
def test_closed_loop_matches_oracle_loop(faults):
    def build():
        m = SystemModel(0.1)
        for i, a in faults:
            m.set_fault(BrokenThruster(i, a))
        return m
    m1, m2 = build(), build()
    hist = ControllerDebug()
    ctrl = SpiralingController(SpiralModel.from_system_model(m1), PARAMS, hist, quiet=True)
    ctrl.load_trajectory("hover", 30)
    env1 = SimulationEnvironment(m1, ctrl, seed=7)
    env2 = SimulationEnvironment(m2, OracleController(m2, 15), seed=7)
    for e in (env1, env2):
        e.set_initial_state(**IC)
    c_start = SpiralModel.from_system_model(m1).robot_to_center(env1.state)
    for _ in range(40):
        env1.step()
        env2.step()
        assert np.abs(env1.state - env2.state).max() < 1e-6
    assert len(hist.history) == 40 and hist.inputs().shape == (40, 16)
    assert (hist.inputs() >= -1e-9).all() and (hist.inputs() <= 3.4 + 1e-9).all()
    assert (hist.inputs()[:, [i for i, _ in faults]] == 0).all()
    c_end = SpiralModel.from_system_model(m1).robot_to_center(env1.state)
    assert np.linalg.norm(c_end[3:6]) < np.linalg.norm(c_start[3:6])     # the orbit centre is being braked


def test_solve_mpc_signature():
    m = SystemModel(0.1)
    sm = SpiralModel.from_system_model(m)
    ctrl = SpiralingController(sm, PARAMS, None, quiet=True)
    ctrl.load_trajectory("hover_0_0_0", 5)
    x = np.array([1, 0, 1, 1, .5, 0, *IC["orientation"], .3, .8, -.1], float)
    xs, us, dt, cost, status = ctrl.solve_mpc(sm.robot_to_center(x))
    assert len(xs) == 16 and len(us) == 15 and us[0].shape == (16,) and status == "Solve_Succeeded" and cost > 0 and dt > 0


def _hover_traj(N, T):
    xr = np.zeros((9, T + N))
    xr[8] = 0.6
    return xr


def test_on_device_closed_loop_matches_oracle_loop_f64(gpu_mpc_factory):
    """ftmpc_simulate_batch (plant step, noise, renormalisation and warm-start shift on the device) against
    oracle/closed_loop.py: 16-thruster vehicle, shipped double fault, float64 kernel."""
    from oracle import closed_loop as cl
    N, NT, B, T = 15, 16, 5, 12
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40)
    x0, _, _, _ = qo.make_batch(B, N, NT, 0, 99)
    ub = np.full((B, NT), 3.4); stuck = np.zeros((B, NT))
    ub[:, [10, 11]] = 0.0; stuck[:, [10, 11]] = 3.4
    xr = _hover_traj(N, T)
    out = mpc.simulate(x0, ub, stuck, xr, T, seed=5, return_inputs=True)
    xo, uo = cl.simulate(qo.QPConfig(N=N, NT=NT), x0, ub, stuck, xr, T, seed=5)
    assert out["not_converged"].sum() == 0
    assert np.abs(out["u"] - uo).max() < 1e-6
    assert np.abs(out["x"] - xo).max() < 1e-7
    assert np.allclose(np.linalg.norm(out["x"][:, 6:10], axis=1), 1.0, atol=1e-12)


def test_on_device_closed_loop_f32_campaign(gpu_mpc_factory):
    """fp32 kernels in the loop (8 thrusters, random double faults, warm-started): the closed loop stays
    within the accumulated per-step tolerance of the float64 oracle loop."""
    from oracle import closed_loop as cl
    N, NT, B, T = 20, 8, 24, 8
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, _ = qo.make_batch(B, N, NT, 2, 1234)
    xr = _hover_traj(N, T)
    out = mpc.simulate(x0, ub, stuck, xr, T, seed=11, return_inputs=True)
    xo, uo = cl.simulate(qo.QPConfig(N=N, NT=NT), x0, ub, stuck, xr, T, seed=11)
    assert out["not_converged"].sum() == 0
    assert np.abs(out["u"] - uo).max() / 3.4 < 5e-4      # 1e-4 per step, perturbation growth over 8 steps
    assert np.abs(out["x"] - xo).max() < 1e-4


def test_on_device_closed_loop_f32_campaign_reference_vehicle(gpu_mpc_factory):
    """The reference's vehicle (16 thrusters, N = 15) on kernel 8 in the on-device loop: warm starts, shifted references,
    faults -- against the float64 oracle loop."""
    from oracle import closed_loop as cl
    N, NT, B, T = 15, 16, 16, 6
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, _ = qo.make_batch(B, N, NT, 2, 1235)
    xr = _hover_traj(N, T)
    out = mpc.simulate(x0, ub, stuck, xr, T, seed=12, return_inputs=True)
    xo, uo = cl.simulate(qo.QPConfig(N=N, NT=NT), x0, ub, stuck, xr, T, seed=12)
    assert out["not_converged"].sum() == 0
    assert np.abs(out["u"] - uo).max() / 3.4 < 5e-4
    assert np.abs(out["x"] - xo).max() < 1e-4


def test_relinearisation_converges_and_matches_oracle(gpu_mpc_factory):
    """Sequential QP (re-linearise about the previous solution): the GPU loop equals the oracle loop and
    the iterates contract (first step: nonlinear dynamics, quadratic terminal cost)."""
    N, NT, B = 15, 16, 4
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40)
    x0, _, _, xref = qo.make_batch(B, N, NT, 0, 321)
    ub = np.full((B, NT), 3.4); stuck = np.zeros((B, NT))
    ub[:, 3] = 0.0; stuck[:, 3] = 1.0
    xr = xref.reshape(-1, order="F")
    cfg = qo.QPConfig(N=N, NT=NT)
    outs = [mpc.solve(x0, ub, stuck, xr, return_U=True, relinearize=k)["U"] for k in (0, 1, 2, 3)]
    W = None
    for k in range(4):
        ref = co.solve_batch(cfg, x0, ub, stuck, xref, warmU=W, max_iters=60)
        assert np.abs(outs[k] - ref["U"]).max() < 1e-6
        W = ref["U"]
    # full-step sequential QP (no line search): the iterates contract linearly, instance by instance
    d = np.array([np.abs(outs[k + 1] - outs[k]).reshape(B, -1).max(axis=1) for k in range(3)])
    assert (d[2] < d[0]).all()


def test_debug_export_has_reference_csv_format(tmp_path):
    m = SystemModel(0.1)
    m.set_fault(BrokenThruster(10, 1.0))
    hist = ControllerDebug()
    ctrl = SpiralingController(SpiralModel.from_system_model(m), PARAMS, hist, quiet=True)
    ctrl.load_trajectory("hover", 5)
    env = SimulationEnvironment(m, ctrl, seed=1)
    env.set_initial_state(**IC)
    env.run_simulation(0.5)
    path = hist.export(str(tmp_path / "debug_data"))
    lines = open(path).read().splitlines()
    head = lines[0].lstrip("# ").split(";")
    assert len(head) == 67 and head[0] == "time" and head[14] == "input_0" and head[-1] == "circle_angular_velocity_error_z"
    data = np.loadtxt(path, delimiter=";")
    assert data.shape == (5, 67) and np.allclose(data[:, 0], [0, .1, .2, .3, .4])
    assert np.allclose(data[:, 14 + 10], 0.0)                      # broken thruster is never commanded
    assert np.allclose(data[:, 30:33], (m.D @ data[:, 14:30].T).T[:, 0:3])   # force = D u


def test_closed_loop_in_the_reference_two_stage_structure():
    """params["formulation"] = "wrench": 6-D generalized-force MPC with the input hull, then allocation (the reference's
    own structure, spiraling_mpc.py:288-317) in the examples/sim.py scenario (10, 11 stuck fully on).  The loop settles on
    the micro-orbit like the thruster-space loop does; the two differ only by the allocation weight rho the
    thruster-space QP carries (the commands stay within a few percent of f_max of each other)."""
    def run(formulation, **kw):
        m = SystemModel(0.1)
        for i in (10, 11):
            m.set_fault(BrokenThruster(i, 1.0))
        sm = SpiralModel.from_system_model(m)
        extra = {k: kw.pop(k) for k in ("terminal_set",) if k in kw}
        ctrl = SpiralingController(sm, dict(PARAMS, formulation=formulation, **extra), ControllerDebug(), quiet=True, **kw)
        ctrl.load_trajectory("hover", 30)
        us = []

        class Recorder:           # the seam is duck-typed: record what the controller commands
            def get_control(self, x, t):
                us.append(ctrl.get_control(x, t).copy())
                return us[-1]
        env = SimulationEnvironment(m, Recorder(), seed=7)
        env.set_initial_state(**IC)
        for k in env.noise:
            env.noise[k] = 0.0
        for _ in range(60):
            env.step()
        return sm.robot_to_center(env.state), np.array(us), m
    c_w, u_w, m = run("wrench")
    c_t, u_t, _ = run("thruster")
    assert (u_w >= -1e-9).all() and (u_w <= m.u_ub_physical + 1e-9).all() and (u_w[:, [10, 11]] == 0).all()
    e0 = np.linalg.norm(SpiralModel.from_system_model(m).robot_to_center(
        np.r_[IC["position"], IC["velocity"], IC["orientation"], IC["angular_velocity"]])[0:6])
    assert np.linalg.norm(c_w[0:6]) < 0.6 * e0 and np.linalg.norm(c_t[0:6]) < 0.6 * e0      # both close in on the reference
    assert np.abs(c_w[6:9] - [0, 0, 0.6]).max() < 0.1
    assert np.abs(c_w[0:9] - c_t[0:9]).max() < 0.1
    # the reference's full formulation -- hull rows AND the terminal set -- on kernel 11 (fp32 handle): same loop, same place
    c_k, u_k, _ = run("wrench", dtype="f32", terminal_set=True)
    assert (u_k >= -1e-9).all() and (u_k <= m.u_ub_physical + 1e-9).all() and (u_k[:, [10, 11]] == 0).all()
    assert np.abs(c_k[0:9] - c_w[0:9]).max() < 0.1


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-8), ("f32", 2e-5)])      # f32: kernel 11 (an x perturbed by 1e-15 moves an fp32 iterate by 1e-7)
def test_on_device_two_stage_loop_equals_the_step_by_step_loop(gpu_mpc_factory, dtype, tol):
    """ftmpc_simulate_wrench_batch (the reference's structure at every step -- generalized-force MPC with the input hull,
    then allocation -- with plant, noise and the repeat-last warm-start shift on the device) against the same loop driven
    step by step from the host: solve_wrench per step, the oracle's plant and counter-based noise.  Mixed faults, so
    several hull tables are in play."""
    from oracle import closed_loop as cl
    N, NT, B, T = 15, 16, 6, 6
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype=dtype, max_iters=60)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, _ = qo.make_batch(B, N, NT, 2, 8500)
    ub[0] = 3.4; stuck[0] = 0.0                                       # one healthy vehicle among the faulty ones
    xr = _hover_traj(N, T)
    out = mpc.simulate(x0, ub, stuck, xr, T, seed=9, return_inputs=True, formulation="wrench")
    assert out["not_converged"].sum() == 0 and out["alloc_failed"].sum() == 0
    x = x0.copy()
    amp = np.repeat(np.full(4, 1e-3), [3, 3, 4, 3])
    warm = None
    for t in range(T):
        step = mpc.solve_wrench(x, ub, stuck, np.ascontiguousarray(xr[:, t:t + N + 1]).reshape(-1, order="F"), warmG=warm, return_G=True)
        assert (step["status"] == 0).all() and (step["alloc_status"] == 0).all()
        assert np.abs(step["u0"] - out["u"][t]).max() < tol, t
        warm = np.ascontiguousarray(np.concatenate([step["G"][:, 1:], step["G"][:, -1:]], axis=1))
        for b in range(B):
            x[b] = co.plant_step(cfg, x[b], step["u0"][b], ub[b], stuck[b])
        idx = (np.uint64(t) * np.uint64(B) + np.arange(B, dtype=np.uint64))[:, None] * np.uint64(13) + np.arange(13, dtype=np.uint64)[None, :]
        x = x + amp[None, :] * cl.u01(9, idx)
        x[:, 6:10] /= np.linalg.norm(x[:, 6:10], axis=1, keepdims=True)
    assert np.abs(out["x"] - x).max() < tol
    assert (out["u"] >= -1e-9).all() and (out["u"] <= ub[None] + 1e-9).all()


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-6), ("f32", 1e-4)])
def test_on_device_two_stage_loop_against_the_oracle_loop(gpu_mpc_factory, dtype, tol):
    """ftmpc_simulate_wrench_batch against oracle/closed_loop.py:simulate_wrench -- the ORACLE's own two-stage loop (NumPy
    interior point with the active-set polish, NumPy allocator, C plant step, the same counter-based noise): round 3 checked this
    entry only against the same GPU solver driven from the host.  Reference: sim_env.py:77-112 around spiraling_mpc.py:288-317,
    control_allocator.py:65-94.  Per-step thrust commands within the solver's tolerance (the closed loop contracts
    perturbations on these eight steps: measured growth below 2x)."""
    from oracle import closed_loop as cl
    N, NT, B, T = 15, 16, 6, 8
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype=dtype, max_iters=60)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, _ = qo.make_batch(B, N, NT, 2, 8700)
    ub[0] = 3.4; stuck[0] = 0.0
    xr = _hover_traj(N, T)
    out = mpc.simulate(x0, ub, stuck, xr, T, seed=21, return_inputs=True, formulation="wrench")
    xo, uo, _, sto, asto = cl.simulate_wrench(cfg, x0, ub, stuck, xr, T, seed=21)
    assert (sto == 0).all() and (asto == 0).all()
    assert out["not_converged"].sum() == 0 and out["alloc_failed"].sum() == 0
    assert np.abs(out["u"][0] - uo[0]).max() / 3.4 <= tol                      # the first step: same state, the solver's tolerance
    assert np.abs(out["u"] - uo).max() / 3.4 <= 2 * tol, np.abs(out["u"] - uo).max(axis=(1, 2)) / 3.4
    assert np.abs(out["x"] - xo).max() <= 2 * tol


def test_on_device_two_stage_campaign_fp32_against_float64(gpu_mpc_factory):
    """A Monte-Carlo campaign in the reference's two-stage structure on kernel 11 (fp32 handle): 2 048 vehicles with random
    double faults, 30 closed-loop steps with measurement noise -- every step of every vehicle converges and allocates, and the
    trajectories stay with the float64 kernel's (same loop, same noise)."""
    N, NT, B, T = 15, 16, 2048, 30
    x0, ub, stuck, _ = qo.make_batch(B, N, NT, 2, 8600)
    from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
    hull = hull_tables(qo.QPConfig(N=N, NT=NT).D, ub, stuck)
    ok = ~np.asarray(hull["degenerate"], bool)
    x0, ub, stuck = x0[ok], ub[ok], stuck[ok]
    hull = hull_tables(qo.QPConfig(N=N, NT=NT).D, ub, stuck)
    xr = _hover_traj(N, T)
    res = {}
    for dt in ("f32", "f64"):
        mpc = gpu_mpc_factory(N=N, NT=NT, dtype=dt, max_iters=60)
        res[dt] = mpc.simulate(x0, ub, stuck, xr, T, seed=31, return_inputs=True, formulation="wrench", hull=hull)
        assert res[dt]["not_converged"].sum() == 0, (dt, res[dt]["not_converged"])
        assert res[dt]["alloc_failed"].sum() == 0, (dt, res[dt]["alloc_failed"])
        assert np.isfinite(res[dt]["x"]).all()
    assert (res["f32"]["u"] >= -1e-9).all() and (res["f32"]["u"] <= ub[None] + 1e-9).all()
    # 1e-4 f_max per step; perturbations grow over 30 steps of an unstable plant under feedback
    assert np.abs(res["f32"]["u"] - res["f64"]["u"]).max() / 3.4 < 5e-3
    assert np.median(np.abs(res["f32"]["u"] - res["f64"]["u"]).max(axis=(0, 2))) / 3.4 < 2e-4
    assert np.abs(res["f32"]["x"] - res["f64"]["x"]).max() < 5e-3
