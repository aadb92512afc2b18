"""GPU parity of the reference's optional STATE BOUNDS (spiraling_mpc.py:129-130,179-185: xlb <= x_t <= xub on the orbit-centre state
of the stages t < N; controller params "xub" / "xlb").  The HIP path adds them to the thruster-space QP on the Riccati kernel
(ftmpc_solve_ric64_kernel<*, true>: a diagonal barrier term on the state weight of their stage); the checker is
oracle/qp_oracle.py:solve_box_state_instance -- the same rows written out DENSE through the sensitivities, ipm_general with its
active-set polish, certified by KKT residuals: the EXACT solution.  Tolerance: north_star's 1e-4 f_max on the whole-horizon
thruster forces -- this kernel stops its interior-point iteration at mu 1e-10 (the barrier weight of an active state row enters
the Riccati recursion's state weight: beyond ~1e-11 the recursion runs out of float64 digits) and has no polish yet, so weakly
active rows leave it up to a few 1e-5 f_max from the exact solution, as the other general-constraint modes were before round 4
gave them theirs; float64 whatever the handle's dtype."""
from pathlib import Path

import numpy as np
import pytest

from oracle import qp_oracle as qo
from oracle import refmath as rm

pytestmark = pytest.mark.gpu
F_MAX = rm.F_MAX
GOLD = Path(__file__).parent / "golden"
TOL = 1e-4


def _bounds(v=0.9, w=1.6):
    xub, xlb = np.full(13, np.inf), np.full(13, -np.inf)
    xub[3:6], xlb[3:6] = v, -v
    xub[6:9], xlb[6:9] = w, -w
    return xlb, xub


@pytest.mark.parametrize("N,NT,dtype,B", [(20, 8, "f32", 24), (15, 16, "f64", 16), (33, 16, "f64", 8)])
def test_state_bounds_against_the_oracle(gpu_mpc_factory, N, NT, dtype, B):
    xlb, xub = _bounds()
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype=dtype, max_iters=60, xlb=xlb, xub=xub)
    free = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=60)
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 4400 + N)
    xr = xref.reshape(-1, order="F")
    out = mpc.solve(x0, ub, stuck, xr, return_U=True)
    ref = free.solve(x0, ub, stuck, xr, return_U=True)
    solved = active = 0
    with np.errstate(all="ignore"):
        for b in range(B):
            _, U, st, _, qp = qo.solve_box_state_instance(cfg, x0[b], ub[b], stuck[b], xref, xlb, xub, iters=60)
            assert (out["status"][b] == 0) == (st == 0), (b, out["status"][b], st)
            assert np.isfinite(out["U"][b]).all()
            if st != 0:
                continue
            solved += 1
            assert max(qo.kkt_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d"], qp["z"])) < 1e-7
            assert np.abs(out["U"][b] - U).max() / F_MAX <= TOL, (b, np.abs(out["U"][b] - U).max() / F_MAX)
            # the bounds hold along the predicted (linearised) trajectory
            assert (qp["C"][qp["nhull"]:] @ qp["d"] <= qp["h"][qp["nhull"]:] + 1e-8).all()
            na = int((qp["z"][qp["nhull"]:] > 0).sum())
            active += int(na > 0)
            if na > 0:
                assert np.abs(out["U"][b] - ref["U"][b]).max() / F_MAX > 1e-5      # the rows matter
    assert solved >= B // 3 and active >= 2 and (solved < B or N != 20)      # bounds that bite; at N = 20 also bounds that cannot be met


def test_state_bounds_against_golden(gpu_mpc_factory):
    d = np.load(GOLD / "qp_state_bounds_n20.npz")
    mpc = gpu_mpc_factory(N=int(d["N"]), NT=int(d["NT"]), dtype="f32", max_iters=60, xlb=d["xlb"], xub=d["xub"])
    out = mpc.solve(d["x0"], d["ub"], d["stuck"], d["xref"].reshape(-1, order="F"), return_U=True)
    ok = d["status"] == 0
    assert np.array_equal(out["status"] == 0, ok) and ok.sum() >= 12 and (d["active_rows"][ok] > 0).sum() >= 8
    assert np.abs(out["U"][ok] - d["U"][ok]).max() / F_MAX <= TOL
    assert np.isfinite(out["U"]).all() and (out["U"] >= -1e-12).all() and (out["U"] <= d["ub"][:, None, :] + 1e-9).all()


def test_no_finite_bound_is_the_plain_problem_and_one_sided_bounds(gpu_mpc_factory):
    """All bounds infinite: the state-bound instantiation returns the plain kernel's solution (no rows); an upper bound alone
    (the reference fills the missing side with infinities, spiraling_mpc.py:181-182) against the oracle."""
    N, NT, B = 20, 8, 12
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 1, 4500)
    xr = xref.reshape(-1, order="F")
    plain = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=60).solve(x0, ub, stuck, xr, return_U=True)
    none = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=60, xub=np.full(13, np.inf)).solve(x0, ub, stuck, xr, return_U=True)
    assert (none["status"] == 0).all() and np.abs(none["U"] - plain["U"]).max() / F_MAX <= 2e-5      # (mu 1e-10 against 1e-13)
    xub = np.full(13, np.inf)
    xub[3:6] = 1.2
    one = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=60, xub=xub).solve(x0, ub, stuck, xr, return_U=True)
    cfg = qo.QPConfig(N=N, NT=NT)
    with np.errstate(all="ignore"):
        for b in range(B):
            _, U, st, _, _ = qo.solve_box_state_instance(cfg, x0[b], ub[b], stuck[b], xref, None, xub, iters=60)
            assert (one["status"][b] == 0) == (st == 0)
            if st == 0:
                assert np.abs(one["U"][b] - U).max() / F_MAX <= TOL


def test_controller_params_xub_xlb(gpu_mpc_factory):
    """The mirror's SpiralingController takes the reference's own keys params["xub"] / params["xlb"]."""
    import yaml
    from ft_mpc_amd.controllers.spiraling_mpc import SpiralingController
    from ft_mpc_amd.models.sys_model import SystemModel
    from ft_mpc_amd.util.controller_debug import ControllerDebug
    from ft_mpc_amd import _lib
    params = yaml.safe_load(open(Path(_lib.__file__).parent / "config" / "reactive.yaml"))["tuning"]["spiraling"]
    xlb, xub = _bounds(0.5, 1.2)
    params = dict(params, xub=xub, xlb=xlb)
    m = SystemModel(0.1)
    ctl = SpiralingController(m, params, ControllerDebug(), quiet=True)
    ctl.load_trajectory("hover", 10)
    x = np.array([1.0, 0.0, 1.0, 0.3, 0.1, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.6])
    u = ctl.get_control(x, 0.0)
    assert u.shape == (16,) and np.isfinite(u).all() and (u >= -1e-9).all() and (u <= 3.4 + 1e-9).all()
    ctl.mpc.close()
