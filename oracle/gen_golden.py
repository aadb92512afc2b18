"""Generates tests/golden/*.npz.  Run from the repo root in the BUILD container (needs
/root/reference):   python -m oracle.gen_golden

Two kinds of fixtures:
 (1) REFERENCE-DERIVED (pin the oracle): produced by importing the reference modules that
     import here (ft_mpc.util.get_trajectory, ft_mpc.util.broken_thruster -- plain numpy/scipy)
     and by reading the reference's DATA files (ft_mpc/config/terminal.yaml,
     ft_mpc/config/reactive.yaml, data/InertialProperties.md's D table, util/animate.py's thruster
     geometry table as numbers).  The reference controller itself cannot be imported
     (casadi/cvxpy/polytope missing: ordinary ModuleNotFoundError).
 (2) ORACLE-DERIVED (pin the GPU path on the box, where /root/reference does not exist): seeded
     instances of every BASELINE config with the exact QP solution from scipy BVLS.
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"


def reference_fixtures():
    sys.path.insert(0, str(REF))
    from ft_mpc.util.get_trajectory import load_trajectory  # importable: numpy/scipy/yaml only
    from ft_mpc.util.broken_thruster import BrokenThruster
    import yaml
    from scipy.spatial.transform import Rotation as R

    out = {}
    # --- trajectories (get_trajectory.py:43-184), dt = 0.1 ---
    hov = load_trajectory("hover", 0.1, 30)
    out["hover_shape"] = np.array(hov.shape)
    out["hover_cols"] = hov[:, [0, 1, 1500, 2999]]
    h123 = load_trajectory("hover_1_2_3", 0.1, 5)
    out["hover123_shape"] = np.array(h123.shape)
    out["hover123_col"] = h123[:, 7]
    circ = load_trajectory("circle_r_0.65_sPerFullCircle_40", 0.1, 30)
    out["circle_shape"] = np.array(circ.shape)
    out["circle_cols"] = circ[:, :64]
    out["circle_sum"] = np.array([circ.sum(), np.abs(circ).sum()])
    # --- fault record (broken_thruster.py) ---
    bt = BrokenThruster(10, 1.0)
    out["bt"] = np.array([bt.index, bt.intensity])
    # --- sim.py:49-54 initial condition ---
    out["ic_quat"] = R.from_euler("zyx", [50, 30, -10], degrees=True).as_quat()
    # --- reactive.yaml (data) ---
    ry = yaml.safe_load(open(REF / "ft_mpc/config/reactive.yaml"))
    sp = ry["tuning"]["spiraling"]
    out["yaml_dt"] = np.array([ry["time_step"]])
    out["yaml_horizon"] = np.array([sp["horizon"]])
    out["yaml_Q"] = np.array(sp[sp["param_set"]]["Q"], float)
    out["yaml_R"] = np.array(sp[sp["param_set"]]["R"], float)
    out["yaml_faults"] = np.array([[f["act_id"], f["intensity"], f["start_time"]] for f in ry["actuator_failures"]], float)
    # --- terminal.yaml (data): cost evaluated with sympy only, set as numbers ---
    from oracle import refmath as rm
    cost, P, A, b = rm.parse_terminal_yaml(open(REF / "ft_mpc/config/terminal.yaml").read())
    pts = np.array([[0] * 9, [0.1] * 9, [.5, -.2, .1, .05, 0, -.05, .1, -.1, .05]], float)
    out["term_points"] = pts
    out["term_cost"] = np.array([cost(p) for p in pts])
    out["term_P"] = P
    out["term_A"] = A
    out["term_b"] = b
    # --- D, three ways: the markdown table (data/InertialProperties.md:30-41) and r x F from
    #     the geometry table of util/animate.py:66-110 (numbers transcribed by regex, not code) ---
    md = (REF / "data/InertialProperties.md").read_text()
    blk = md[md.index("### 3D Spacecraft"):]
    body = blk[blk.index("smallmatrix}") + 12: blk.index(r"\end{smallmatrix}")]
    sym = {"a": 0.12, "b": 0.09, "c": 0.05}
    Dmd = []
    for line in body.splitlines():
        ent = [e.strip() for e in line.replace("\\", " ").split("&")]
        if len(ent) != 16:
            continue
        row = []
        for e in ent:
            sgn = -1.0 if e.startswith("-") else 1.0
            key = e.lstrip("-").strip()
            row.append(sgn * sym[key] if key in sym else float(e))
        Dmd.append(row)
    out["D_md"] = np.array(Dmd, float)
    an = (REF / "ft_mpc/util/animate.py").read_text()
    dvals = {k: float(v) for k, v in re.findall(r"^\s*(d[1-4]) = ([\d.]+)", an, re.M)}

    def table(name):
        body = an[an.index(name + " = {"):]
        body = body[:body.index("}")]
        res = []
        for m in re.finditer(r"\d+:\s*\(([^)]*)\)", body):
            vals = []
            for tok in m.group(1).split(","):
                tok = tok.strip()
                sgn = -1.0 if tok.startswith("-") else 1.0
                tok = tok.lstrip("-").strip()
                vals.append(sgn * (dvals[tok] if tok in dvals else float(tok)))
            res.append(vals)
        return np.array(res, float)

    pos, dirs = table("thruster_positions"), table("thruster_directions")
    Dg = np.zeros((6, 16))
    for i in range(16):
        f = -dirs[i]  # exhaust direction -> thrust on the body
        Dg[0:3, i] = f
        Dg[3:6, i] = np.cross(pos[i], f)
    out["D_geom"] = Dg
    np.savez_compressed(OUT / "reference_pins.npz", **out)
    print("reference_pins.npz:", {k: v.shape for k, v in out.items()})


def sim_env_fixture():
    """The reference's own SimulationEnvironment (ft_mpc/simulation/sim_env.py:11-112; imports without
    casadi) driving the mirror's duck-typed SystemModel + dummy controller for 1 s with the noise switched
    off: pins set_initial_state, the step order (control -> dynamics -> noise -> renormalise) and the
    loop count of run_simulation.  tests/test_host_mirror.py compares the mirror's loop with it."""
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(ROOT / "fault-tolerant-mpc_amd"))   # (this one fixture pins the host mirror's seam classes under the REFERENCE's loop)
    from ft_mpc.simulation.sim_env import SimulationEnvironment as RefEnv
    from ft_mpc_amd.controllers.dummy_controller import Controller
    from ft_mpc_amd.models.sys_model import SystemModel
    from ft_mpc_amd.util.controller_debug import ControllerDebug
    from scipy.spatial.transform import Rotation as R

    q = R.from_euler("zyx", [50, 30, -10], degrees=True).as_quat()
    out = {}
    for name, dur in (("1s", 1.0), ("2p5s", 2.5)):
        m = SystemModel(0.1)
        env = RefEnv(m, Controller(m, ControllerDebug()))
        env.set_initial_state(position=[1, 0, 1], velocity=[1, .5, 0], orientation=q, angular_velocity=[.3, .8, -.1])
        x_init = np.array(env.state, float).reshape(-1).copy()
        for k in env.noise:
            env.noise[k] = 0.0
        env.run_simulation(dur)
        out["state_" + name] = np.array(env.state, float).reshape(-1)
        out["init_" + name] = x_init
    np.savez_compressed(OUT / "sim_env_pin.npz", **out)
    print("sim_env_pin.npz:", {k: v.shape for k, v in out.items()})


def qp_fixtures():
    from oracle import qp_oracle as qo
    from oracle import refmath as rm

    specs = {
        # name: (N, NT, nfault, seed, count, warm, uref)
        "cfg2_single_fault": (20, 8, 1, 1002, 16, False, False),
        "cfg3_double_fault": (20, 8, 2, 1003, 24, False, False),
        "cfg3_warm_uref": (20, 8, 2, 1013, 8, True, True),
        "nominal_nt8": (20, 8, 0, 1001, 8, False, False),
        "short_horizon": (5, 8, 2, 1021, 8, False, False),
    }
    for name, (N, NT, nf, seed, cnt, warm, uref) in specs.items():
        cfg = qo.QPConfig(N=N, NT=NT)
        x0, ub, stuck, xref = qo.make_batch(cnt, N, NT, nf, seed)
        rng = np.random.default_rng(seed + 7)
        W = rng.uniform(-0.3, 3.7, (cnt, N, NT)) if warm else None
        ur = None
        if uref:
            # a circle-like reference window (get_trajectory.py:125-141 shape) for xref/uref
            traj = rm.circle_trajectory(0.1, 5, radius=0.65, s_per_circle=40.0)
            xr_all, ur_all = rm.assign_trajectory(traj, N)
            xref, ur = rm.trajectory_window(xr_all, ur_all, 1.0, N)
        U = np.zeros((cnt, N, NT))
        u0 = np.zeros((cnt, NT))
        kkt = np.zeros(cnt)
        for b in range(cnt):
            u0[b], U[b], qp = qo.solve_instance(cfg, x0[b], ub[b], stuck[b], xref, uref=ur,
                                                warmU=None if W is None else W[b], exact=True)
            d = U[b][:, qp["act"]].reshape(-1) - qp["Ubar"]
            kkt[b] = qo.kkt_residual(qp["H"], qp["g"], -qp["Ubar"], qp["ub"] - qp["Ubar"], d)
        assert kkt.max() < 1e-9, kkt
        np.savez_compressed(OUT / f"qp_{name}.npz", N=N, NT=NT, x0=x0, ub=ub, stuck=stuck, xref=xref,
                            uref=np.zeros(0) if ur is None else ur, warm=np.zeros(0) if W is None else W,
                            U=U, u0=u0, kkt=kkt, D=cfg.D, rho=cfg.rho)
        print(f"qp_{name}.npz  kkt max {kkt.max():.1e}")


def qp_fixtures_large():
    """Exact (BVLS) solutions of the thruster-space QP for the shapes beyond the 8-thruster benchmark: the reference
    vehicle at its shipped horizon, its N = 20 variant, and BASELINE config 5 (N = 40, 16 thrusters, two faults); 16 each."""
    from oracle import qp_oracle as qo
    specs = {"refvehicle_n15": (15, 16, 2, 1011, 16), "refvehicle_n20": (20, 16, 2, 1013, 16), "cfg5_n40_nt16": (40, 16, 2, 1005, 16)}
    for name, (N, NT, nf, seed, cnt) in specs.items():
        cfg = qo.QPConfig(N=N, NT=NT)
        x0, ub, stuck, xref = qo.make_batch(cnt, N, NT, nf, seed)
        U = np.zeros((cnt, N, NT))
        u0 = np.zeros((cnt, NT))
        kkt = np.zeros(cnt)
        for b in range(cnt):
            u0[b], U[b], qp = qo.solve_instance(cfg, x0[b], ub[b], stuck[b], xref, exact=True)
            d = U[b][:, qp["act"]].reshape(-1) - qp["Ubar"]
            kkt[b] = qo.kkt_residual(qp["H"], qp["g"], -qp["Ubar"], qp["ub"] - qp["Ubar"], d)
        assert kkt.max() < 1e-8, kkt
        np.savez_compressed(OUT / f"qp_{name}.npz", N=N, NT=NT, x0=x0, ub=ub, stuck=stuck, xref=xref, uref=np.zeros(0), warm=np.zeros(0),
                            U=U, u0=u0, kkt=kkt, D=cfg.D, rho=cfg.rho)
        print(f"qp_{name}.npz  kkt max {kkt.max():.1e}", flush=True)


def _near_terminal_set(x0, At, bt, scale, seed):
    """Moves the states so that the orbit-centre tracking error starts on `scale` times the boundary of the terminal set."""
    from oracle import refmath as rm
    rng = np.random.default_rng(seed)
    r = rm.spiral_r()
    for b in range(x0.shape[0]):
        e = rng.standard_normal(9)
        e *= scale / max((At @ e / bt).max(), 1e-9)
        R = rm.rot(x0[b, 6:10])
        w = rm.OMEGA_DES + e[6:9]
        x0[b, 0:3] = e[0:3] - R.T @ r                    # robot_to_center (spiral_model.py:103-109) inverted
        x0[b, 3:6] = e[3:6] - R.T @ np.cross(w, r)
        x0[b, 10:13] = w


def _certified(qo, qp, st):
    """max KKT residual of the (polished) oracle solution and its distance to the active-set certificate (f_max)."""
    if st != 0:
        return np.nan, np.nan
    cert = max(qo.kkt_general(qp["H"], qp["g"], qp["C"], qp["h"], qp["d"], qp["z"]))
    dx, _ = qo.solve_general_exact(qp["H"], qp["g"], qp["C"], qp["h"], qp["d"], qp["z"], qp["s"])
    return cert, (np.abs(dx - qp["d"]).max() / 3.4 if dx is not None else np.nan)


def general_constraint_fixtures():
    """The reference's own formulation: 6-D generalized-force QP with the input-hull rows
    (spiraling_mpc.py:133-137,175-177), the thruster-space QP with the 72-row terminal set (:199-202), and the two together
    (hull rows + terminal set: what the reference's NLP always carries), solved by oracle/qp_oracle.py:ipm_general WITH its
    active-set polish and kept only where the solver-independent certificates hold (kkt_general; distance to the
    primal-dual active-set solution solve_general_exact); unreachable terminal sets are recorded as such (status != 0).
    The terminal set is read from the REFERENCE's data file by the oracle's own parser (no product code in here)."""
    from oracle import qp_oracle as qo
    from oracle import refmath as rm
    N, NT, cnt = 15, 16, 16
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(cnt, N, NT, 1, 7015)
    G = np.zeros((cnt, N, 6))
    tau0 = np.zeros((cnt, 6))
    st = np.zeros(cnt, np.int32)
    cert = np.zeros(cnt)
    dist = np.zeros(cnt)
    for b in range(cnt):
        tau0[b], G[b], st[b], _, qp = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref)
        cert[b], dist[b] = _certified(qo, qp, st[b])
    assert (st == 0).all() and cert.max() < 1e-8 and dist.max() < 1e-8, (st, cert, dist)
    np.savez_compressed(OUT / "qp_wrench_hull_n15.npz", N=N, NT=NT, x0=x0, ub=ub, stuck=stuck, xref=xref, G=G, tau0=tau0, status=st, kkt=cert, D=cfg.D)
    print(f"qp_wrench_hull_n15.npz  kkt max {cert.max():.1e}  vs active-set certificate {dist.max():.1e} f_max", flush=True)
    # terminal set: states placed around the boundary of the set, so that reachable and unreachable instances both occur
    _, _, At, bt = rm.parse_terminal_yaml(open(REF / "ft_mpc/config/terminal.yaml").read())
    N, NT, cnt = 20, 8, 24
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(cnt, N, NT, 2, 12)
    _near_terminal_set(x0, At, bt, 2.0, 13)
    U = np.zeros((cnt, N, NT))
    st = np.zeros(cnt, np.int32)
    cert = np.zeros(cnt)
    dist = np.zeros(cnt)
    with np.errstate(all="ignore"):
        for b in range(cnt):
            _, U[b], st[b], _, qp = qo.solve_box_terminal_instance(cfg, x0[b], ub[b], stuck[b], xref, (At, bt), iters=60)
            cert[b], dist[b] = _certified(qo, qp, st[b])
    assert (st == 0).sum() >= 6 and np.nanmax(cert) < 1e-7 and np.nanmax(dist) < 1e-8, (st, cert, dist)
    np.savez_compressed(OUT / "qp_terminal_set_n20.npz", N=N, NT=NT, x0=x0, ub=ub, stuck=stuck, xref=xref, U=U, status=st, kkt=cert,
                        term_A=At, term_b=bt, D=cfg.D, rho=cfg.rho)
    print(f"qp_terminal_set_n20.npz  reachable {(st == 0).sum()} of {cnt}, kkt max {np.nanmax(cert):.1e}  vs certificate {np.nanmax(dist):.1e}", flush=True)
    # hull rows + terminal set (the reference's own NLP structure, spiraling_mpc.py:175-177 with :199-202), N = 15, 16 thrusters:
    # tracking error on the boundary of the set, so that hull and terminal rows are active together in many instances
    N, NT, cnt = 15, 16, 32
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(cnt, N, NT, 2, 9200)
    _near_terminal_set(x0, At, bt, 1.0, 9201)
    G = np.zeros((cnt, N, 6))
    st = np.zeros(cnt, np.int32)
    cert = np.zeros(cnt)
    dist = np.zeros(cnt)
    act_h = np.zeros(cnt, np.int32)
    act_t = np.zeros(cnt, np.int32)
    with np.errstate(all="ignore"):
        for b in range(cnt):
            try:
                _, G[b], st[b], _, qp = qo.solve_wrench_instance(cfg, x0[b], ub[b], stuck[b], xref, term_set=(At, bt), iters=60)
            except ValueError:       # flat hull (healthy thrusters do not span R^6)
                st[b], cert[b], dist[b] = 3, np.nan, np.nan
                continue
            cert[b], dist[b] = _certified(qo, qp, st[b])
            if st[b] == 0:
                act_h[b] = (qp["z"][:qp["nhull"]] > 0).sum()
                act_t[b] = (qp["z"][qp["nhull"]:] > 0).sum()
    assert (st == 0).sum() >= 16 and np.nanmax(cert) < 1e-7 and np.nanmax(dist) < 1e-8 and ((act_h > 0) & (act_t > 0)).sum() >= 6, (st, cert, dist)
    np.savez_compressed(OUT / "qp_wrench_hull_terminal_n15.npz", N=N, NT=NT, x0=x0, ub=ub, stuck=stuck, xref=xref, G=G, status=st, kkt=cert,
                        active_hull=act_h, active_term=act_t, term_A=At, term_b=bt, D=cfg.D)
    print(f"qp_wrench_hull_terminal_n15.npz  reachable {(st == 0).sum()} of {cnt}, hull and terminal rows active together in "
          f"{((act_h > 0) & (act_t > 0)).sum()}, kkt max {np.nanmax(cert):.1e}  vs certificate {np.nanmax(dist):.1e}", flush=True)


def state_bound_fixture():
    """The reference's optional state bounds (spiraling_mpc.py:129-130,179-185: xlb <= x_t <= xub on the stages t < N) in the
    thruster-space QP: velocity and angular-rate bounds that bite on most instances, solved by oracle/qp_oracle.py:ipm_general with
    its polish and certified (KKT residuals, distance to the primal-dual active-set solution)."""
    from oracle import qp_oracle as qo
    N, NT, cnt = 20, 8, 24
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(cnt, N, NT, 2, 4321)
    xub = np.full(13, np.inf)
    xlb = np.full(13, -np.inf)
    xub[3:6], xlb[3:6] = 0.9, -0.9            # |v| <= 0.9 m/s
    xub[6:9], xlb[6:9] = 1.6, -1.6            # |omega| <= 1.6 rad/s
    U = np.zeros((cnt, N, NT))
    st = np.zeros(cnt, np.int32)
    cert = np.full(cnt, np.nan)
    dist = np.full(cnt, np.nan)
    act = np.zeros(cnt, np.int32)
    with np.errstate(all="ignore"):
        for b in range(cnt):
            _, U[b], st[b], _, qp = qo.solve_box_state_instance(cfg, x0[b], ub[b], stuck[b], xref, xlb, xub, iters=60)
            if st[b] == 0:
                cert[b], dist[b] = _certified(qo, qp, 0)
                act[b] = (qp["z"][qp["nhull"]:] > 0).sum()
    assert (st == 0).sum() >= 12 and (act > 0).sum() >= 8 and np.nanmax(cert) < 1e-7 and np.nanmax(dist) < 1e-8, (st, act, cert, dist)
    np.savez_compressed(OUT / "qp_state_bounds_n20.npz", N=N, NT=NT, x0=x0, ub=ub, stuck=stuck, xref=xref, xlb=xlb, xub=xub, U=U, status=st, kkt=cert,
                        active_rows=act, D=cfg.D, rho=cfg.rho)
    print(f"qp_state_bounds_n20.npz  solved {(st == 0).sum()} of {cnt}, with active state rows {(act > 0).sum()}, kkt max {np.nanmax(cert):.1e}  vs certificate {np.nanmax(dist):.1e}", flush=True)


if __name__ == "__main__":
    OUT.mkdir(parents=True, exist_ok=True)
    if "--only-state-bounds" in sys.argv:
        state_bound_fixture()
        sys.exit(0)
    if "--only-general" not in sys.argv:
        if "--only-round3" not in sys.argv:
            reference_fixtures()
            sim_env_fixture()
            qp_fixtures()
        qp_fixtures_large()
    general_constraint_fixtures()
    state_bound_fixture()
