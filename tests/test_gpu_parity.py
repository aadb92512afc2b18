"""GPU parity tests proper: the HIP path (through the C-ABI) against the float64 oracle on the
same seeded inputs.  Tolerances:
  * condensed QP data (H, g built in fp32 on the GPU):   rel-inf 2e-5 on H, 2e-5*scale on g
  * thrust command u0 vs the EXACT QP solution (BVLS):   <= 1e-4 * f_max  (north-star tolerance)
  * the whole-horizon solution U vs the same:             <= 1e-4 * f_max  (TOL_U; measured on the full 65 536-instance
    config-3 batch: 2.8e-5 -- rounds 1-3 accepted 2e-3 here, which would have hidden a real regression)
"""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import qp_oracle as qo
from oracle import refmath as rm

pytestmark = pytest.mark.gpu
F_MAX = rm.F_MAX
TOL_U = 1e-4


def _cfg(N, NT, rho=0.05):
    return qo.QPConfig(N=N, NT=NT, rho=rho)


@pytest.mark.parametrize("nfault", [0, 1, 2])
def test_build_matches_oracle(gpu_mpc_factory, nfault):
    N, NT = 20, 8
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(8, N, NT, nfault, 2000 + nfault)
    cfg = _cfg(N, NT)
    for inst in (0, 3, 7):
        H, g, lo, hi = mpc.debug_build_qp(x0, ub, stuck, xref.reshape(-1, order="F"), inst)
        qp = qo.build_qp(cfg, x0[inst], ub[inst], stuck[inst], xref)
        assert H.shape == qp["H"].shape
        scale = np.abs(qp["H"]).max()
        assert np.abs(H - qp["H"]).max() <= 2e-5 * scale
        assert np.abs(g - qp["g"]).max() <= 2e-5 * max(1.0, np.abs(qp["g"]).max())
        assert np.allclose(lo, -qp["Ubar"], atol=1e-6)
        assert np.allclose(hi, qp["ub"] - qp["Ubar"], atol=1e-6)


@pytest.mark.parametrize("nfault,B", [(2, 48), (1, 16), (0, 16)])
def test_u0_matches_exact_solution(gpu_mpc_factory, nfault, B):
    N, NT = 20, 8
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nfault, 1003)
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    assert (out["status"] == 0).all(), out["status"]
    cfg = _cfg(N, NT)
    err = np.zeros(B)
    for b in range(B):
        u0, U, _ = qo.solve_instance(cfg, x0[b], ub[b], stuck[b], xref, exact=True)
        err[b] = np.abs(out["u0"][b] - u0).max() / F_MAX
        assert np.abs(out["U"][b] - U).max() / F_MAX <= TOL_U
    assert err.max() <= 1e-4, err
    assert (out["u0"][ub == 0] == 0).all()
    assert out["iters"].max() <= 30


@pytest.mark.parametrize("nfault,B", [(2, 16384), (1, 4096), (0, 4096)])
def test_large_batches_against_the_c_oracle(gpu_mpc_factory, nfault, B):
    """Whole batches through every instantiation (NB = 8 / 9 / 10), every persistent workgroup looping over many
    instances: scheduling-dependent faults (an instruction hazard inside inline asm once corrupted a few instances
    per ten thousand) only show up at this size."""
    import os
    N, NT = 20, 8
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nfault, 4100 + nfault)
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    assert (out["status"] == 0).all(), np.bincount(out["status"])
    ref = co.solve_batch(_cfg(N, NT), x0, ub, stuck, xref, nthreads=min(32, os.cpu_count() or 1), max_iters=60, mu_stop=1e-13)
    assert (ref["status"] == 0).all()
    # (north_star's tolerance is 1e-4; since the polish takes its float64 gradient between its two steps the worst of 230 000 instances
    # over five batches is 1.8e-5 -- an instance that went back to the interior-point iteration --, 99.9 % are within 7e-7)
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX <= 3e-5
    assert np.abs(out["U"] - ref["U"]).max() / F_MAX <= 5e-5


@pytest.mark.parametrize("N,nfault", [(3, 2), (6, 0), (18, 2), (23, 2), (26, 2)])
def test_horizons_not_multiple_of_four(gpu_mpc_factory, N, nfault):
    """The reference-gradient sweeps run four stages per round with a scalar tail, and the three
    instantiations split at n = 128 / 144 / 160: horizons on every side of those seams, against the
    C oracle converged to mu 1e-13 (N=26 with six healthy thrusters: n = 156, the NB=10 kernel)."""
    NT, B = 8, 24
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nfault, 3100 + N)
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    assert (out["status"] == 0).all(), out["status"]
    ref = co.solve_batch(_cfg(N, NT), x0, ub, stuck, xref, nthreads=4, max_iters=60, mu_stop=1e-13)
    assert (ref["status"] == 0).all()
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX <= 1e-4
    assert np.abs(out["U"] - ref["U"]).max() / F_MAX <= TOL_U


@pytest.mark.parametrize("nfault", [0, 2])
def test_sixteen_thrusters_short_horizon_on_the_fp32_kernels(gpu_mpc_factory, nfault):
    """The reference's own 16-thruster allocation matrix with a horizon short enough for the register-resident
    fp32 kernels (N = 10: n = 160 nominal -> NB = 10, n = 140 with two faults -> NB = 9)."""
    N, NT, B = 10, 16, 24
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nfault, 3300 + nfault)
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    assert (out["status"] == 0).all(), out["status"]
    ref = co.solve_batch(_cfg(N, NT), x0, ub, stuck, xref, nthreads=4, max_iters=60, mu_stop=1e-13)
    assert (ref["status"] == 0).all()
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX <= 1e-4
    assert np.abs(out["U"] - ref["U"]).max() / F_MAX <= TOL_U
    assert (out["u0"][ub == 0] == 0).all()


# ---------------------------------------------------------------------------------------------
# fp32 workgroup kernel with the factor in LDS (160 < n <= 240: the reference vehicle at its shipped horizon)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nfault", [0, 1, 2])
def test_wg_kernel_build_matches_oracle(gpu_mpc_factory, nfault):
    N, NT = 15, 16
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(6, N, NT, nfault, 2100 + nfault)
    cfg = _cfg(N, NT)
    for inst in (0, 5):
        H, g, lo, hi = mpc.debug_build_qp(x0, ub, stuck, xref.reshape(-1, order="F"), inst)
        qp = qo.build_qp(cfg, x0[inst], ub[inst], stuck[inst], xref)
        assert H.shape == qp["H"].shape and H.shape[0] == N * (NT - nfault)
        assert np.abs(H - qp["H"]).max() <= 2e-5 * np.abs(qp["H"]).max()
        assert np.abs(g - qp["g"]).max() <= 2e-5 * max(1.0, np.abs(qp["g"]).max())
        assert np.allclose(lo, -qp["Ubar"], atol=1e-6) and np.allclose(hi, qp["ub"] - qp["Ubar"], atol=1e-6)


@pytest.mark.parametrize("sel", ["auto", "workgroup", "dense"])
@pytest.mark.parametrize("N,nfault,B", [(15, 2, 64), (15, 0, 32), (15, 1, 32), (12, 0, 32), (13, 2, 32)])
def test_wg_kernel_u0_against_the_c_oracle(gpu_mpc_factory, N, nfault, B, sel):
    """n = 210 (reactive.yaml's horizon, two faults), 240 (nominal), 225, 192 and 182: block counts 12..15.
    auto: kernel 10 (the QP through wrench space on one wave per instance, the default for these shapes); workgroup: kernel 8
    (the same form on a 4-wave workgroup); dense: kernel 7."""
    NT = 16
    mpc = gpu_mpc_factory(N=N, NT=NT, kernel_select=sel)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nfault, 3400 + N + nfault)
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    assert (out["status"] == 0).all(), out["status"]
    ref = co.solve_batch(_cfg(N, NT), x0, ub, stuck, xref, nthreads=8, max_iters=60, mu_stop=1e-13)
    assert (ref["status"] == 0).all()
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX <= 1e-4
    assert np.abs(out["U"] - ref["U"]).max() / F_MAX <= TOL_U
    assert (out["u0"][ub == 0] == 0).all() and out["iters"].max() <= 30


@pytest.mark.parametrize("sel", ["auto", "workgroup"])
@pytest.mark.parametrize("N,NT,seed", [(11, 16, 1), (16, 16, 2), (14, 13, 3), (16, 12, 4), (21, 11, 5), (20, 11, 6), (13, 15, 7),
                                       (20, 16, 8), (21, 16, 9), (18, 14, 10)])
def test_wrench_space_kernel_other_vehicles_and_horizons(gpu_mpc_factory, N, NT, seed, sel):
    """Kernels 10 (auto: one wave per instance) and 8 (workgroup per instance) away from the reference shape: 11..16 thrusters (a random allocation matrix of full row rank), horizons
    11..21 (both instantiations: six and eight tiles a side; N * NT up to 336: two thruster variables per thread), mixed
    fault counts, warm start; against a reference for EVERY instance (C oracle, BVLS for what it does not finish).
    All of them within the 1e-4 f_max of the specification: the eight-tile instantiation takes the float64 reference
    gradient at every iterate below mu = 1e-3 (these synthetic vehicles have cond(H) ~ 1e7)."""
    rng = np.random.default_rng(900 + seed)
    B = 96
    D = None
    if NT != 16:
        D = rng.standard_normal((6, NT)) * np.array([1, 1, 1, 0.3, 0.3, 0.3])[:, None]
    cfg = _cfg(N, NT) if D is None else qo.QPConfig(N=N, NT=NT, D=D)
    mpc = gpu_mpc_factory(N=N, NT=NT, kernel_select=sel) if D is None else gpu_mpc_factory(N=N, NT=NT, D=D, kernel_select=sel)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 1, 4100 + seed)
    for b in range(B):
        k = int(rng.integers(0, 3))
        idx = rng.choice(np.flatnonzero(ub[b] > 0), k, replace=False)
        ub[b, idx] = 0.0
        stuck[b, idx] = rng.uniform(0, 1, k) * F_MAX
    W = np.ascontiguousarray(rng.uniform(0, 0.3, (B, N, NT)) * ub[:, None, :])
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), warmU=W.copy(), return_U=True)
    assert "ftmpc_solve_ws32_kernel" in mpc.kernel_name(5)
    ref = co.solve_batch_complete(cfg, x0, ub, stuck, xref, warmU=W, nthreads=8)
    assert (out["status"] == 0).all(), np.bincount(out["status"])
    err = np.abs(out["u0"] - ref["u0"]).max(axis=1) / F_MAX
    assert err.max() <= 1e-4 and np.percentile(err, 90) <= 1e-5, (err.max(), np.percentile(err, 90), np.bincount(ref["how"]))
    assert (out["u0"][ub == 0] == 0).all()


def test_wrench_space_kernel_agrees_with_the_dense_workgroup_kernel(gpu_mpc_factory):
    """Kernel 8 solves the Newton systems of the SAME interior-point iteration through the 6N-variable wrench-space form
    x = Dg^-1 (r - DD' L (I + L' S L)^-1 L' DD Dg^-1 r): same iterates as the dense kernel 7 up to fp32 rounding, also
    where the healthy thrusters do not span R^6 (faults 12 + 13: S is singular, the form does not care)."""
    N, NT, B = 15, 16, 1024
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 3800)
    for b in range(0, B, 8):            # every eighth instance: the degenerate pair
        ub[b], stuck[b] = rm.F_MAX, 0.0
        ub[b, [12, 13]] = 0.0
        stuck[b, [12, 13]] = [0.3 * rm.F_MAX, 0.9 * rm.F_MAX]
    for b in range(4, B, 8):            # and a heavily damaged vehicle (8 faults: n = 120 goes to the one-wave kernel)
        idx = np.random.default_rng(b).choice(NT, 8, replace=False)
        ub[b, idx] = 0.0
    xr = xref.reshape(-1, order="F")
    b = gpu_mpc_factory(N=N, NT=NT, kernel_select="dense").solve(x0, ub, stuck, xr, return_U=True)
    for sel in ("auto", "workgroup"):      # kernel 10 (one wave per instance) and kernel 8 (workgroup per instance)
        a = gpu_mpc_factory(N=N, NT=NT, kernel_select=sel).solve(x0, ub, stuck, xr, return_U=True)
        assert (a["status"] == 0).all() and (b["status"] == 0).all()
        assert np.abs(a["U"] - b["U"]).max() / F_MAX <= 2e-5, sel
        assert np.abs(a["u0"] - b["u0"]).max() / F_MAX <= 5e-6, sel
        assert (np.abs(a["iters"].astype(int) - b["iters"]) <= 1).all() and (a["iters"] == b["iters"]).mean() > 0.97, sel


@pytest.mark.parametrize("N,NT,dtype", [(15, 16, "f32"), (15, 16, "f64"), (20, 8, "f32")])
def test_results_are_bitwise_repeatable(gpu_mpc_factory, N, NT, dtype):
    """The same batch twice on one handle and once on a fresh one: identical bits.  (Kernel 7's start gradient once summed
    its tiles with LDS float atomics from four waves: right to 1e-6, different on every run.)"""
    B = 1024
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 3600 + N)
    xr = xref.reshape(-1, order="F")
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype=dtype)
    a = mpc.solve(x0, ub, stuck, xr, return_U=True)
    b = mpc.solve(x0, ub, stuck, xr, return_U=True)
    c = gpu_mpc_factory(N=N, NT=NT, dtype=dtype).solve(x0, ub, stuck, xr, return_U=True)
    for k in ("u0", "U", "status", "iters"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), k


@pytest.mark.parametrize("N,NT,dtype,B", [(20, 8, "f32", 700), (20, 8, "f32", 9000), (20, 8, "f32", 20000), (15, 16, "f32", 700),
                                          (15, 16, "f64", 700)])
def test_direction_split_linearisation_gives_the_same_bits(gpu_mpc_factory, N, NT, dtype, B):
    """Batches of <= 8192 instances are linearised by 13 blocks per 64 instances (one tangent direction each), up to 16 384
    by 4 and up to 40 960 by 2 shares of the directions; the records, and with them every output bit, equal those of the
    full-record kernel (forced here through ftmpc_config.lin_split_max < 0).  So a batch split into shards gives the bits of
    the unsplit call whichever kernel each piece takes."""
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 3700 + N)
    xr = xref.reshape(-1, order="F")
    traj = rm.circle_trajectory(0.1, 10, radius=0.65, s_per_circle=40.0)
    xr_all, ur_all = rm.assign_trajectory(traj, N)
    xw, uw = rm.trajectory_window(xr_all, ur_all, 1.0, N)
    W = np.ascontiguousarray(np.random.default_rng(5).uniform(0, 1, (B, N, NT)) * ub[:, None, :])
    split = gpu_mpc_factory(N=N, NT=NT, dtype=dtype)
    full = gpu_mpc_factory(N=N, NT=NT, dtype=dtype, lin_split_max=-1)
    for warm in (False, True):
        xx = xw.reshape(-1, order="F") if warm else xr
        ka = dict(uref=uw.reshape(-1, order="F"), warmU=W.copy()) if warm else {}
        kb = dict(uref=uw.reshape(-1, order="F"), warmU=W.copy()) if warm else {}
        a = split.solve(x0, ub, stuck, xx, return_U=True, **ka)
        b = full.solve(x0, ub, stuck, xx, return_U=True, **kb)
        for k in ("u0", "U", "status", "iters"):
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("sel", ["auto", "workgroup", "dense"])
def test_wg_kernel_whole_batch_mixed_fault_counts_warm_start_and_uref(gpu_mpc_factory, sel):
    """4096 instances (16 per workgroup), fault counts 0..10 mixed in one batch (routed between the one-wave kernels
    NB = 8 / 9 / 10 and the workgroup kernel 8 or 7), then a warm-started step with a circle reference window."""
    import os
    N, NT, B = 15, 16, 4096
    mpc = gpu_mpc_factory(N=N, NT=NT, kernel_select=sel)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 3500)
    rng = np.random.default_rng(9)
    for b in range(0, B, 5):                      # every fifth instance: a random number of extra broken thrusters
        k = int(rng.integers(0, 9))
        idx = rng.choice(np.flatnonzero(ub[b] > 0), k, replace=False)
        ub[b, idx] = 0.0
        stuck[b, idx] = rng.uniform(0, 1, k) * F_MAX
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    nt = min(32, len(os.sched_getaffinity(0)))
    ref = co.solve_batch_complete(_cfg(N, NT), x0, ub, stuck, xref, nthreads=nt)    # every instance has a reference
    assert (out["status"] == 0).all(), np.bincount(out["status"])
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX <= 1e-4, np.bincount(ref["how"])
    traj = rm.circle_trajectory(0.1, 10, radius=0.65, s_per_circle=40.0)
    xr_all, ur_all = rm.assign_trajectory(traj, N)
    xw, uw = rm.trajectory_window(xr_all, ur_all, 1.0, N)
    W = np.ascontiguousarray(np.concatenate([out["U"][:, 1:], np.zeros((B, 1, NT))], axis=1))
    W0 = W.copy()
    out2 = mpc.solve(x0, ub, stuck, xw.reshape(-1, order="F"), uref=uw.reshape(-1, order="F"), warmU=W)
    ref2 = co.solve_batch_complete(_cfg(N, NT), x0, ub, stuck, xw, uref=uw, warmU=W0, nthreads=nt)
    assert (out2["status"] == 0).all(), np.bincount(out2["status"])
    assert np.abs(out2["u0"] - ref2["u0"]).max() / F_MAX <= 1e-4, np.bincount(ref2["how"])
    assert np.abs(W - ref2["U"]).max() / F_MAX <= TOL_U


# ---------------------------------------------------------------------------------------------
# float64 general-size kernel (reference 16-thruster vehicle, long horizons)
# ---------------------------------------------------------------------------------------------
def _faults16(B, pattern):
    ub = np.full((B, 16), F_MAX)
    stuck = np.zeros((B, 16))
    for i, a in pattern:
        ub[:, i] = 0.0
        stuck[:, i] = a * F_MAX
    return ub, stuck


@pytest.mark.parametrize("sel", ["auto", "workgroup", "dense"])
@pytest.mark.parametrize("N,pattern", [(15, [(10, 1.0), (11, 1.0)]), (15, []), (20, [(0, 0.5)]), (15, [(12, 0.3), (13, 0.9)])])
def test_f64_kernel_reference_vehicle(gpu_mpc_factory, N, pattern, sel):
    """NT=16 (the reference's D, sys_model.py:73-123); N=15 with thrusters 10,11 stuck fully on is the
    shipped reactive.yaml scenario; 12 + 13 is the pair whose healthy thrusters do not span R^6 (S singular: the
    wrench-space form does not care).  float64 path: u0 and U within 1e-7 f_max of the exact solution, through the Riccati
    recursion (auto: ftmpc_solve_ric64_kernel), the wrench-space form (workgroup: ftmpc_solve_ws64_kernel) and the dense
    factorisation (dense: ftmpc_solve_f64_kernel)."""
    B = 6
    mpc = gpu_mpc_factory(N=N, NT=16, dtype="f64", max_iters=40, kernel_select=sel)
    x0, _, _, xref = qo.make_batch(B, N, 16, 0, 3000 + N)
    ub, stuck = _faults16(B, pattern)
    cfg = _cfg(N, 16)
    H, g, lo, hi = mpc.debug_build_qp(x0, ub, stuck, xref.reshape(-1, order="F"), 2)
    qp = qo.build_qp(cfg, x0[2], ub[2], stuck[2], xref)
    assert np.abs(H - qp["H"]).max() <= 1e-11 * np.abs(qp["H"]).max()
    assert np.abs(g - qp["g"]).max() <= 1e-11 * max(1.0, np.abs(qp["g"]).max())
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    ref = co.solve_batch(cfg, x0, ub, stuck, xref, nthreads=4, max_iters=60)
    assert (out["status"] == 0).all(), out["status"]
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX < 1e-7
    assert np.abs(out["U"] - ref["U"]).max() / F_MAX < 1e-6


@pytest.mark.parametrize("sel", ["auto", "dense"])
def test_f64_kernel_config5_shape(gpu_mpc_factory, sel):
    """BASELINE config 5 shape: N=40, 16 thrusters, two random faults, fp64 KKT (n = 560; K of the wrench-space form: 240)."""
    N, NT, B = 40, 16, 6
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40, kernel_select=sel)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 1005)
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    ref = co.solve_batch(_cfg(N, NT), x0, ub, stuck, xref, nthreads=6, max_iters=60)
    assert (out["status"] == 0).all(), out["status"]
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX < 1e-7
    assert np.abs(out["U"] - ref["U"]).max() / F_MAX < 1e-6


@pytest.mark.parametrize("sel", ["auto", "workgroup"])
@pytest.mark.parametrize("N,NT,seed", [(15, 16, 1), (24, 12, 2), (40, 16, 3), (33, 14, 4), (42, 16, 5), (7, 10, 6)])
def test_f64_wrench_space_kernel_other_vehicles_horizons_mixed_faults_warm_start(gpu_mpc_factory, N, NT, seed, sel):
    """ftmpc_solve_ric64_kernel (auto, N <= 40; one wave per instance, Newton systems by the Riccati recursion) and
    ftmpc_solve_ws64_kernel (workgroup, and auto beyond N = 40) away from the config-5 shape: 10..16 thrusters (random allocation matrices of full row rank),
    horizons 7..42 (6 N = 42 .. 252: up to sixteen tiles a side, one and three thruster variables per thread), 0..9 broken
    thrusters mixed in one batch (down to fewer healthy thrusters than wrench components: S rank deficient), warm start and
    a moving reference with a non-zero uref; against the C oracle at the float64 tolerance."""
    rng = np.random.default_rng(7700 + seed)
    B = 40
    D = None
    if NT != 16:
        D = rng.standard_normal((6, NT)) * np.array([1, 1, 1, 0.3, 0.3, 0.3])[:, None]
    cfg = _cfg(N, NT) if D is None else qo.QPConfig(N=N, NT=NT, D=D)
    kw = {} if D is None else dict(D=D)
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40, kernel_select=sel, **kw)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 1, 7800 + seed)
    for b in range(B):
        k = int(rng.integers(0, 9))
        idx = rng.choice(np.flatnonzero(ub[b] > 0), k, replace=False)
        ub[b, idx] = 0.0
        stuck[b, idx] = rng.uniform(0, 1, k) * F_MAX
    traj = rm.circle_trajectory(0.1, 10, radius=0.65, s_per_circle=40.0)
    xr_all, ur_all = rm.assign_trajectory(traj, N)
    xw, uw = rm.trajectory_window(xr_all, ur_all, 1.0, N)
    W = np.ascontiguousarray(rng.uniform(0, 0.3, (B, N, NT)) * ub[:, None, :])
    W0 = W.copy()
    out = mpc.solve(x0, ub, stuck, xw.reshape(-1, order="F"), uref=uw.reshape(-1, order="F"), warmU=W, return_U=True)
    mpc.set_profiling(True)
    mpc.solve(x0[:4], ub[:4], stuck[:4], xw.reshape(-1, order="F"))
    assert ("ftmpc_solve_ric64_kernel" if (sel == "auto" and N <= 40) else "ftmpc_solve_ws64_kernel") in mpc.last_kernel_ms()
    # (kernel 12 finishes by the active-set polish and is held against the polished port; kernel 9 runs the iteration to mu 1e-13 alone)
    ref = co.solve_batch_complete(cfg, x0, ub, stuck, xw, uref=uw, warmU=W0, nthreads=8, polish=(sel == "auto" and N <= 40))
    co.exact_where_apart(cfg, ref, out["U"], x0, ub, stuck, xw, uref=uw, warmU=W0, tol=1e-6, tol_u0=1e-7, cap=8)      # (safety net: see there)
    assert (out["status"] == 0).all(), np.bincount(out["status"])
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX < 1e-7, np.bincount(ref["how"])
    assert np.abs(out["U"] - ref["U"]).max() / F_MAX < 1e-6 and np.array_equal(W, out["U"])
    assert (out["u0"][ub == 0] == 0).all()


def test_f64_kernel_beyond_640_variables(gpu_mpc_factory):
    """The float64 kernel has three instantiations (n <= 256, <= 640, <= 1024); N = 52 with all 16 thrusters healthy is
    n = 832: four columns per thread and block rows beyond the prefetched ones in the triangular sweeps."""
    N, NT, B = 52, 16, 2
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 0, 1006)
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"), return_U=True)
    ref = co.solve_batch(_cfg(N, NT), x0, ub, stuck, xref, nthreads=2, max_iters=60)
    assert (out["status"] == 0).all(), out["status"]
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX < 1e-7
    assert np.abs(out["U"] - ref["U"]).max() / F_MAX < 1e-6


def test_f64_dtype_on_small_problem_and_warm_start(gpu_mpc_factory):
    """dtype='f64' routes n<=160 problems through the float64 kernel too; warm start + uref."""
    d = np.load(__import__("pathlib").Path(__file__).parent / "golden" / "qp_cfg3_warm_uref.npz")
    N, NT = int(d["N"]), int(d["NT"])
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64", max_iters=40)
    W = np.ascontiguousarray(d["warm"])
    out = mpc.solve(d["x0"], d["ub"], d["stuck"], d["xref"].reshape(-1, order="F"),
                    uref=d["uref"].reshape(-1, order="F"), warmU=W, return_U=True)
    assert (out["status"] == 0).all()
    assert np.abs(out["U"] - d["U"]).max() / F_MAX < 1e-6
    assert np.abs(W - d["U"]).max() / F_MAX < 1e-6      # warm buffer updated in place with U*


@pytest.mark.parametrize("name", ["cfg2_single_fault", "cfg3_double_fault", "cfg3_warm_uref", "nominal_nt8", "short_horizon"])
def test_f32_kernels_against_golden(gpu_mpc_factory, name):
    """fp32 LDS kernels on the committed golden QP fixtures (exact BVLS solutions), including the
    warm-start + uref window and a short horizon; u0 <= 1e-4 f_max."""
    d = np.load(__import__("pathlib").Path(__file__).parent / "golden" / f"qp_{name}.npz")
    N, NT = int(d["N"]), int(d["NT"])
    mpc = gpu_mpc_factory(N=N, NT=NT)
    ur = d["uref"].reshape(-1, order="F") if d["uref"].size else None
    W = np.ascontiguousarray(d["warm"]).copy() if d["warm"].size else None
    out = mpc.solve(d["x0"], d["ub"], d["stuck"], d["xref"].reshape(-1, order="F"), uref=ur, warmU=W, return_U=True)
    assert (out["status"] == 0).all()
    assert np.abs(out["u0"] - d["u0"]).max() / F_MAX <= 1e-4
    assert np.abs(out["U"] - d["U"]).max() / F_MAX <= TOL_U
    assert (out["U"][np.repeat(d["ub"][:, None, :], N, 1) == 0] == 0).all()


def test_mixed_fault_counts_and_empty_instances(gpu_mpc_factory):
    """One batch mixing 0/1/2/8 broken thrusters: routed between the NB=8 and NB=10 instantiations."""
    N, NT, B = 20, 8, 32
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 77)
    ub[0:8] = F_MAX; stuck[0:8] = 0.0                      # nominal
    ub[8:16, 1:] = F_MAX; stuck[8:16, 1:] = 0.0            # exactly thruster 0 broken
    ub[8:16, 0] = 0.0; stuck[8:16, 0] = 1.7
    ub[31, :] = 0.0; stuck[31, :] = 0.5                    # nothing left to command
    out = mpc.solve(x0, ub, stuck, xref.reshape(-1, order="F"))
    ref = co.solve_batch(_cfg(N, NT), x0, ub, stuck, xref, nthreads=4)
    assert (out["status"] == 0).all()
    assert np.abs(out["u0"] - ref["u0"]).max() / F_MAX <= 1e-4
    assert (out["u0"][31] == 0).all() and out["iters"][31] == 0


def test_device_pointer_entry_matches_host_entry(gpu_mpc_factory):
    """ftmpc_solve_batch_device (HBM-resident buffers on a caller stream; what bench.py times) gives
    the same answers as the host-buffer entry, including in-place warm-start update (out_U == warmU)."""
    import torch
    N, NT, B = 20, 8, 96
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, 2, 555)
    xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
    host = mpc.solve(x0, ub, stuck, xr, return_U=True)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_x0, d_ub, d_st, d_xr = t(x0), t(ub), t(stuck), t(xr)
    d_u0 = torch.zeros(B, NT, dtype=torch.float64, device=dev)
    d_U = torch.zeros(B, N, NT, dtype=torch.float64, device=dev)
    d_status = torch.full((B,), -1, dtype=torch.int32, device=dev)
    d_iters = torch.zeros(B, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        mpc.solve_device(B, d_x0.data_ptr(), d_ub.data_ptr(), d_st.data_ptr(), d_xr.data_ptr(), 0, 0, 0, 0,
                         d_u0.data_ptr(), d_U.data_ptr(), d_status.data_ptr(), d_iters.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert (d_status.cpu().numpy() == 0).all()
    assert np.array_equal(d_u0.cpu().numpy(), host["u0"]) and np.array_equal(d_U.cpu().numpy(), host["U"])
    # second step, warm-started in place from the first solution
    with torch.cuda.stream(s):
        mpc.solve_device(B, d_x0.data_ptr(), d_ub.data_ptr(), d_st.data_ptr(), d_xr.data_ptr(), 0, 0, 0, d_U.data_ptr(),
                         d_u0.data_ptr(), d_U.data_ptr(), d_status.data_ptr(), d_iters.data_ptr(), s.cuda_stream)
    s.synchronize()
    W = host["U"].copy()
    host2 = mpc.solve(x0, ub, stuck, xr, warmU=W, return_U=True)
    assert np.array_equal(d_U.cpu().numpy(), host2["U"]) and np.array_equal(W, host2["U"])


@pytest.mark.parametrize("name,dtype,sel,tol", [("refvehicle_n15", "f32", "auto", 1e-4), ("refvehicle_n15", "f32", "dense", 1e-4),
                                                ("refvehicle_n15", "f32", "workgroup", 1e-4), ("refvehicle_n20", "f32", "workgroup", 1e-4),
                                                ("refvehicle_n20", "f32", "auto", 1e-4), ("refvehicle_n15", "f64", "auto", 1e-7),
                                                ("refvehicle_n20", "f64", "auto", 1e-7), ("refvehicle_n20", "f64", "dense", 1e-7),
                                                ("cfg5_n40_nt16", "f64", "auto", 1e-7), ("cfg5_n40_nt16", "f64", "dense", 1e-7)])
def test_sixteen_thruster_kernels_against_golden(gpu_mpc_factory, name, dtype, sel, tol):
    """Committed exact (BVLS) solutions for the reference vehicle at N = 15 / 20 and BASELINE config 5 (tests/golden/, made by
    oracle/gen_golden.py:qp_fixtures_large): kernels 8 / 7 (fp32), 9 and the dense float64 kernel."""
    d = np.load(__import__("pathlib").Path(__file__).parent / "golden" / f"qp_{name}.npz")
    N, NT = int(d["N"]), int(d["NT"])
    assert d["x0"].shape[0] >= 16
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype=dtype, kernel_select=sel, max_iters=40)
    out = mpc.solve(d["x0"], d["ub"], d["stuck"], d["xref"].reshape(-1, order="F"), return_U=True)
    assert (out["status"] == 0).all(), out["status"]
    assert np.abs(out["u0"] - d["u0"]).max() / F_MAX <= tol
    assert np.abs(out["U"] - d["U"]).max() / F_MAX <= (TOL_U if dtype == "f32" else 1e-6)
