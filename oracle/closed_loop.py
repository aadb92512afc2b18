"""oracle/closed_loop.py -- float64 restatement of the batched closed loop (TEST INFRASTRUCTURE ONLY):
MPC step (C oracle) -> plant RK4 (ft_mpc/models/sys_model.py:138-226) -> U(0, a) noise
(ft_mpc/simulation/sim_env.py:88-91, from the same counter-based generator as csrc/ftmpc_sim.hip) ->
quaternion renormalisation (sim_env.py:93), warm start shifted by one stage (spiraling_mpc.py:324-334)."""
import numpy as np

from . import c_oracle as co

_M = (1 << 64) - 1


def u01(seed, idx):
    """splitmix64-based uniform in [0,1); idx: uint64 array.  Mirrors ftmpc_sim.hip:u01."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def simulate(qcfg, x0, ub, stuck, xref_traj, T, uref_traj=None, noise=(1e-3,) * 4, seed=0, max_iters=60, mu_stop=1e-13,
             nthreads=4):
    N, NT = qcfg.N, qcfg.NT
    x = np.array(x0, float).reshape(-1, 13)
    B = x.shape[0]
    amp = np.repeat(np.asarray(noise, float), [3, 3, 4, 3])
    warm = None
    us = np.zeros((T, B, NT))
    for t in range(T):
        xr = np.ascontiguousarray(xref_traj[:, t:t + N + 1])
        ur = None if uref_traj is None else np.ascontiguousarray(uref_traj[:, t:t + N + 1])
        out = co.solve_batch(qcfg, x, ub, stuck, xr, uref=ur, warmU=warm, max_iters=max_iters, mu_stop=mu_stop, nthreads=nthreads)
        us[t] = out["u0"]
        warm = np.concatenate([out["U"][:, 1:], np.zeros((B, 1, NT))], axis=1)
        for b in range(B):
            x[b] = co.plant_step(qcfg, x[b], out["u0"][b], ub[b], stuck[b])
        idx = (np.uint64(t) * np.uint64(B) + np.arange(B, dtype=np.uint64))[:, None] * np.uint64(13) + np.arange(13, dtype=np.uint64)[None, :]
        x = x + amp[None, :] * u01(seed, idx) * (amp[None, :] > 0)
        x[:, 6:10] /= np.linalg.norm(x[:, 6:10], axis=1, keepdims=True)
    return x, us


def simulate_wrench(qcfg, x0, ub, stuck, xref_traj, T, uref_traj=None, noise=(1e-3,) * 4, seed=0, iters=60, term_set=None):
    """The closed loop in the reference's TWO-STAGE structure (sim_env.py:77-112 around spiraling_mpc.py:288-317): per step the
    generalized-force MPC with the input hull [+ the terminal set] (oracle/qp_oracle.py:solve_wrench_instance, with its active-set
    polish), the min-norm allocation of tau_0 - D stuck (oracle/alloc_oracle.py:allocate; control_allocator.py:65-94), the plant
    step, noise and renormalisation as in `simulate`; the wrench warm start of the next step is this step's solution shifted by one
    stage with its last stage repeated.  Returns (x [B,13], u [T,B,NT], tau0 [T,B,6], status [T,B], alloc_status [T,B])."""
    from . import alloc_oracle as ao
    from . import qp_oracle as qo
    N, NT = qcfg.N, qcfg.NT
    x = np.array(x0, float).reshape(-1, 13)
    B = x.shape[0]
    amp = np.repeat(np.asarray(noise, float), [3, 3, 4, 3])
    hulls = [qo.zonotope_hrep(qcfg.D, ub[b], stuck[b]) for b in range(B)]      # fixed over the run: the fault pattern does not change
    warm = [None] * B
    us = np.zeros((T, B, NT))
    taus = np.zeros((T, B, 6))
    st = np.zeros((T, B), np.int32)
    ast = np.zeros((T, B), np.int32)
    for t in range(T):
        xr = np.ascontiguousarray(xref_traj[:, t:t + N + 1])
        ur = None if uref_traj is None else np.ascontiguousarray(uref_traj[:, t:t + N + 1])
        for b in range(B):
            with np.errstate(all="ignore"):
                tau0, G, st[t, b], _, _ = qo.solve_wrench_instance(qcfg, x[b], ub[b], stuck[b], xr, uref=ur, warmG=warm[b], hull=hulls[b],
                                                                 term_set=term_set, iters=iters)
            taus[t, b] = tau0
            us[t, b], ast[t, b], _ = ao.allocate(qcfg.D, tau0 - qcfg.D @ stuck[b], ub[b])
            warm[b] = np.concatenate([G[1:], G[-1:]], axis=0)
            x[b] = co.plant_step(qcfg, x[b], us[t, b], ub[b], stuck[b])
        idx = (np.uint64(t) * np.uint64(B) + np.arange(B, dtype=np.uint64))[:, None] * np.uint64(13) + np.arange(13, dtype=np.uint64)[None, :]
        x = x + amp[None, :] * u01(seed, idx) * (amp[None, :] > 0)
        x[:, 6:10] /= np.linalg.norm(x[:, 6:10], axis=1, keepdims=True)
    return x, us, taus, st, ast
