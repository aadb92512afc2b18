"""CPU: the algebra kernel 8 (csrc/ftmpc_solve_ws.hip) rests on -- the condensed thruster-space Hessian is the wrench-space
Hessian seen through blockdiag(D_a) plus 2 rho I, and the Newton systems of the interior-point iteration can be solved
through a 6N x 6N matrix K = I + L' S L that stays well conditioned whatever the barrier weights do."""
import numpy as np
import pytest

from oracle import qp_oracle as qo


@pytest.mark.parametrize("N,NT,nf,seed", [(15, 16, 2, 1), (15, 16, 0, 2), (20, 8, 0, 3), (12, 16, 5, 4)])
def test_thruster_hessian_is_the_wrench_hessian_through_the_allocation_matrix(N, NT, nf, seed):
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(3, N, NT, nf, 5000 + seed)
    rng = np.random.default_rng(seed)
    for b in range(3):
        W = rng.uniform(0, 1, (N, NT)) * ub[b] if b else None
        qp = qo.build_qp(cfg, x0[b], ub[b], stuck[b], xref, warmU=W)
        Hw, Da = qo.wrench_form(cfg, x0[b], ub[b], stuck[b], xref, warmU=W)
        DD = np.kron(np.eye(N), Da)
        assert np.abs(qp["H"] - (DD.T @ Hw @ DD + 2 * cfg.rho * np.eye(N * qp["na"]))).max() <= 1e-12 * np.abs(qp["H"]).max()


def test_newton_systems_through_wrench_space_also_with_singular_S():
    N, NT = 15, 16
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(2, N, NT, 2, 5100)
    ub[1] = cfg.f_max if hasattr(cfg, "f_max") else ub[1].max()
    stuck[1] = 0.0
    ub[1, [12, 13]] = 0.0                                   # the pair that leaves the healthy thrusters rank 5 in R^6
    rng = np.random.default_rng(0)
    for b in range(2):
        qp = qo.build_qp(cfg, x0[b], ub[b], stuck[b], xref)
        Hw, Da = qo.wrench_form(cfg, x0[b], ub[b], stuck[b], xref)
        n = qp["n"]
        for Sig in (np.exp(rng.uniform(-8, 12, n)), np.full(n, 1e-9), np.where(rng.random(n) < 0.8, 1e10, 1e-6)):
            r = rng.standard_normal(n)
            x = qo.schur_newton_solver(Hw, Da, N, cfg.rho)(Sig)(r)
            xd = np.linalg.solve(qp["H"] + np.diag(Sig), r)
            assert np.abs(x - xd).max() <= 1e-9 * max(1.0, np.abs(xd).max()), (b, np.abs(x - xd).max())
            x32 = qo.schur_newton_solver(Hw, Da, N, cfg.rho, dtype=np.float32)(Sig)(r)
            assert np.isfinite(x32).all() and np.abs(x32 - xd).max() <= 2e-3 * np.abs(xd).max()
    assert np.linalg.matrix_rank(cfg.D[:, ub[1] > 0]) == 5
