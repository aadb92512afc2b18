"""CPU: the allocation oracle (oracle/alloc_oracle.py) against an independent solver (SLSQP) and its KKT
certificate, on the reference's 16-thruster allocation matrix with the shipped fault patterns."""
import numpy as np
import pytest

from oracle import alloc_oracle as ao
from oracle import refmath as rm


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    D = rm.allocation_matrix_16()
    for _ in range(n):
        ub = np.full(16, rm.F_MAX)
        for i in rng.choice(16, rng.integers(0, 3), replace=False):
            ub[i] = 0.0
        u_true = rng.uniform(0, 1, 16) * ub * rng.uniform(0.1, 1.0)       # attainable by construction
        sat = rng.choice(16, rng.integers(0, 9), replace=False)            # ... some thrusters saturated: bounds become active
        u_true[sat] = ub[sat]
        yield D, D @ u_true, ub


def test_dual_newton_matches_slsqp_and_kkt():
    for D, tau, ub in _cases(11, 40):
        u, status, it, lam = ao.allocate(D, tau, ub, return_lambda=True)
        assert status == 0 and it <= 30
        assert ao.kkt_residual(D, tau, ub, u, lam) < 1e-7
        us, ok = ao.allocate_slsqp(D, tau, ub)
        if ok:
            assert np.abs(u - us).max() < 1e-5
        assert u @ u <= us @ us + 1e-7 or not ok


def test_unattainable_wrench_is_reported():
    D = rm.allocation_matrix_16()
    ub = np.full(16, rm.F_MAX)
    tau = np.array([1e3, 0, 0, 0, 0, 0.0])      # far outside the attainable set
    u, status, _ = ao.allocate(D, tau, ub)
    assert status == 2
    assert (u >= 0).all() and (u <= ub).all()


def test_zero_request_and_broken_thrusters():
    D = rm.allocation_matrix_16()
    ub = np.full(16, rm.F_MAX)
    ub[[10, 11]] = 0.0
    u, status, _ = ao.allocate(D, np.zeros(6), ub)
    assert status == 0 and np.abs(u).max() == 0.0
