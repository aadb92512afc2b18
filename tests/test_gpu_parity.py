"""GPU parity tests proper: the HIP path (through the C-ABI) against the float64 oracle on the
same seeded inputs.  Tolerances:
  * condensed QP data (H, g built in fp32 on the GPU):   rel-inf 2e-5 on H, 2e-5*scale on g
  * thrust command u0 vs the EXACT QP solution (BVLS):   <= 1e-4 * f_max  (north-star tolerance)
"""
import numpy as np
import pytest

from oracle import qp_oracle as qo
from oracle import refmath as rm

pytestmark = pytest.mark.gpu
F_MAX = rm.F_MAX


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
        assert np.abs(out["U"][b] - U).max() / F_MAX < 2e-3
    assert err.max() <= 1e-4, err
    assert (out["u0"][ub == 0] == 0).all()
    assert out["iters"].max() <= 16
