"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ftmpc.h
declares.  No compute entry point is called without a GPU; creating a solver must fail loudly."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    hdr = (ROOT / "include" / "ftmpc.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ftmpc_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from ft_mpc_amd import _lib
    _lib.build_library()
    lib = _lib.load_library()
    names = _declared()
    assert len(names) >= 13
    assert sorted(_lib.SYMBOLS) == names
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.ftmpc_version() >= 100


def test_default_config_carries_reference_constants():
    from ft_mpc_amd import _lib
    from ft_mpc_amd.models.sys_model import allocation_matrix_16
    lib = _lib.load_library()
    c = _lib.ftmpc_config()
    assert lib.ftmpc_default_config(C.byref(c), 15, 16) == 0
    assert (c.N, c.NT, c.dt, c.mass) == (15, 16, 0.1, 16.8)
    assert np.array_equal(np.array(list(c.D)).reshape(6, 16), allocation_matrix_16())
    assert list(c.Q) == [1, 1, 1, 1, 1, 1, 2, 2, 2] and list(c.R) == [0.1, 0.1, 0.1, 0.01, 0.01, 0.01]
    assert c.r[1] == pytest.approx(0.5787037037037037) and c.f_virt[1] == 3.5
    assert lib.ftmpc_default_config(C.byref(c), 0, 16) != 0      # bad horizon
    assert lib.ftmpc_default_config(C.byref(c), 20, 17) != 0     # too many thrusters


def test_shift_warm_matches_reference_shift():
    from ft_mpc_amd import _lib
    lib = _lib.load_library()
    w = np.arange(2 * 4 * 3, dtype=float).reshape(2, 4, 3).copy()
    ref = np.concatenate([w[:, 1:], np.zeros((2, 1, 3))], axis=1)   # spiraling_mpc.py:327-329
    assert lib.ftmpc_shift_warm(2, 4, 3, w.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert np.array_equal(w, ref)


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ft_mpc_amd
    with pytest.raises(ft_mpc_amd.FtmpcError) as e:
        ft_mpc_amd.BatchedMPC(N=20, NT=8)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)
