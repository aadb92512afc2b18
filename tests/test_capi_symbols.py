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


def test_config_struct_layout_matches_the_header_and_guards_the_abi(tmp_path):
    """The ctypes mirror has the size and the field offsets a C compiler gives include/ftmpc.h; ftmpc_default_config
    stamps that size into struct_size and ftmpc_create refuses any other value BEFORE it looks at a device (so a caller
    built against an older, shorter struct is told instead of being read past its end)."""
    import subprocess
    from ft_mpc_amd import _lib
    src = tmp_path / "layout.c"
    fields = ["N", "struct_size", "dt", "D", "P", "mu_stop", "terminal_set", "term_A", "tc_npoly", "tc_root_exp", "tc_const",
              "kernel_select", "stage_chunks", "lin_split_max"]
    body = "".join(f'printf("{f} %zu\\n", offsetof(ftmpc_config, {f}));' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ftmpc.h"\nint main(void){printf("sizeof %zu\\n", sizeof(ftmpc_config));'
                   + body + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(got["sizeof"]) == C.sizeof(_lib.ftmpc_config)
    for f in fields:
        assert int(got[f]) == getattr(_lib.ftmpc_config, f).offset, f
    lib = _lib.load_library()
    c = _lib.ftmpc_config()
    assert lib.ftmpc_default_config(C.byref(c), 20, 8) == 0
    assert c.struct_size == C.sizeof(_lib.ftmpc_config) and c.kernel_select == 0 and c.lin_split_max == 0 and c.stage_chunks == 0
    h = C.c_void_p()
    c.struct_size = 0                      # what a caller built against the round-2 header (reserved0 = 0) would pass
    assert lib.ftmpc_create(C.byref(c), C.byref(h)) == -1 and not h.value
    assert b"struct_size" in lib.ftmpc_last_error(None)
