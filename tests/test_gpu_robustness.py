"""GPU: failure paths and build variants of the C-ABI library.
  * a workspace growth that fails (out of device memory) must leave the handle usable (stale capacity -> launch on
    freed pointers);
  * ftmpc_solve_wrench_batch refuses a hull table number outside [0, n_sets) instead of reading past the tables on the device;
  * the shipped code object is hipcc's assembly with the asm-side wait states LOWERED by scripts/check_hazards.py
    (csrc/Makefile); `make plain` keeps them as written.  Both builds must give the same bits on every kernel family: a
    wrong entry of the wait-state table would show as slightly different numbers, nowhere else.
"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import ft_mpc_amd
from ft_mpc_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_failed_reserve_leaves_the_handle_usable(gpu_mpc_factory):
    N, NT = 20, 8
    mpc = gpu_mpc_factory(N=N, NT=NT)
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(512, N, NT, 2, 7100)
    xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
    a = mpc.solve(x0, ub, stuck, xr, return_U=True)
    with pytest.raises(ft_mpc_amd.FtmpcError) as e:
        mpc.reserve(1 << 31)                  # 2^31 instances x 24 KB of stage records = 52 TB: hipMalloc must refuse
    assert e.value.code == -4
    b = mpc.solve(x0[:100], ub[:100], stuck[:100], xr, return_U=True)     # smaller than the capacity held before the failure
    for k in ("u0", "U", "status", "iters"):
        assert np.array_equal(a[k][:100], b[k]), k
    c = mpc.solve(x0, ub, stuck, xr, return_U=True)
    assert np.array_equal(a["u0"], c["u0"])


def test_wrench_entry_refuses_a_hull_table_number_out_of_range(gpu_mpc_factory):
    from ft_mpc_amd.controllers.tools.input_bounds import hull_tables
    N, NT, B = 15, 16, 8
    mpc = gpu_mpc_factory(N=N, NT=NT, dtype="f64")
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, 1, 7200)
    xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
    hull = hull_tables(mpc.D, ub, stuck)
    good = mpc.solve_wrench(x0, ub, stuck, xr, hull=hull)
    assert (good["status"] == 0).all()
    for bad in (-1, int(np.asarray(hull["A"]).shape[0])):
        h2 = dict(hull)
        h2["set"] = np.array(hull["set"], np.int32).copy()
        h2["set"][3] = bad
        with pytest.raises(ft_mpc_amd.FtmpcError) as e:
            mpc.solve_wrench(x0, ub, stuck, xr, hull=h2)
        assert e.value.code == -1 and "hull_set[3]" in str(e.value)
    again = mpc.solve_wrench(x0, ub, stuck, xr, hull=hull)
    assert np.array_equal(good["u0"], again["u0"])


_AB_CASES = [  # (name, N, NT, faults, B, dtype, kernel_select): one batch per kernel family of the library
    ("headline_f32_nb8", 20, 8, 2, 2048, "f32", "auto"),
    ("f32_nb9_nb10", 20, 8, 0, 1024, "f32", "auto"),
    ("refvehicle_wsw32_6", 15, 16, 2, 1024, "f32", "auto"),
    ("refvehicle_ws32_6", 15, 16, 2, 1024, "f32", "workgroup"),
    ("ws32_8_two_per_thread", 20, 16, 2, 256, "f32", "workgroup"),
    ("refvehicle_wg32", 15, 16, 2, 512, "f32", "dense"),
    ("wsw32_8", 20, 16, 2, 512, "f32", "auto"),
    ("config5_ric64", 40, 16, 2, 128, "f64", "auto"),
    ("refvehicle_ric64", 15, 16, 2, 256, "f64", "auto"),
    ("config5_ws64", 40, 16, 2, 128, "f64", "workgroup"),
    ("config5_f64_dense", 40, 16, 2, 64, "f64", "dense"),
    ("refvehicle_f64", 15, 16, 2, 256, "f64", "dense"),
    ("refvehicle_hull32", 15, 16, 2, 512, "f32", "wrench"),      # kernel 11 through ftmpc_solve_wrench_batch
    ("refvehicle_hull32_tset", 15, 16, 2, 256, "f32", "wrench_tset"),      # ... with the terminal set (its own instantiation)
]

_AB_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/fault-tolerant-mpc_amd")
import ft_mpc_amd
cases = eval(sys.argv[3])
out = {}
for name, N, NT, nf, B, dtype, sel in cases:
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, nf, 7300 + N + NT + nf)
    if sel.startswith("wrench"):
        kw = {}
        if sel == "wrench_tset":
            from ft_mpc_amd.controllers.tools.terminal_ingredients import load_terminal
            kw = dict(terminal_set=load_terminal().term_set, max_iters=60)
            x0[:, 0:6] *= 0.02                           # near the reference: the set is reachable for most
        mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype=dtype, **kw)
        r = mpc.solve_wrench(x0, ub, stuck, np.ascontiguousarray(xref.reshape(-1, order="F")), return_G=True)
        mpc.close()
        keep = (r["status"] != 3) if sel == "wrench" else (r["status"] == 0)      # (3: no hull; with the set: the reachable ones)
        for k in ("u0", "G", "tau0", "iters"):
            out[name + "/" + k] = r[k][keep]
        out[name + "/status"] = r["status"][keep]
        out[name + "/all_status"] = r["status"] * 0 + (r["status"] == 0)      # (bitwise the same verdicts on both builds)
        continue
    mpc = ft_mpc_amd.BatchedMPC(N=N, NT=NT, dtype=dtype, kernel_select=sel)
    r = mpc.solve(x0, ub, stuck, np.ascontiguousarray(xref.reshape(-1, order="F")), return_U=True)
    mpc.close()
    for k in ("u0", "U", "status", "iters"):
        out[name + "/" + k] = r[k]
out["build_id"] = np.frombuffer(ft_mpc_amd._lib.load_library().ftmpc_build_id(), dtype=np.uint8)
np.savez(sys.argv[2], **out)
print(ft_mpc_amd._lib.library_path())
"""


def test_lowered_wait_states_give_the_bits_of_the_plain_build(tmp_path):
    plain = _lib._HERE / "libftmpc_hip_plain.so"
    if not plain.exists():
        pytest.fail(f"{plain} is missing: build it with `make -C fault-tolerant-mpc_amd/csrc plain` (__graft_entry__.build() does)")
    res = {}
    for tag, so in (("shipped", None), ("plain", plain)):
        env = dict(os.environ)
        env.pop("FTMPC_LIB", None)
        if so is not None:
            env["FTMPC_LIB"] = str(so)
        npz = tmp_path / f"{tag}.npz"
        p = subprocess.run([sys.executable, "-c", _AB_SCRIPT, str(ROOT), str(npz), repr(_AB_CASES)], env=env, capture_output=True, text=True,
                           timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        assert p.stdout.strip().endswith("libftmpc_hip_plain.so" if so is not None else "libftmpc_hip.so"), p.stdout
        res[tag] = dict(np.load(npz))
    assert res["shipped"].keys() == res["plain"].keys()
    ids = [bytes(res[t]["build_id"]).decode() for t in ("shipped", "plain")]
    assert ids[0] == ids[1] and ids[0] != "unknown", f"the plain build is stale (sources {ids[1]}, shipped {ids[0]}): make -C fault-tolerant-mpc_amd/csrc plain"
    for k in res["shipped"]:
        assert np.array_equal(res["shipped"][k], res["plain"][k]), k
        if k.endswith("/status"):
            assert (res["shipped"][k] == 0).all(), (k, np.bincount(res["shipped"][k]))
