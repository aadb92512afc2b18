"""GPU check of the cross-lane building blocks of the fp32 solve kernel (scripts/test_prims.hip):
DPP / permlane broadcasts and sums, and the in-register 16x16 potrf + inverse against a float64
Cholesky computed on the host.  Compiles the diagnostic with hipcc on the GPU box (~30 s)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_inregister_potrf_and_lane_primitives(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "test_prims"
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-o", str(exe), str(ROOT / "scripts" / "test_prims.hip")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120).stdout
    # group broadcast of row-group QS: lane l gets the value of lane 16*QS + (l & 15)
    for qs in range(4):
        m = re.search(rf"gb{qs}:(.*)", out)
        vals = dict((int(a), float(b)) for a, b in re.findall(r"\[(\d+)\]=([-\d.e+]+)", m.group(1)))
        assert all(v == 16 * qs + (l & 15) for l, v in vals.items()), (qs, vals)
    vals = dict((int(a), float(b)) for a, b in re.findall(r"\[(\d+)\]=([-\d.e+]+)", re.search(r"rb5:(.*)", out).group(1)))
    assert all(v == 16 * (l >> 4) + 5 for l, v in vals.items())
    vals = dict((int(a), float(b)) for a, b in re.findall(r"\[(\d+)\]=([-\d.e+]+)", re.search(r"qsum:(.*)", out).group(1)))
    assert all(v == 4 * (l & 15) + 96 for l, v in vals.items())        # sum over the four row-groups of lane index
    m = re.search(r"potrf ok=(\d+)\s+max\|W\*L-I\|=([-\d.e+]+)", out)
    assert m and int(m.group(1)) == 1 and float(m.group(2)) < 1e-5, out[-500:]
    assert float(re.search(r"max upper\(W\)=([-\d.e+]+)", out).group(1)) == 0.0
