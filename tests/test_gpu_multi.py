"""GPU: the in-process multi-GPU driver (include/ftmpc.h ftmpc_multi_*; SURVEY.md section 8(e): one host thread +
one handle + one stream set per device, contiguous shards, no collective).  This box has one GPU, so the device
slots all name device 0: several handles and host threads at once on one device, results bitwise equal to the
serial single-handle run."""
import numpy as np
import pytest

import ft_mpc_amd
from ft_mpc_amd.sharding import MultiGPUMPC, shard_bounds, solve_multi_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("slots,B", [(2, 4099), (3, 1000)])
def test_host_buffer_entry_equals_the_serial_run(gpu_mpc_factory, slots, B):
    N, NT = 20, 8
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, 2, 6100 + slots)
    xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
    serial = gpu_mpc_factory(N=N, NT=NT).solve(x0, ub, stuck, xr, return_U=True)
    m = MultiGPUMPC(ft_mpc_amd.MPCConfig(N=N, NT=NT), devices=[0] * slots)
    try:
        assert m.n_devices == slots
        edges = [m.shard_bounds(B, g) for g in range(slots)]
        assert edges == [shard_bounds(B, slots, g) for g in range(slots)]      # same split as the process-per-rank path
        W = np.zeros((B, N, NT))
        out = m.solve(x0, ub, stuck, xr, warmU=None, return_U=True)
        for k in ("u0", "U", "status", "iters"):
            assert np.array_equal(out[k], serial[k]), k
        # warm-started second step, in place, against the serial warm-started step
        W[...] = serial["U"]
        W2 = W.copy()
        a = m.solve(x0, ub, stuck, xr, warmU=W)
        b = gpu_mpc_factory(N=N, NT=NT).solve(x0, ub, stuck, xr, warmU=W2)
        assert np.array_equal(a["u0"], b["u0"]) and np.array_equal(W, W2)
    finally:
        m.close()


def test_resident_shards_step_and_download(gpu_mpc_factory):
    N, NT, B = 20, 8, 6000
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, 2, 6200)
    xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
    serial = gpu_mpc_factory(N=N, NT=NT).solve(x0, ub, stuck, xr, return_U=True)
    m = MultiGPUMPC(ft_mpc_amd.MPCConfig(N=N, NT=NT), devices=[0, 0])
    try:
        with pytest.raises(ft_mpc_amd.FtmpcError):
            m.step(1)                                   # nothing uploaded yet
        m.upload(x0, ub, stuck, xr)
        m.step(3, keep_U=True)
        out = m.download(return_U=True)
        for k in ("u0", "U", "status", "iters"):
            assert np.array_equal(out[k], serial[k]), k
        m.set_profiling(True)
        m.step(1)
        ms = m.last_kernel_ms(1)
        assert ms["ftmpc_linearize_kernel"] > 0 and ms["ftmpc_solve_f32_kernel<8>"] > 0
    finally:
        m.close()


def test_one_call_helper_and_float64_path():
    N, NT, B = 15, 16, 96          # the reference vehicle: float64 workgroup kernel, two handles at once
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, 2, 6300)
    xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
    cfg = ft_mpc_amd.MPCConfig(N=N, NT=NT, max_iters=40)
    one = ft_mpc_amd.BatchedMPC(cfg)
    serial = one.solve(x0, ub, stuck, xr)
    one.close()
    out = solve_multi_gpu(cfg, x0, ub, stuck, xr, devices=[0, 0])
    assert (out["status"] == 0).all() and np.array_equal(out["u0"], serial["u0"])


def test_bad_device_is_refused():
    with pytest.raises(ft_mpc_amd.FtmpcError):
        MultiGPUMPC(ft_mpc_amd.MPCConfig(N=20, NT=8), devices=[0, 99])


@pytest.mark.parametrize("N,NT,dtype,B,nfault,seed", [(20, 8, "f32", 262144, 2, 1004), (40, 16, "f64", 16384, 2, 1005)])
def test_eight_device_slots_at_full_baseline_size_equal_the_serial_run(N, NT, dtype, B, nfault, seed):
    """BASELINE configs[3] (262 144 instances, 32 768 per GPU) and configs[4] (N = 40, 16 thrusters, fp64, 16 384 instances,
    2 048 per GPU) through the in-process driver with EIGHT device slots -- all on this box's one GPU: eight worker threads,
    handles and stream sets at once -- bitwise equal to one handle solving the whole batch; the workers report how many
    host cores they are bound to (the cores nearest their GPU, or 0 when the container grants none of them)."""
    x0, ub, stuck, xref = ft_mpc_amd.make_synthetic_batch(B, N, NT, nfault, seed)
    xr = np.ascontiguousarray(xref.reshape(-1, order="F"))
    cfg = ft_mpc_amd.MPCConfig(N=N, NT=NT, dtype=dtype)
    one = ft_mpc_amd.BatchedMPC(cfg)
    serial = one.solve(x0, ub, stuck, xr)
    one.close()
    assert (serial["status"] == 0).all()
    m = MultiGPUMPC(cfg, devices=[0] * 8)
    try:
        assert m.n_devices == 8 and [m.shard_bounds(B, g) for g in range(8)] == [shard_bounds(B, 8, g) for g in range(8)]
        assert all(m.worker_cpus(g) >= 0 for g in range(8))
        out = m.solve(x0, ub, stuck, xr)                                   # host buffers in / out
        for k in ("u0", "status", "iters"):
            assert np.array_equal(out[k], serial[k]), k
        m.upload(x0, ub, stuck, xr)                                        # shards resident in HBM (pinned upload staging)
        m.step(1)
        res = m.download()
        for k in ("u0", "status", "iters"):
            assert np.array_equal(res[k], serial[k]), k
    finally:
        m.close()
