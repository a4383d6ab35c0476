"""GPU: the C ABI used directly, the way INTEGRATION.md section 2 shows a maintainer of the reference would bind it
(ctypes structs declared here from include/snac_hip.h, caller-owned device arrays, caller's stream) -- no
snac_amd.batched in between.  Checked against the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


class EnvDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dynamic", C.c_int32), ("num_envs", C.c_int32), ("num_plans", C.c_int32),
                ("obs_dtype", C.c_int32), ("static_plan", C.c_int32), ("seed", C.c_uint64), ("env_id_base", C.c_int64),
                ("total_step", C.c_int32), ("rules", C.c_int32), ("frame_value", C.c_int32), ("obs_scalars", C.c_int32),
                ("obs_tail", C.c_int32), ("reserved", C.c_int32)]


class State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("hdr", "episode", "grid", "plans", "plan_tb", "stat_episodes", "stat_return", "stat_iou_fx")]


class Sizes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("obs_dim", "num_actions", "total_step", "half_window", "env_height", "env_width",
                                         "plan_height", "plan_width", "grid_elems", "grid_elem_bytes", "plan_elems", "plan_elem_bytes")]


def test_raw_c_abi_reset_step_rollout_iou():
    import torch  # first: libsnac_hip.so shares PyTorch-ROCm's HIP runtime

    from snac_amd import plans

    L = C.CDLL(os.path.join(helpers.ROOT, "snac_amd", "libsnac_hip.so"))
    L.snac_last_error.restype = C.c_char_p
    assert L.snac_version() == 12
    sz = Sizes()
    assert L.snac_env_sizes(2, 1, C.byref(sz)) == 0 and (sz.obs_dim, sz.grid_elems, sz.grid_elem_bytes) == (51, 20, 4)

    N, seed = 1000, 77
    table = helpers.plan_table(2, True, "dense_train")
    packed, tb = plans.pack_plans(2, table.reshape(-1, 26, 26))
    dev = "cuda"
    hdr = torch.zeros((N, 4), dtype=torch.int32, device=dev)                 # snac_env_hdr[N]
    epi = torch.full((N,), -1, dtype=torch.int32, device=dev)
    grid = torch.zeros((N, sz.grid_elems), dtype=torch.int32, device=dev)
    d_plans = torch.from_numpy(packed.view(np.int32)).to(dev)
    d_tb = torch.from_numpy(tb).to(dev)
    stats = torch.zeros((3, N), dtype=torch.int64, device=dev)
    desc = EnvDesc(2, 1, N, len(packed), 0, 0, seed, 0, 0, 0)
    st = State(hdr.data_ptr(), epi.data_ptr(), grid.data_ptr(), d_plans.data_ptr(), d_tb.data_ptr(),
               stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    orc = helpers.oracle().OracleBatch(2, True, N, table, seed=seed)
    obs = torch.empty((N, 51), dtype=torch.float64, device=dev)
    assert L.snac_reset(C.byref(desc), C.byref(st), None, None, vp(obs), stream) == 0, L.snac_last_error()
    assert obs.cpu().numpy().tobytes() == orc.reset().tobytes()

    rew = torch.empty(N, dtype=torch.float32, device=dev)
    done = torch.empty(N, dtype=torch.uint8, device=dev)
    rng = np.random.default_rng(0)
    for t in range(50):
        a = rng.integers(0, 5, N).astype(np.int8)
        k = rng.integers(1, 4, N).astype(np.int8)
        da, dk = torch.from_numpy(a).to(dev), torch.from_numpy(k).to(dev)
        assert L.snac_step(C.byref(desc), C.byref(st), C.c_uint32(t), vp(da), vp(dk), 1, vp(obs), vp(rew), vp(done), stream) == 0
        oc, rc, dc = orc.step(t, a, k, auto_reset=True)
        assert obs.cpu().numpy().tobytes() == oc.tobytes() and rew.cpu().numpy().tobytes() == rc.tobytes()
        assert np.array_equal(done.cpu().numpy(), dc)

    T = 300
    traj = torch.empty((T, N, 51), dtype=torch.float64, device=dev)
    r2 = torch.empty((T, N), dtype=torch.float32, device=dev)
    d2 = torch.empty((T, N), dtype=torch.uint8, device=dev)
    assert L.snac_rollout(C.byref(desc), C.byref(st), T, C.c_uint32(50), None, None, 1, vp(traj), vp(r2), vp(d2), stream) == 0
    oc, rc, dc = orc.rollout(T, t0=50)
    assert traj.cpu().numpy().tobytes() == oc.tobytes() and r2.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(d2.cpu().numpy(), dc)

    iou = torch.empty(N, dtype=torch.float64, device=dev)
    assert L.snac_iou(C.byref(desc), C.byref(st), vp(iou), stream) == 0
    assert iou.cpu().numpy().tobytes() == orc.iou().tobytes()
    mem = torch.empty((N, 26, 26), dtype=torch.float64, device=dev)
    assert L.snac_export_grid(C.byref(desc), C.byref(st), vp(mem), stream) == 0
    assert np.array_equal(mem.cpu().numpy().reshape(N, -1), orc.state()["grid"])
    s = orc.stats()
    assert stats.sum(dim=1).tolist() == [int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum())]
    # errors are reported, not raised: obs_mode without a buffer
    assert L.snac_rollout(C.byref(desc), C.byref(st), 4, C.c_uint32(0), None, None, 1, None, None, None, stream) == -1
    assert b"obs" in L.snac_last_error()
