"""CPU-side checks (no GPU, no compute launches): the C-ABI library loads and exports every symbol the header
declares, argument validation, the host-side plan packing, sharding arithmetic, and that the product refuses to
run without its HIP path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers


def _header_functions():
    src = open(os.path.join(helpers.ROOT, "include", "snac_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(snac_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from snac_amd import _lib

    L = _lib.lib()
    names = _header_functions()
    assert set(names) == set(_lib.EXPORTS), names
    for n in names:
        assert hasattr(L, n), n
    assert L.snac_version() == _lib.ABI_VERSION


def test_env_sizes_match_reference_constants():
    from snac_amd import _lib

    want = {(1, 0): (7, 3, 750, 2, 1, 34, 1, 30), (1, 1): (7, 3, 750, 2, 1, 34, 1, 30),
            (2, 0): (51, 5, 600, 3, 26, 26, 20, 20), (2, 1): (51, 5, 600, 3, 26, 26, 20, 20),
            (3, 0): (51, 8, 1300, 3, 26, 26, 20, 20), (3, 1): (51, 8, 1000, 3, 26, 26, 20, 20)}
    for (k, d), w in want.items():
        s = _lib.env_sizes(k, d)
        assert (s.obs_dim, s.num_actions, s.total_step, s.half_window, s.env_height, s.env_width, s.plan_height, s.plan_width) == w
    with pytest.raises(_lib.SnacError):
        _lib.env_sizes(7, 0)


def test_argument_validation_happens_before_any_launch():
    from snac_amd import _lib

    L = _lib.lib()
    d = _lib.EnvDesc(2, 1, 16, 4, 0, 0, 1, 0, 0, 0)
    st = _lib.State(0, 0, 0, 0, 0, 0, 0, 0)   # null pointers
    assert L.snac_step(C.byref(d), C.byref(st), 0, None, None, 0, None, None, None, None) == -1
    assert b"null pointer" in L.snac_last_error()
    bad = _lib.EnvDesc(9, 1, 16, 4, 0, 0, 1, 0, 0, 0)
    assert L.snac_reset(C.byref(bad), C.byref(st), None, None, None, None) == -1
    assert b"kind" in L.snac_last_error()
    bad = _lib.EnvDesc(2, 1, 0, 4, 0, 0, 1, 0, 0, 0)
    assert L.snac_iou(C.byref(bad), C.byref(st), None, None) == -1
    assert L.snac_rollout(None, None, 1, 0, None, None, 0, None, None, None, None) == -1
    # trajectory memory: arguments first, and without a GPU the call reports an error instead of handing out host memory
    p = C.c_void_p()
    assert L.snac_traj_alloc(0, 0, C.byref(p)) == -1 and b"bytes" in L.snac_last_error()
    assert L.snac_traj_alloc(1 << 20, 0, None) == -1
    assert L.snac_traj_free(None) == 0
    assert L.snac_traj_free(C.c_void_p(0x1000)) == -1 and b"snac_traj_alloc" in L.snac_last_error()
    import torch
    if not torch.cuda.is_available():
        assert L.snac_traj_alloc(1 << 20, 0, C.byref(p)) < 0 and not p.value


def test_hdr_struct_is_16_bytes():
    src = open(os.path.join(helpers.ROOT, "include", "snac_hip.h")).read()
    assert "int16_t ep_return" in src and "int16_t cross" in src

    class Hdr(C.Structure):
        _fields_ = [("pos_r", C.c_int8), ("pos_c", C.c_int8), ("flags", C.c_uint8), ("reserved", C.c_uint8),
                    ("cb", C.c_int16), ("cs", C.c_int16), ("tb", C.c_int16), ("pidx", C.c_int16), ("ep", C.c_int16), ("cross", C.c_int16)]

    assert C.sizeof(Hdr) == 16


def test_static_plans_and_packing():
    from snac_amd import plans

    z = helpers.static_plans_npz()
    for pc in (0, 1, 2):
        assert np.array_equal(plans.static_plan(1, pc), z["1d_p%d" % pc])
    for pc in (0, 1):
        assert np.array_equal(plans.static_plan(2, pc), z["2d_p%d" % pc])
        assert np.array_equal(plans.static_plan(3, pc), z["3d_p%d" % pc])
    with pytest.raises(ValueError):
        plans.static_plan(2, 2)
    # 2D: bit j of row i = plan[3 + i, 3 + j]; total_brick floored at 30
    full = plans.dataset(2, "sparse", "train")
    packed, tb = plans.pack_plans(2, full)
    assert packed.shape == (400, 20) and packed.dtype == np.uint32
    for p in (0, 17, 399):
        bits = (packed[p][:, None] >> np.arange(20, dtype=np.uint32)[None, :]) & 1
        assert np.array_equal(bits, full[p, 3:23, 3:23].astype(np.uint32))
        assert tb[p] == max(int(full[p].sum()), 30)
    assert (full.reshape(400, -1).sum(1) < 30).sum() == 197
    packed, tb = plans.pack_plans(3, plans.dataset(3, "dense", "train"))
    assert packed.shape == (400, 400) and tb.min() == 306 and tb.max() == 654
    packed, tb = plans.pack_plans(1, plans.dataset(1))
    assert packed.shape == (400, 32) and tb.min() == 592 and tb.max() == 604 and np.all(packed[:, 30:] == 0)
    bad = np.zeros((1, 26, 26))
    bad[0, 0, 0] = 1
    with pytest.raises(ValueError):
        plans.pack_plans(2, bad)
    with pytest.raises(ValueError):
        plans.pack_plans(2, np.full((1, 26, 26), 0.5))


def test_shard_arithmetic():
    from snac_amd import dist

    for total, world in ((524288, 8), (10, 3), (7, 8), (65536, 1)):
        spans = [dist.shard(total, r, world) for r in range(world)]
        assert sum(n for n, _ in spans) == total
        pos = 0
        for n, base in spans:
            assert base == pos
            pos += n
    assert dist.shard(524288, 3, 8) == (65536, 3 * 65536)
    with pytest.raises(ValueError):
        dist.shard(8, 8, 8)


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from snac_amd import SnacError
    from snac_amd.batched import BatchedDMPEnv
    from snac_amd.envs import deep_mobile_printing_2d1r_static

    with pytest.raises(SnacError):
        BatchedDMPEnv(2, True, 8)
    with pytest.raises(SnacError):
        deep_mobile_printing_2d1r_static(plan_choose=0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(helpers.ROOT, "snac_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dirpath, f)


def test_bench_gpus_flag_must_match_the_world_size():
    """bench.py --gpus N under a torch.distributed environment of another size exits non-zero before touching torch."""
    import subprocess
    import sys

    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr and not r.stdout.strip()


def test_dispatch_table_is_one_table_with_documented_entries():
    """snac_tuning(): every batch-size threshold / switch that selects a kernel, with its effective value and what it decides (host-only
    call: no GPU needed); the defaults the tests of both sides of each threshold rely on."""
    from snac_amd import _lib

    t = _lib.tuning()
    assert len(t) >= 62 and all(k.startswith("SNAC_") and what for k, (v, what) in t.items())
    want = {"SNAC_3D_PIPELINE": 1, "SNAC_2D_STAGE_MIN_F64": 32769, "SNAC_2D_STAGE_MIN_F32": 32768, "SNAC_2D_TP_MAX_F64": 19456, "SNAC_2D_TP_GAP_LO": 15873,
            "SNAC_2D_TP_GAP_HI": 16384, "SNAC_2D_TP_MAX_F32": 30719, "SNAC_2D_TP_VAR_MAX_PLAN": 49152, "SNAC_2D_TP_VAR_MAX_SHORT": 6144, "SNAC_2D_TILE32_MIN": 22528,
            "SNAC_1D_TP_MAX_F64": 65536, "SNAC_1D_TP_EB8_MIN": 3584, "SNAC_NODES2D_NT": 1, "SNAC_STEP3D_NTLOAD_MIN": 376832, "SNAC_STEP3D_HUGE_MIN": 557056, "SNAC_STEP3D_HUGE_FORM": 3, "SNAC_STEP3D_FORM": -1, "SNAC_STEP2D_PLAIN_LO": 20480, "SNAC_STEP2D_RES_HI": 278528, "SNAC_STEP2D_PLAIN_HI": 475136, "SNAC_STEP2D_HUGE_MIN": 475137, "SNAC_STEP2D_HUGE_FORM": 2, "SNAC_STEP2D_FORM": -1, "SNAC_1D_TP_MAX_F32": 65536, "SNAC_3D_BLOCK_MIN_F64": 4096, "SNAC_3D_BLOCK_MIN_F32": 4096,
            "SNAC_STEP3D_SPAN_MIN": 81920, "SNAC_3D_BLOCK_VAR": 1, "SNAC_3D_BLOCK_VAR_MIN": 64, "SNAC_3D_BLOCK_VAR_PLAN_F64": 10240, "SNAC_3D_BLOCK_VAR_PLAN_F32": 16384,
            "SNAC_2D_BLOCK": 1, "SNAC_2D_BLOCK_MIN_F64": 11264, "SNAC_2D_BLOCK_MAX_F64": 32768, "SNAC_2D_BLOCK_MIN_F32": 15360, "SNAC_2D_BLOCK_MAX_F32": 32768,
            "SNAC_2D_BLOCK_TWO_F64": 16384, "SNAC_2D_BLOCK_TWO_F32": 16385, "SNAC_2D_BLOCK_VAR_MIN": 6148, "SNAC_2D_BLOCK_VAR_MAX": 32768, "SNAC_2D_BLOCK_VAR_TWO": 16385, "SNAC_STEP3D_QUARTER": 1, "SNAC_STEP3D_QUARTER_MIN": 4, "SNAC_STEP3D_QUARTER_MAX": 1 << 30, "SNAC_2D_BLOCK_FOUR_MAX_F64": 38912, "SNAC_2D_BLOCK_FOUR_MAX_F32": 45056, "SNAC_STEP_VAR_MIN_SHORT": 24576, "SNAC_STEP_VAR_FULL_F64": 45056, "SNAC_STEP_VAR_FULL_F32": 32769, "SNAC_STEP_VAR3_MIN": 24576}
    import os
    for k, v in want.items():
        if k not in os.environ:
            assert t[k][0] == v, (k, t[k])


def test_round5_entry_points_validate_their_arguments_before_any_hip_call():
    from snac_amd import _lib

    L = _lib.lib()
    mb = C.c_void_p()
    big = _lib.EnvDesc(2, 1, 257, 4, 0, 0, 1, 0, 0, 0)             # a mailbox steps four wavefronts: 256 envs at most
    assert L.snac_mailbox_create(C.byref(big), 0, C.byref(mb)) != 0 and b"1 .. 256" in L.snac_last_error() and not mb.value
    assert L.snac_mailbox_create(None, 0, C.byref(mb)) != 0
    assert L.snac_mailbox_step(None, None, None, 0, 1) != 0 and b"null mailbox" in L.snac_last_error()
    assert L.snac_mailbox_touch(None) != 0
    assert L.snac_mailbox_settle(None) == 0 and L.snac_mailbox_quit(None) == 0 and L.snac_mailbox_destroy(None) == 0   # nothing to wait for / end
    assert not L.snac_mailbox_row(None) and not L.snac_mailbox_reward(None) and not L.snac_mailbox_done(None)
    d = _lib.EnvDesc(2, 1, 16, 4, 0, 0, 1, 0, 0, 0)
    st = _lib.State(1, 1, 1, 1, 1, 1, 1, 1)                        # (never dereferenced: every call below fails its checks first)
    assert L.snac_plans_from_grids(C.byref(d), C.byref(st), 2, 3, None, 0, None, None, None, None) != 0 and b"out of range" in L.snac_last_error()
    assert L.snac_plans_from_grids(C.byref(d), C.byref(st), 2, 0, None, 0, None, None, None, None) != 0 and b"exactly one" in L.snac_last_error()
    mem = (C.c_double * (2 * 676))()
    assert L.snac_plans_from_grids(C.byref(d), C.byref(st), 2, 0, None, 0, None, mem, None, None) != 0 and b"total_brick" in L.snac_last_error()
    buf = C.create_string_buffer(8)
    assert L.snac_tuning(buf, 8) != 0 and L.snac_tuning(None, 0) != 0
