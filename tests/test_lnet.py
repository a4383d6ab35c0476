"""The reference's L-Net env variants (Env/1D/DMP_Env_1D_static_Lnet.py, Env/2D/DMP_Env_2D_static_Lnet.py,
Env/3D/DMP_simulator_3d_static_circle_Lnet.py): goldens recorded by tests/golden/make_golden_lnet.py replayed through the
CPU oracle (CPU test) and, from np.random.seed alone, through the drop-in classes on the HIP path (GPU test)."""
import importlib
import os
import sys

import numpy as np
import pytest

import helpers

_Z = None


def _file():
    global _Z
    if _Z is None:
        _Z = np.load(os.path.join(helpers.GOLDEN, "traj_lnet.npz"))
    return _Z


def _names():
    return _file()["cases"].tolist()


def _rec(name):
    z = _file()
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


@pytest.mark.parametrize("name", _names())
def test_oracle_replays_lnet_goldens(name):
    orc = helpers.oracle()
    rec = _rec(name)
    dim, pc = int(name[0]), int(name.split(".")[1][1:])
    env = orc.OracleEnv(dim, False)
    if dim == 2:
        env.configure(obs_norm=1, rules_dyn=0, frame=2)           # Env/2D/DMP_Env_2D_static_Lnet.py:61-64,75-76
    elif dim == 3:
        env.configure(obs_norm=1, rules_dyn=1, total_step=1300)   # Env/3D/DMP_simulator_3d_static_circle_Lnet.py:28,210-236
    plan = orc.static_plan(dim, pc)
    starts = dict((int(s), e) for e, s in enumerate(rec["ep_start"]))
    S = len(rec["actions"])
    for t in range(S):
        if t in starts:
            e = starts[t]
            o = env.reset(plan)
            assert o.tobytes() == np.concatenate([rec["ep_reset_win"][e].astype(np.float64), rec["ep_reset_sc"][e]]).tobytes()
            assert env.e.tb == rec["ep_total_brick"][e]
        o, r, d = env.step(int(rec["actions"][t]), int(rec["step_size"][t]))
        assert o.tobytes() == np.concatenate([rec["win"][t].astype(np.float64), rec["sc"][t]]).tobytes(), (name, t)
        assert r == rec["reward"][t] and d == bool(rec["done"][t]), (name, t)
        assert (env.pos[0] == rec["pos"][t][0]) if dim == 1 else (env.pos == tuple(rec["pos"][t]))
        if (t + 1) in starts or t == S - 1:
            e = starts[t + 1] - 1 if (t + 1) in starts else len(rec["ep_start"]) - 1
            assert np.array_equal(env.grid, rec["ep_final_grid"][e].astype(np.int32))
            assert np.float64(env.iou()).tobytes() == np.float64(rec["ep_iou"][e]).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names())
def test_lnet_facades_on_hip_from_seed(name):
    mods = {1: ("1D", "DMP_Env_1D_static_Lnet"), 2: ("2D", "DMP_Env_2D_static_Lnet_test"), 3: ("3D", "DMP_simulator_3d_static_circle_Lnet")}
    rec = _rec(name)
    dim, pc = int(name[0]), int(name.split(".")[1][1:])
    sub, mod = mods[dim]
    path = os.path.join(helpers.ROOT, "snac_amd", "Env", sub)
    if path not in sys.path:
        sys.path.append(path)
    cls = getattr(importlib.import_module(mod), "deep_mobile_printing_%dd1r" % dim)
    np.random.seed(int(rec["seed"]))
    env = cls(plan_choose=pc)
    starts = dict((int(s), e) for e, s in enumerate(rec["ep_start"]))
    S = min(len(rec["actions"]), 1500)
    W = helpers.DIMS[dim]["W"]

    def split(obs):
        if dim == 1:
            o = np.asarray(obs, np.float64)
            assert o.shape == (1, 8)
            return o.reshape(-1)[:7], int(o[0, 7])
        assert obs[0].shape == (1, 51) and len(obs) == 2
        return obs[0].reshape(-1), tuple(obs[1])

    for t in range(S):
        if t in starts:
            e = starts[t]
            o, p = split(env.reset())
            assert o.tobytes() == np.concatenate([rec["ep_reset_win"][e].astype(np.float64), rec["ep_reset_sc"][e]]).tobytes()
            assert p == (2 if dim == 1 else (3, 3))
        obs, r, d = env.step(int(rec["actions"][t]))
        o, p = split(obs)
        assert o.tobytes() == np.concatenate([rec["win"][t].astype(np.float64), rec["sc"][t]]).tobytes(), (name, t)
        assert r == rec["reward"][t] and d == bool(rec["done"][t]) and env.step_size == rec["step_size"][t], (name, t)
        assert (p == rec["pos"][t][0]) if dim == 1 else (p == tuple(rec["pos"][t]))
        if (t + 1) in starts:
            e = starts[t + 1] - 1
            assert np.array_equal(env.environment_memory.reshape(-1), rec["ep_final_grid"][e].astype(np.float64))
            assert np.float64(env.iou()).tobytes() == np.float64(rec["ep_iou"][e]).tobytes()
    assert env.total_step == (750, 600, 1300)[dim - 1]
