"""The reference's static hindsight-replay env variants (step(action, step_size)): goldens recorded from
Env/*/DMP_*_static_hindsight_replay.py (tests/golden/make_golden_hindsight.py) replayed through the CPU oracle (CPU
test) and through the drop-in classes on the HIP path (GPU test)."""
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
        _Z = np.load(os.path.join(helpers.GOLDEN, "traj_hindsight_static.npz"))
    return _Z


def _names():
    return _file()["cases"].tolist()


def _rec(name):
    z = _file()
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def _replay(name, make_env, step, reset, state):
    rec = _rec(name)
    dim, pc = int(name[0]), int(name.split(".")[1][1:])
    env = make_env(dim, pc)
    starts = dict((int(s), e) for e, s in enumerate(rec["ep_start"]))
    S = len(rec["actions"])
    W = helpers.DIMS[dim]["W"]
    for t in range(S):
        if t in starts:
            e = starts[t]
            o = reset(env)
            want = np.concatenate([rec["ep_reset_win"][e].astype(np.float64), rec["ep_reset_sc"][e]])
            assert np.asarray(o, np.float64).reshape(-1).tobytes() == want.tobytes()
        o, r, d = step(env, int(rec["actions"][t]), int(rec["step_size"][t]))
        want = np.concatenate([rec["win"][t].astype(np.float64), rec["sc"][t]])
        assert np.asarray(o, np.float64).reshape(-1).tobytes() == want.tobytes(), (name, t)
        assert r == rec["reward"][t] and bool(d) == bool(rec["done"][t]), (name, t)
        if (t + 1) in starts or t == S - 1:
            e = starts[t + 1] - 1 if (t + 1) in starts else len(rec["ep_start"]) - 1
            grid, iou, pos = state(env)
            assert np.array_equal(np.asarray(grid).reshape(-1), rec["ep_final_grid"][e].astype(np.float64))
            assert np.float64(iou).tobytes() == np.float64(rec["ep_iou"][e]).tobytes()


@pytest.mark.parametrize("name", _names())
def test_oracle_replays_hindsight_goldens(name):
    orc = helpers.oracle()
    plan = {}

    def make(dim, pc):
        plan["p"] = orc.static_plan(dim, pc)
        return orc.OracleEnv(dim, False)

    _replay(name, make, lambda e, a, k: e.step(a, k), lambda e: e.reset(plan["p"]),
            lambda e: (e.grid.astype(np.float64), e.iou(), e.pos))


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names())
def test_hindsight_facades_on_hip(name):
    mods = {1: ("1D", "DMP_Env_1D_static_hindsight_replay"), 2: ("2D", "DMP_Env_2D_static_hindsight_replay"),
            3: ("3D", "DMP_simulator_3d_static_circle_hindsight_replay")}

    def make(dim, pc):
        sub, mod = mods[dim]
        path = os.path.join(helpers.ROOT, "snac_amd", "Env", sub)
        if path not in sys.path:
            sys.path.append(path)
        cls = getattr(importlib.import_module(mod), "deep_mobile_printing_%dd1r_hindsight" % dim)
        return cls(plan_choose=pc)

    state0 = np.random.get_state()[1].copy()
    _replay(name, make, lambda e, a, k: e.step(a, k), lambda e: e.reset(),
            lambda e: (e.environment_memory, e.iou(), e.position_memory[-1]))
    assert np.array_equal(np.random.get_state()[1], state0)       # the hindsight classes never touch numpy's global stream


# ---- 1D dynamic hindsight: a random sin plan per reset (Env/1D/DMP_Env_1D_dynamic_hindsight_replay.py) -----------------
_ZD = None


def _dyn_file():
    global _ZD
    if _ZD is None:
        _ZD = np.load(os.path.join(helpers.GOLDEN, "traj_hindsight_dynamic_1d.npz"))
    return _ZD


def _dyn_rec(name):
    z = _dyn_file()
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def _dyn_replay(name, env, reset, step, state):
    rec = _dyn_rec(name)
    starts = dict((int(s), e) for e, s in enumerate(rec["ep_start"]))
    S = len(rec["actions"])
    for t in range(S):
        if t in starts:
            e = starts[t]
            o = reset(env, e, rec)
            want = np.concatenate([rec["ep_reset_win"][e].astype(np.float64), rec["ep_reset_sc"][e]])
            assert np.asarray(o, np.float64).reshape(-1).tobytes() == want.tobytes()
        o, r, d = step(env, int(rec["actions"][t]), int(rec["step_size"][t]))
        want = np.concatenate([rec["win"][t].astype(np.float64), rec["sc"][t]])
        assert np.asarray(o, np.float64).reshape(-1).tobytes() == want.tobytes(), (name, t)
        assert r == rec["reward"][t] and bool(d) == bool(rec["done"][t]), (name, t)
        if (t + 1) in starts or t == S - 1:
            e = starts[t + 1] - 1 if (t + 1) in starts else len(rec["ep_start"]) - 1
            grid, iou = state(env)
            assert np.array_equal(np.asarray(grid).reshape(-1), rec["ep_final_grid"][e].astype(np.float64))
            assert np.float64(iou).tobytes() == np.float64(rec["ep_iou"][e]).tobytes()


@pytest.mark.parametrize("name", _dyn_file()["cases"].tolist())
def test_oracle_replays_dynamic_hindsight_goldens(name):
    orc = helpers.oracle()

    def reset(env, e, rec):
        o = env.reset(rec["ep_plan"][e].astype(np.int32))
        assert env.e.tb == rec["ep_total_brick"][e]
        return o

    _dyn_replay(name, orc.OracleEnv(1, False), reset, lambda e, a, k: e.step(a, k), lambda e: (e.grid.astype(np.float64), e.iou()))


def test_random_sin_plan_consumes_the_stream_like_the_reference():
    """plans.random_sin_plan against the recorded plans, from the seed alone (host-side numpy: uniform, randint, uniform)."""
    from snac_amd import plans

    for name in _dyn_file()["cases"].tolist():
        rec = _dyn_rec(name)
        np.random.seed(int(rec["seed"]))
        y, area, hot = plans.random_sin_plan()
        assert y.tobytes() == rec["ep_plan"][0].tobytes() and area == rec["ep_total_brick"][0]
        assert np.asarray(hot, np.float64).tobytes() == rec["ep_one_hot"][0].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("name", _dyn_file()["cases"].tolist())
def test_dynamic_hindsight_facade_on_hip(name):
    path = os.path.join(helpers.ROOT, "snac_amd", "Env", "1D")
    if path not in sys.path:
        sys.path.append(path)
    cls = getattr(importlib.import_module("DMP_Env_1D_dynamic_hindsight_replay"), "deep_mobile_printing_1d1r_hindsight")
    rec0 = _dyn_rec(name)
    np.random.seed(int(rec0["seed"]))
    env = cls()

    def reset(env, e, rec):
        obs = env.reset()
        assert obs[1] is env.plan and env.plan.tobytes() == rec["ep_plan"][e].tobytes()
        assert env.total_brick == rec["ep_total_brick"][e]
        assert np.asarray(env.one_hot, np.float64).tobytes() == rec["ep_one_hot"][e].tobytes()
        return obs[0]

    def step(env, a, k):
        obs, r, d = env.step(a, k)
        assert obs[1] is env.plan and obs[0].shape == (1, 7)
        return obs[0], r, d

    _dyn_replay(name, env, reset, step, lambda e: (e.environment_memory, e.iou()))
