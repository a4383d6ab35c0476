"""GPU: the drop-in class surface (snac_amd.envs + the import shims + VectorizedEnvWrapper) against the golden
trajectories, driven the way the reference scripts drive the reference: np.random.seed(s), then reset()/step()."""
import importlib
import os
import sys

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

_MODS = {(1, False): ("1D", "DMP_Env_1D_static", "deep_mobile_printing_1d1r"),
         (1, True): ("1D", "DMP_Env_1D_dynamic_usedata_plan", "deep_mobile_printing_1d1r"),
         (2, False): ("2D", "DMP_Env_2D_static", "deep_mobile_printing_2d1r"),
         (2, True): ("2D", "DMP_Env_2D_dynamic_usedata_plan", "deep_mobile_printing_2d1r"),
         (3, False): ("3D", "DMP_simulator_3d_static_circle", "deep_mobile_printing_3d1r"),
         (3, True): ("3D", "DMP_simulator_3d_dynamic_triangle_usedata", "deep_mobile_printing_3d1r")}


def _cls(dim, dyn):
    """Import through the shim exactly like script/DQN/2d/DQN_2d_dynamic.py:8-11 does with the reference tree."""
    sub, mod, name = _MODS[(dim, dyn)]
    path = os.path.join(helpers.ROOT, "snac_amd", "Env", sub)
    if path not in sys.path:
        sys.path.append(path)
    return getattr(importlib.import_module(mod), name)


def _make(dim, dyn, tag, random_choose=True):
    cls = _cls(dim, dyn)
    if not dyn:
        return cls(plan_choose=int(tag[1:]))
    dens, split = tag.split("_")
    fname = "data_1d_dynamic_sin_envplan_500_%s.pkl" % split if dim == 1 else "data_%dd_dynamic_%s_envplan_500_%s.pkl" % (dim, dens, split)
    return cls(data_path=os.path.join("/nonexistent/Env/%dD" % dim, fname), random_choose_paln=random_choose)


def _primary(dim, dyn, obs):
    if not dyn:
        assert obs.shape == (1, helpers.DIMS[dim]["D"]) and obs.dtype == np.float64
        return obs.reshape(-1), None
    if dim == 1:
        assert obs[0].shape == (1, 7) and obs[1].shape == (1, 7) and obs[2].shape == (30,)
        return obs[1].reshape(-1), obs[0].reshape(-1)
    assert obs[0].shape == (1, 51) and obs[1].shape == (20, 20) and len(obs[2]) == 2
    return obs[0].reshape(-1), None


def _cases():
    out = []
    for dim, dyn, name in helpers.case_ids():
        mix = name.split(".")[1]
        # every mix for one plan set per env type, plus the sequential-plan cases; the rest is covered by
        # tests/test_gpu_parity.py::test_golden_replay_through_hip
        if mix == "sequential" or name.split(".")[0] in ("p0", "sin_train", "dense_train"):
            out.append((dim, dyn, name))
    return out


@pytest.mark.parametrize("dim,dyn,name", _cases(), ids=lambda v: str(v))
def test_facade_reproduces_reference_trajectory_from_seed(dim, dyn, name):
    rec = helpers.load_case(dim, dyn, name)
    tag = name.split(".")[0]
    np.random.seed(int(rec["seed"]))
    env = _make(dim, dyn, tag, random_choose=bool(rec["random_choose"]))
    starts = dict((int(s), e) for e, s in enumerate(rec["ep_start"]))
    S = min(len(rec["actions"]), 1400)
    W = helpers.DIMS[dim]["W"]
    for t in range(S):
        if t in starts:
            e = starts[t]
            o, raw = _primary(dim, dyn, env.reset())
            want = np.concatenate([rec["ep_reset_win"][e].astype(np.float64), rec["ep_reset_sc"][e]])
            assert o.tobytes() == want.tobytes()
            assert env.total_brick == rec["ep_total_brick"][e]
            if dyn and rec["random_choose"]:
                assert env.index_random == rec["ep_plan_idx"][e]
            assert len(env.position_memory) == 1
        obs, reward, done = env.step(int(rec["actions"][t]))
        o, raw = _primary(dim, dyn, obs)
        assert o.tobytes() == helpers.obs_from_golden(rec, t, dim).tobytes(), (name, t)
        if raw is not None:
            assert raw[:W].tobytes() == rec["win"][t].astype(np.float64).tobytes() and tuple(raw[W:]) == tuple(rec["sc_raw"][t])
        assert isinstance(reward, float) and reward == rec["reward"][t] and done is bool(rec["done"][t]), (name, t)
        assert env.step_size == rec["step_size"][t]                   # drawn from np.random's global stream
        assert env.count_step == rec["cs"][t] and env.count_brick == rec["cb"][t]
        pos = env.position_memory[-1]
        assert (pos == rec["pos"][t][0]) if dim == 1 else (tuple(pos) == tuple(rec["pos"][t]))
        if (t + 1) in starts or t == len(rec["actions"]) - 1:
            e = starts[t + 1] - 1 if (t + 1) in starts else len(rec["ep_start"]) - 1
            mem = env.environment_memory
            assert mem.dtype == np.float64 and mem.shape == ((1, 34) if dim == 1 else (26, 26))
            assert np.array_equal(mem.reshape(-1), rec["ep_final_grid"][e].astype(np.float64))
            assert np.float64(env.iou()).tobytes() == np.float64(rec["ep_iou"][e]).tobytes()
            if dim == 2:  # the caller-side IoU of script/DQN/2d/DQN_2d_dynamic.py:63-71 from the public attributes
                h = env.HALF_WINDOW_SIZE
                g = env.environment_memory[h:h + env.plan_height, h:h + env.plan_width]
                p = env.plan[h:h + env.plan_height, h:h + env.plan_width]
                assert np.sum(np.logical_and(g, p)) / np.sum(np.logical_or(g, p)) == rec["ep_iou"][e]


def test_reference_attribute_surface():
    env = _make(2, True, "dense_train")
    np.random.seed(1)
    s = env.reset()
    for a in ("HALF_WINDOW_SIZE", "plan", "plan_width", "plan_height", "total_brick", "environment_memory", "action_dim", "state_dim",
              "count_step", "total_step", "count_brick", "position_memory", "input_plan", "step_size", "one_hot", "index_random",
              "plan_dataset", "plan_dataset_len", "random_choose_paln", "environment_width", "environment_height"):
        assert hasattr(env, a), a
    assert (env.action_dim, env.state_dim, env.total_step, env.plan_dataset_len) == (5, 51, 600, 400)
    assert s[1].shape == (20, 20) and s[2] == [3, 3]
    e1 = _make(1, False, "p0")
    e1.reset()
    assert hasattr(e1, "conut_brick") and e1.state_dim == 7 and e1.total_brick == 600
    e3 = _make(3, True, "dense_train")
    assert (e3.z, e3.action_dim, e3.total_step) == (6, 8, 1000)
    assert _cls(3, False)().plan_choose == 1                          # the reference's default (static circle :8)


def test_invalid_action_raises_like_the_reference():
    for dim in (1, 2):
        env = _make(dim, False, "p0")
        np.random.seed(3)
        env.reset()
        env.step(0)
        before = np.random.get_state()[2]
        with pytest.raises(UnboundLocalError):
            env.step(env.action_dim)
        assert env.count_step == 2                                    # count_step advanced before the error
        assert np.random.get_state()[2] != before                     # and so did the global RNG
    with pytest.raises(ValueError):
        _cls(1, False)(plan_choose=5).reset()


@pytest.mark.parametrize("n", [24, 200, 260])                         # one resident wave; four (round 6: up to 256 envs); beyond: the launch path (rows in page-locked host memory)
@pytest.mark.parametrize("name,kind", [("1DStatic", (1, False)), ("2DDynamic", (2, True)), ("3DDynamic", (3, True))])
def test_vectorized_wrapper_follows_the_reference_loop(name, kind, n):
    """multiprocess.py:78-84 with N independent envs: seed, reset, T ticks of np.random actions.  The oracle side
    replays numpy's stream (MT19937 restatement) in the same order: N plan draws, then per tick N action draws
    and N step-size draws."""
    from snac_amd import VectorizedEnvWrapper
    from snac_amd.multiprocess import make_plans

    dim, dyn = kind
    T, seed = 120 if n < 100 else 40, 5
    orc_mod = helpers.oracle()
    plans = make_plans(name, 0)
    np.random.seed(seed)
    env = VectorizedEnvWrapper(plans, num_envs=n)
    assert (env._mrows is not None) == (n <= 256)                     # the resident waves up to 256 envs, the launch path beyond
    obs = env.reset()
    A = env.action_dim
    table = np.ascontiguousarray(plans[2].reshape(len(plans[2]), -1), np.int32)
    mt = orc_mod.MT19937(seed)
    orc = orc_mod.OracleBatch(dim, dyn, n, table)
    pidx = [mt.randint(0, len(table)) for _ in range(n)] if dyn else None
    assert obs.shape == (n, 1, env.batched.obs_dim)
    assert obs.tobytes() == orc.reset(plan_idx=pidx).tobytes()
    for t in range(T):
        actions = np.random.randint(A, size=n)
        o, r, d = env.step(actions)
        a2 = [mt.randint(0, A) for _ in range(n)]
        k2 = [mt.randint(1, 4) for _ in range(n)]
        assert list(actions) == a2
        oc, rc, dc = orc.step(t, np.asarray(a2, np.int8), np.asarray(k2, np.int8))
        assert o.shape == (n, 1, env.batched.obs_dim) and r.shape == (n,) and d.shape == (n,)
        assert o.tobytes() == oc.tobytes() and np.array_equal(r, rc.astype(np.float64)) and np.array_equal(d.astype(np.uint8), dc)
    one = env.reset_at(3)
    assert one.shape == (1, env.batched.obs_dim)
    assert int(env.batched.count_step[3]) == 0 and int(env.batched.count_step[4]) == T
