"""Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py :: deep_mobile_printing_2d1r_hindsight -- the last drop-in class
without a recorded reference trajectory until round 3.  Its reset() draws a throw-away random triangle with cv2, which is not
installed where goldens are recorded; tests/golden/make_golden_hindsight_2d_dynamic.py imports the reference class with a `cv2`
stand-in that draws with the build's restatement of cv2's rules (pinned by the 2000 dataset plans, tests/test_plan_generators.py),
so the recording pins the class's OWN logic -- the np.random consumption of reset() (redraws until the area passes 50 / 20, then the
index draw), sequential plan order, step(action, step_size), raw counters -- "pinned modulo rasteriser".
CPU: the oracle + a RandomState twin replay the goldens.  GPU: the drop-in class replays them from np.random.seed alone, and the
batched path with the recorded plan rows."""
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
        _Z = np.load(os.path.join(helpers.GOLDEN, "traj_hindsight_dynamic_2d.npz"))
    return _Z


def _names():
    return _file()["cases"].tolist()


def _rec(name):
    z = _file()
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def _table(name):
    dens, split = name.split(".")[1].split("-")
    return helpers.plan_table(2, True, "%s_%s" % (dens, split)), dens == "sparse"


def _replay(name, reset, step, state):
    rec = _rec(name)
    starts = dict((int(s), e) for e, s in enumerate(rec["ep_start"]))
    S = len(rec["actions"])
    for t in range(S):
        if t in starts:
            e = starts[t]
            o = np.asarray(reset(e, rec), np.float64).reshape(-1)
            assert o.tobytes() == np.concatenate([rec["ep_reset_win"][e].astype(np.float64), rec["ep_reset_sc"][e]]).tobytes()
        o, r, d, pos = step(int(rec["actions"][t]), int(rec["step_size"][t]))
        want = np.concatenate([rec["win"][t].astype(np.float64), rec["sc"][t]])
        assert np.asarray(o, np.float64).reshape(-1).tobytes() == want.tobytes(), (name, t)
        assert r == rec["reward"][t] and bool(d) == bool(rec["done"][t]) and list(pos) == list(rec["pos"][t]), (name, t)
        if (t + 1) in starts or t == S - 1:
            e = starts[t + 1] - 1 if (t + 1) in starts else len(rec["ep_start"]) - 1
            grid, iou = state()
            assert np.array_equal(np.asarray(grid).reshape(-1), rec["ep_final_grid"][e].astype(np.float64))
            assert np.float64(iou).tobytes() == np.float64(rec["ep_iou"][e]).tobytes()


@pytest.mark.parametrize("name", _names())
def test_oracle_and_rng_twin_replay_the_recording(name):
    """The oracle env (dataset dynamics, raw counters, caller's step sizes) reproduces every observation / reward / done; a
    RandomState twin that consumes numpy's stream the way reset() does -- two randint(0, 20, size=3) per attempt while the oracle's
    rasteriser reports an area <= 50 (dense) / 20 (sparse), then randint(0, len) for the plan when the choice is random -- lands on
    the recorded plan rows and on the recorded next word of the global stream."""
    orc = helpers.oracle()
    rec0 = _rec(name)
    table, sparse = _table(name)
    env = orc.OracleEnv(2, True).configure(obs_norm=0, rules_dyn=1)
    st = np.random.RandomState(int(rec0["seed"]))
    random_choose = bool(int(rec0["random_choose"]))
    seq = [0]

    def reset(e, rec):
        while True:                                               # create_plan(): the throw-away triangle
            x, y = st.randint(0, 20, size=3), st.randint(0, 20, size=3)
            if orc.raster_triangle(x, y, int(sparse))[1] > (20 if sparse else 50):
                break
        if random_choose:
            idx = int(st.randint(0, len(table)))
        else:
            idx = seq[0]
            seq[0] = (seq[0] + 1) % len(table)
        assert idx == rec["ep_plan_idx"][e], (name, e)
        o = env.reset(table[idx].reshape(-1), idx)
        assert env.e.tb == rec["ep_total_brick"][e]
        return o

    def step(a, k):
        o, r, d = env.step(a, k)
        return o, r, d, env.pos

    _replay(name, reset, step, lambda: (env.grid.astype(np.float64), env.iou()))
    assert int(st.randint(0, 1 << 30)) == int(rec0["rng_after"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names())
def test_drop_in_class_replays_the_recording_from_the_seed(name):
    path = os.path.join(helpers.ROOT, "snac_amd", "Env", "2D")
    if path not in sys.path:
        sys.path.append(path)
    cls = getattr(importlib.import_module("DMP_Env_2D_dynamic_hindsight_replay_usedata"), "deep_mobile_printing_2d1r_hindsight")
    rec0 = _rec(name)
    dens, split = name.split(".")[1].split("-")
    np.random.seed(int(rec0["seed"]))
    env = cls(data_path="/nonexistent/data_2d_dynamic_%s_envplan_500_%s.pkl" % (dens, split), random_choose_paln=bool(int(rec0["random_choose"])))

    def reset(e, rec):
        obs = env.reset()
        assert len(obs) == 3 and obs[1] is env.input_plan and list(obs[2]) == [3, 3]
        assert int(env.total_brick) == rec["ep_total_brick"][e]
        if env.random_choose_paln:
            assert env.index_random == rec["ep_plan_idx"][e]
        return obs[0]

    def step(a, k):
        obs, r, d = env.step(a, k)
        assert len(obs) == 3 and obs[1] is env.input_plan and list(obs[2]) == list(env.position_memory[-1]) and env.step_size == k
        return obs[0], r, d, env.position_memory[-1]

    def iou():
        g, p = env.environment_memory[3:23, 3:23], env.plan[3:23, 3:23]
        return float(np.sum(np.logical_and(g, p)) / np.sum(np.logical_or(g, p)))

    _replay(name, reset, step, lambda: (env.environment_memory, iou()))
    assert int(np.random.randint(0, 1 << 30)) == int(rec0["rng_after"])      # the global stream stands where the reference's stood


@pytest.mark.gpu
def test_batched_path_replays_all_recordings_at_once():
    """One BatchedDMPEnv row per recording that shares a dataset (raw counters, the recorded plan rows / actions / step sizes as
    explicit inputs, masked resets where an episode of that row begins): the batched kernels write the recorded rows."""
    import torch
    from snac_amd import BatchedDMPEnv

    for name in _names():
        rec = _rec(name)
        table, _ = _table(name)
        env = BatchedDMPEnv(2, True, 4, plans=table.reshape(len(table), 26, 26), obs_scalars="raw", seed=1)     # four copies of the one env
        starts = dict((int(s), e) for e, s in enumerate(rec["ep_start"]))
        for t in range(len(rec["actions"])):
            if t in starts:
                e = starts[t]
                o = env.reset(plan_idx=np.full(4, rec["ep_plan_idx"][e]))
                want = np.concatenate([rec["ep_reset_win"][e].astype(np.float64), rec["ep_reset_sc"][e]])
                assert o[3].cpu().numpy().tobytes() == want.tobytes()
                assert int(env.total_brick[0]) == rec["ep_total_brick"][e]
            a = torch.full((4,), int(rec["actions"][t]), dtype=torch.int8)
            k = torch.full((4,), int(rec["step_size"][t]), dtype=torch.int8)
            o, r, d = env.step(a, k)
            want = np.concatenate([rec["win"][t].astype(np.float64), rec["sc"][t]])
            assert o[0].cpu().numpy().tobytes() == want.tobytes() and o[3].cpu().numpy().tobytes() == want.tobytes(), (name, t)
            assert float(r[2]) == rec["reward"][t] and bool(d[1]) == bool(rec["done"][t]), (name, t)
