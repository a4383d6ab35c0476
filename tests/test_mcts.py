"""The reference's MCTS env variants (Env/*/DMP_*_MCTS*.py, nine files): goldens recorded by
tests/golden/make_golden_mcts.py -- step() and the functional transition(state, action) on recorded input states --
replayed through the CPU oracle (CPU test) and, from np.random.seed alone, through the drop-in classes on the HIP path
(GPU test, tests/test_gpu_mcts.py)."""
import os

import numpy as np
import pytest

import helpers

_Z = None


def golden():
    global _Z
    if _Z is None:
        _Z = np.load(os.path.join(helpers.GOLDEN, "traj_mcts.npz"))
    return _Z


def names():
    return golden()["cases"].tolist()


def rec(name):
    z = golden()
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def variant(name):
    dim = int(name[0])
    kind = name.split(".")[1]
    return dim, kind == "dynamic", kind


@pytest.mark.parametrize("name", names())
def test_oracle_replays_mcts_goldens(name):
    orc = helpers.oracle()
    r = rec(name)
    dim, dyn, kind = variant(name)
    gated = dim == 3 and dyn          # Env/3D/DMP_simulator_3d_dynamic_triangle_MCTS.py:258
    assert int(r["total_step"]) == {1: 750, 2: 600, 3: 1000 if dyn else 1300}[dim]
    envs = []
    for e in range(len(r["ep_total_brick"])):
        env = orc.OracleEnv(dim, dyn).configure(obs_norm=0, rules_dyn=int(dyn))   # raw count_brick / count_step in every variant
        o = env.reset(r["ep_plan"][e].astype(np.int32), int(r["ep_plan_idx"][e]))
        assert o.tobytes() == r["ep_reset_obs"][e].tobytes()
        assert env.e.tb == r["ep_total_brick"][e]
        envs.append(env)
    n_edit = 0
    for t in range(len(r["op"])):
        env = envs[int(r["episode"][t])]
        pos = int(r["in_pos"][t][0]) if dim == 1 else r["in_pos"][t]
        env.set_state(r["in_grid"][t], pos, int(r["in_cb"][t]), int(r["in_cs"][t]))
        gate = int(r["gate_cb"][t]) if (gated and r["op"][t] != 0) else -1
        new, o, rew, d = env.transition(int(r["action"][t]), int(r["step_size"][t]), gate_cb=gate)
        assert o.tobytes() == r["obs"][t].tobytes(), (name, t)
        assert rew == r["reward"][t] and d == bool(r["done"][t]), (name, t, rew, d)
        assert np.array_equal(new.grid, r["out_grid"][t].astype(np.int32)), (name, t)
        assert (new.e.cb, new.e.cs) == (r["out_cb"][t], r["out_cs"][t])
        assert (new.pos[0] == r["out_pos"][t][0]) if dim == 1 else (new.pos == tuple(r["out_pos"][t]))
        if r["op"][t] != 0:
            # what the reference does to the caller's grid: edited in place, or left alone (1D test / dynamic copy it)
            want = r["out_grid"][t] if r["aliased"][t] else r["in_grid"][t]
            assert np.array_equal(r["in_grid_after"][t], want)
            n_edit += int(r["aliased"][t])
    assert (n_edit > 0) == (not (dim == 1 and kind in ("test", "dynamic")))


def test_oracle_batch_transition_matches_single_env():
    """orc_batch_transition (the semantics of snac_transition) against orc_transition env by env, with gathers, an in-place
    subset and counter-RNG step sizes."""
    orc = helpers.oracle()
    import rng_spec

    for dim, dyn in ((1, False), (2, True), (3, True), (3, False)):
        table = helpers.plan_table(dim, dyn, ("dense_train" if dim > 1 else "sin_train") if dyn else "p0")
        n, seed = 96, 5
        b = orc.OracleBatch(dim, dyn, n, table, seed=seed)
        b.reset()
        b.rollout(17, obs=None)                                   # diverse states (some past an auto-reset)
        before, stats_before = b.state(), b.stats()
        rng = np.random.default_rng(dim)
        m = 64
        src = rng.integers(0, 32, m).astype(np.int32)             # several children per parent
        dst = (32 + np.arange(m)).astype(np.int32)
        acts = rng.integers(0, b.num_actions, m).astype(np.int8)
        obs, rew, done = b.transition(acts, None, src, dst, t=123)
        after = b.state()
        w = rng_spec.words(seed, 0, np.arange(m, dtype=np.uint64), 123)
        ks = 1 + (((w & 0xFFFF) * 3) >> 16)
        for i in range(m):
            e = orc.OracleEnv(dim, dyn)
            e.reset(table[before["plan_idx"][src[i]]].reshape(-1), int(before["plan_idx"][src[i]]))
            pos = before["pos"][src[i]]
            e.set_state(before["grid"][src[i]], int(pos[0]) if dim == 1 else pos, before["cb"][src[i]], before["cs"][src[i]])
            _, o, r, d = e.transition(int(acts[i]), int(ks[i]), inplace=True)
            assert o.tobytes() == obs[i].tobytes() and r == rew[i] and d == bool(done[i])
            assert np.array_equal(e.grid, after["grid"][dst[i]])
            assert after["ep_return"][dst[i]] == before["ep_return"][src[i]] + int(r)
            assert after["need_reset"][dst[i]] == int(d) and after["episode"][dst[i]] == before["episode"][src[i]]
        # sources are untouched, stats untouched
        for key in ("grid", "pos", "cb", "cs"):
            assert np.array_equal(after[key][:32], before[key][:32])
        assert all(np.array_equal(b.stats()[key], stats_before[key]) for key in ("episodes", "ret", "iou_fx", "steps"))
        with pytest.raises(ValueError):
            b.transition(acts, None, src, dst + n)
