"""GPU: tree-search entry points (snac_transition, snac_import_state, snac_obs_equal) against the CPU oracle, and the nine
MCTS drop-in classes replayed against the goldens recorded from the reference (tests/golden/make_golden_mcts.py)."""
import importlib
import os
import sys

import numpy as np
import pytest

import helpers
import test_mcts as tm

pytestmark = pytest.mark.gpu

CONFIGS = [(1, False), (1, True), (2, False), (2, True), (3, False), (3, True)]


def _tag(dim, dyn):
    return ("dense_train" if dim > 1 else "sin_train") if dyn else ("p1" if dim == 3 else "p0")


def _same_state(env, orc, rows=None):
    st = orc.state()
    rows = np.arange(orc.n) if rows is None else np.asarray(rows)
    mem = env.environment_memory().cpu().numpy().reshape(env.num_envs, -1)
    assert np.array_equal(mem[rows], st["grid"][rows].astype(np.float64))
    pos = env.position.cpu().numpy()
    assert np.array_equal(pos[rows, 0], st["pos"][rows, 0])
    if env.kind != 1:
        assert np.array_equal(pos[rows, 1], st["pos"][rows, 1])
    assert np.array_equal(env.count_brick.cpu().numpy()[rows], st["cb"][rows])
    assert np.array_equal(env.count_step.cpu().numpy()[rows], st["cs"][rows])
    assert np.array_equal(env.total_brick.cpu().numpy()[rows], st["tb"][rows])
    assert np.array_equal(env.plan_idx.cpu().numpy()[rows], st["plan_idx"][rows])
    assert np.array_equal(env.episode_return.cpu().numpy()[rows], st["ep_return"][rows])
    assert np.array_equal(env.need_reset.cpu().numpy()[rows].astype(np.uint8), st["need_reset"][rows])
    assert np.array_equal(env.episode.cpu().numpy()[rows], st["episode"][rows])


@pytest.mark.parametrize("dim,dyn", CONFIGS)
def test_transition_matches_oracle(dim, dyn):
    """A search-shaped workload: roots from a rollout, then waves of expansions (every action of a sampled parent written to
    fresh pool rows), in-place steps on a subset, explicit and counter-RNG step sizes, float32 observations too."""
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(dim, dyn, _tag(dim, dyn))
    full = table.reshape((-1, 30) if dim == 1 else (-1, 26, 26))
    pool, roots, seed = 3000, 200, 11
    env = BatchedDMPEnv(dim, dyn, pool, plans=full, seed=seed)
    orc = helpers.oracle().OracleBatch(dim, dyn, pool, table, seed=seed)
    env.reset(); orc.reset()
    env.rollout(23, obs=None); orc.rollout(23, obs=None)
    _same_state(env, orc)
    stats0 = env.episodic_stats()
    rng = np.random.default_rng(100 + dim)
    A = env.num_actions
    used = roots
    for wave in range(6):
        parents = rng.integers(0, used, 40)
        src = np.repeat(parents, A).astype(np.int32)
        acts = np.tile(np.arange(A), len(parents)).astype(np.int8)
        m = len(src)
        dst = (used + np.arange(m)).astype(np.int32)
        ks = rng.integers(1, 4, m).astype(np.int8) if wave % 2 == 0 else None
        o, r, d = env.transition(acts, ks, src, dst, t=wave)
        oo, ro, do = orc.transition(acts, ks, src, dst, t=wave)
        assert o.cpu().numpy().tobytes() == oo.tobytes(), (dim, dyn, wave)
        assert r.cpu().numpy().tobytes() == ro.tobytes() and np.array_equal(d.cpu().numpy().astype(np.uint8), do)
        used += m
        # a few in-place steps (dst == src) with a bad action mixed in
        rows = rng.choice(used, 64, replace=False).astype(np.int32)
        acts2 = rng.integers(0, A, 64).astype(np.int8)
        o, r, d = env.transition(acts2, None, rows, rows, t=1000 + wave)
        oo, ro, do = orc.transition(acts2, None, rows, rows, t=1000 + wave)
        assert o.cpu().numpy().tobytes() == oo.tobytes() and r.cpu().numpy().tobytes() == ro.tobytes()
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), do)
    assert used <= pool
    _same_state(env, orc)
    assert env.episodic_stats() == stats0                       # a search is not an episode
    # identity form (no index arrays) on the first m rows, no observation wanted
    acts = rng.integers(0, A, 500).astype(np.int8)
    o, r, d = env.transition(acts, want_obs=False)
    _, ro, do = orc.transition(acts)
    assert o is None and r.cpu().numpy().tobytes() == ro.tobytes() and np.array_equal(d.cpu().numpy().astype(np.uint8), do)
    _same_state(env, orc)
    # the pool keeps working as an env batch afterwards
    oe, re_, de = env.step(auto_reset=True)
    oo, ro, do = orc.step(env.t - 1, auto_reset=True)
    assert oe.cpu().numpy().tobytes() == oo.tobytes() and re_.cpu().numpy().tobytes() == ro.tobytes()
    with pytest.raises(ValueError):
        env.transition(acts[:4], None, [0, 1, 2, 3], [1, 7, 8, 9])    # row 1 is written by edge 0 and read by edge 1
    with pytest.raises(ValueError):
        env.transition(acts[:2], None, [0, 1], [5, 5])
    with pytest.raises(ValueError):
        env.transition(acts[:2], None, [0, pool], [5, 6])


def test_transition_f32_obs():
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(2, True, "dense_train")
    env = BatchedDMPEnv(2, True, 512, plans=table.reshape(-1, 26, 26), seed=3, obs_dtype=torch.float32)
    orc = helpers.oracle().OracleBatch(2, True, 512, table, seed=3)
    env.reset(); orc.reset()
    env.rollout(9, obs=None); orc.rollout(9, obs=None)
    src = np.arange(256, dtype=np.int32); dst = src + 256
    acts = (np.arange(256) % 5).astype(np.int8)
    o, _, _ = env.transition(acts, None, src, dst, t=2)
    oo, _, _ = orc.transition(acts, None, src, dst, t=2)
    assert o.dtype == torch.float32 and np.array_equal(o.cpu().numpy(), oo.astype(np.float32))


@pytest.mark.parametrize("dim,dyn", CONFIGS)
def test_import_states_roundtrip(dim, dyn):
    """environment_memory()/position/... of one batch -> import_states() of another (scattered rows): identical records,
    IoU (3D: the running min(height, plan) sum is rebuilt), observations and futures."""
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(dim, dyn, _tag(dim, dyn))
    full = table.reshape((-1, 30) if dim == 1 else (-1, 26, 26))
    n, seed = 777, 21
    a = BatchedDMPEnv(dim, dyn, n, plans=full, seed=seed)
    a.reset()
    a.rollout(31, obs=None)
    a.step(auto_reset=True)                                      # nobody is waiting for a reset: flags clear below
    keep = ~a.need_reset
    rows = torch.nonzero(keep).reshape(-1)
    m = int(rows.numel())
    b = BatchedDMPEnv(dim, dyn, n + 5, plans=full, seed=seed)
    perm = torch.randperm(n + 5, generator=torch.Generator().manual_seed(1))[:m].to(a.device)
    pos = a.position[rows] if dim != 1 else a.position[rows, 0]
    b.import_states(pos, a.count_brick[rows], a.count_step[rows], a.environment_memory()[rows], plan_idx=a.plan_idx[rows], dst=perm)
    assert torch.equal(b._grid[perm], a._grid[rows])
    ha, hb = a._hdr[rows].view(torch.int16), b._hdr[perm].view(torch.int16)
    assert torch.equal(ha[:, :6], hb[:, :6])                     # position, flags, cb, cs, tb, plan row
    assert torch.equal(hb[:, 6], torch.zeros_like(hb[:, 6]))     # the running return restarts
    if dim == 3:
        assert torch.equal(ha[:, 7], hb[:, 7])                   # running sum of min(height, plan)
    assert torch.equal(b.iou()[perm], a.iou()[rows])
    assert torch.equal(b.observe()[perm], a.observe()[rows])
    acts = torch.randint(0, a.num_actions, (m,), dtype=torch.int8, device=a.device)
    ks = torch.randint(1, 4, (m,), dtype=torch.int8, device=a.device)
    oa, ra, da = a.transition(acts, ks, rows, rows)
    ob, rb, db = b.transition(acts, ks, perm, perm)
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    # total_brick override and the argument checks
    b.import_states(pos[:3], [1, 2, 3], [4, 5, 6], a.environment_memory()[rows[:3]], plan_idx=[0, 0, 0], total_brick=[-32768, 77, 32767], dst=[0, 1, 2])
    assert b.total_brick[:3].tolist() == [-32768, 77, 32767] and b.count_step[:3].tolist() == [4, 5, 6]
    bad = pos[:1].clone()
    bad[...] = 0
    with pytest.raises(ValueError):
        b.import_states(bad, [0], [0], a.environment_memory()[:1], plan_idx=[0])
    with pytest.raises(ValueError):
        b.import_states(pos[:1], [0], [0], a.environment_memory()[:1], plan_idx=[b.num_plans])
    c = BatchedDMPEnv(dim, dyn, 4, plans=full)
    with pytest.raises(Exception):
        c.import_states(pos[:1], [0], [0], a.environment_memory()[:1])        # no plan row known yet


def test_import_states_matches_oracle_set_state():
    """Hand-made states (heights above the plan, counts next to their limits) through import_states + transition vs the oracle."""
    import torch
    from snac_amd import BatchedDMPEnv

    orc_mod = helpers.oracle()
    for dim, dyn in CONFIGS:
        table = helpers.plan_table(dim, dyn, _tag(dim, dyn))
        full = table.reshape((-1, 30) if dim == 1 else (-1, 26, 26))
        n = 300
        rng = np.random.default_rng(7 * dim + dyn)
        env = BatchedDMPEnv(dim, dyn, n, plans=full, seed=1)
        orc = orc_mod.OracleBatch(dim, dyn, n, table, seed=1)
        H, W = (1, 34) if dim == 1 else (26, 26)
        hw = 2 if dim == 1 else 3
        mem = -np.ones((n, H, W))
        hi = {1: 40, 2: 2, 3: 9}[dim]
        if dim == 1:
            mem[:, :, hw:W - hw] = rng.integers(0, hi, (n, 1, 30)) * (rng.random((n, 1, 30)) < 0.6)
        else:
            mem[:, hw:H - hw, hw:W - hw] = rng.integers(0, hi, (n, 20, 20)) * (rng.random((n, 20, 20)) < 0.3)
        pos = rng.integers(hw, hw + (30 if dim == 1 else 20), (n, 2))
        if dim == 3:                                               # the agent stands on an empty cell
            mem[np.arange(n), pos[:, 0], pos[:, 1]] = 0
        pidx = rng.integers(0, len(table), n)
        T = env.total_step
        cs = np.where(rng.random(n) < 0.3, T - 1 - rng.integers(0, 2, n), rng.integers(0, T - 2, n))
        cb = rng.integers(0, 60, n)
        env.import_states(pos if dim != 1 else pos[:, 0], cb, cs, mem, plan_idx=pidx)
        for i in range(n):
            orc.set_state(i, mem[i].astype(np.int32), int(pos[i, 0]) if dim == 1 else pos[i], int(cb[i]), int(cs[i]), plan_idx=int(pidx[i]))
        _same_state(env, orc)
        assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()
        acts = rng.integers(0, env.num_actions, n).astype(np.int8)
        ks = rng.integers(1, 4, n).astype(np.int8)
        o, r, d = env.transition(acts, ks)
        oo, ro, do = orc.transition(acts, ks)
        assert o.cpu().numpy().tobytes() == oo.tobytes() and r.cpu().numpy().tobytes() == ro.tobytes()
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), do)
        _same_state(env, orc)
        assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()


def test_obs_equal():
    import torch
    from snac_amd import BatchedDMPEnv

    for dim, dt in ((1, torch.float64), (2, torch.float64), (3, torch.float32)):
        env = BatchedDMPEnv(dim, False, 8, obs_dtype=dt)
        D = env.obs_dim
        g = torch.Generator().manual_seed(dim)
        a = torch.randint(-1, 3, (1000, D), generator=g).to(dt).cuda()
        b = a[torch.randint(0, 1000, (700,), generator=g)].clone()
        ia = torch.randint(0, 1000, (5000,), generator=g).cuda()
        ib = torch.randint(0, 700, (5000,), generator=g).cuda()
        b[::7, D - 1] += 1                                        # differ in the last slot only
        b[3, 0] = float("nan")                                    # np.array_equal: nan != nan
        want = (a[ia] == b[ib]).all(dim=1)
        assert torch.equal(env.obs_equal(a, b, ia, ib), want) and bool(want.any()) and not bool(want.all())
        assert torch.equal(env.obs_equal(a[:700], b), (a[:700] == b).all(dim=1))
        assert bool(env.obs_equal(b, b)[3]) is False
        with pytest.raises(ValueError):
            env.obs_equal(a, b, ia + 1000, ib)


# ---- the nine drop-in classes against the reference's goldens -----------------------------------------------------------
MODS = {
    "1d.static": ("1D", "DMP_Env_1D_static_MCTS", "deep_mobile_printing_1d1r_MCTS"),
    "1d.test": ("1D", "DMP_Env_1D_static_MCTS_test", "deep_mobile_printing_1d1r_MCTS_obs_test"),
    "1d.dynamic": ("1D", "DMP_Env_1D_dynamic_MCTS", "deep_mobile_printing_1d1r_MCTS_obs"),
    "2d.static": ("2D", "DMP_ENV_2D_static_MCTS", "deep_mobile_printing_2d1r_MCTS"),
    "2d.test": ("2D", "DMP_ENV_2D_static_MCTS_test", "deep_mobile_printing_2d1r_MCTS_test"),
    "2d.dynamic": ("2D", "DMP_ENV_2D_dynamic_MCTS", "deep_mobile_printing_2d1r"),
    "3d.static": ("3D", "DMP_simulator_3d_static_circle_MCTS", "deep_mobile_printing_3d1r"),
    "3d.test": ("3D", "DMP_simulator_3d_static_circle_MCTS_test", "deep_mobile_printing_3d1r"),
    "3d.dynamic": ("3D", "DMP_simulator_3d_dynamic_triangle_MCTS", "deep_mobile_printing_3d1r"),
}


def _load(variant):
    d, mod, cls = MODS[variant]
    path = os.path.join(helpers.ROOT, "snac_amd", "Env", d)
    sys.path.insert(0, path)
    try:
        sys.modules.pop(mod, None)
        return getattr(importlib.import_module(mod), cls)
    finally:
        sys.path.remove(path)


@pytest.mark.parametrize("name", tm.names())
def test_facade_replays_mcts_goldens(name):
    """From np.random.seed alone: the same reset / step / transition sequence as the capture script ran on the reference."""
    r = tm.rec(name)
    dim, dyn, kind = tm.variant(name)
    variant = name.split(".")[0] + "." + kind
    cls = _load(variant)
    plan = name.split(".")[2]
    np.random.seed(int(r["seed"]))
    if dyn:
        dens, split = plan.split("-")
        pre = "data_1d_dynamic_sin_envplan_500_" if dim == 1 else "data_%dd_dynamic_%s_envplan_500_" % (dim, dens)
        env = cls(data_path="/nonexistent/" + pre + split + ".pkl", random_choose_paln=True)
    else:
        env = cls(plan_choose=int(plan))
    assert env.action_space.n == {1: 3, 2: 5, 3: 8}[dim]
    shape = (1, 34) if dim == 1 else (26, 26)
    nodes, episode = [], -1

    def unpack(state):
        pos, grid, cb, cs = state
        p = (int(pos), 0) if dim == 1 else (int(pos[0]), int(pos[1]))
        return p, np.asarray(grid), int(cb), int(cs)

    for t in range(len(r["op"])):
        if int(r["episode"][t]) != episode:
            episode += 1
            state, obs = env.reset()
            assert np.asarray(obs).tobytes() == r["ep_reset_obs"][episode].reshape(1, -1).tobytes()
            assert int(env.total_brick) == r["ep_total_brick"][episode]
            assert np.array_equal(np.asarray(env.plan).reshape(-1), r["ep_plan"][episode])
            nodes = [state]
        op, a = int(r["op"][t]), int(r["action"][t])
        if op == 0:
            state, obs, rew, d = env.step(a)
            assert env.step_size == r["step_size"][t]
        else:
            src = nodes[int(r["parent"][t])]
            if op == 2:
                pos, grid, _, _ = src
                src = (list(pos) if dim != 1 else pos, np.array(grid, copy=True), int(r["in_cb"][t]), int(r["in_cs"][t]))
            p, g, cb, cs = unpack(src)
            assert (p, cb, cs) == (tuple(r["in_pos"][t]), r["in_cb"][t], r["in_cs"][t]), (name, t)
            assert np.array_equal(g.reshape(-1), r["in_grid"][t]), (name, t)      # includes every earlier in-place edit
            saved = env.conut_brick if dim == 1 else env.count_brick
            if op == 3:
                if dim == 1:
                    env.conut_brick = int(env.total_brick)
                else:
                    env.count_brick = int(env.total_brick)
            state, obs, rew, d = env.transition(src, a)
            if dim == 1:
                env.conut_brick = saved
            else:
                env.count_brick = saved
            assert np.array_equal(np.asarray(src[1]).reshape(-1), r["in_grid_after"][t]), (name, t)
            assert (state[1] is src[1]) == bool(r["aliased"][t])
        p, g, cb, cs = unpack(state)
        assert g.shape == shape and g.dtype == np.float64
        assert (p, cb, cs) == (tuple(r["out_pos"][t]), r["out_cb"][t], r["out_cs"][t]), (name, t)
        assert np.array_equal(g.reshape(-1), r["out_grid"][t]), (name, t)
        assert np.asarray(obs).shape == (1, 7 if dim == 1 else 51)
        assert np.asarray(obs, np.float64).tobytes() == r["obs"][t].reshape(1, -1).tobytes(), (name, t)
        assert rew == r["reward"][t] and d == bool(r["done"][t]), (name, t)
        assert env.equality_operator(obs, r["obs"][t].reshape(1, -1)) and not env.equality_operator(obs, np.asarray(obs) + 1)
        nodes.append(state)
    if variant == "1d.static":
        assert env.iou_MCTS(nodes[-1][1]) == _iou1(env, nodes[-1][1]) and not hasattr(_load("1d.test"), "iou_MCTS")


def _iou1(env, mem):
    g = np.asarray(mem)[0][2:32]
    p = np.asarray(env.plan)
    cross = g.sum() - np.maximum(g - p, 0).sum()
    return float(cross / (p.sum() + g.sum() - cross))


@pytest.mark.parametrize("dim,dyn", [(1, False), (2, True), (3, True), (3, False)])
def test_default_policy_evaluation_matches_the_reference_loop(dim, dyn):
    """BatchedDMPEnv.evaluate against a restatement of script/MCTS/utils/mcts.py:100-110 on the oracle: random steps until
    done or the horizon, estimate += reward * gamma**t in python floats -- bit-equal, terminal leaves untouched."""
    import rng_spec
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(dim, dyn, _tag(dim, dyn))
    full = table.reshape((-1, 30) if dim == 1 else (-1, 26, 26))
    n, seed, H, gamma = 400, 13, 60 if dim != 3 else 40, 0.9
    env = BatchedDMPEnv(dim, dyn, n, plans=full, seed=seed)
    orc_mod = helpers.oracle()
    orc = orc_mod.OracleBatch(dim, dyn, n, table, seed=seed)
    env.reset(); orc.reset()
    T0 = {1: 37, 2: 600, 3: 21}[dim]                              # 2D: everybody still running hits the time limit at tick 600
    env.rollout(T0, obs=None); orc.rollout(T0, obs=None)          # some rows end on a terminal step
    rng = np.random.default_rng(dim)
    rows = rng.integers(0, n, 150)
    first = rng.integers(-1, 11, 150).astype(np.float64)
    before = (env._hdr.clone(), env._grid.clone())
    est, steps = env.evaluate(rows, H, gamma, first_reward=first)
    assert torch.equal(env._hdr, before[0]) and torch.equal(env._grid, before[1])
    st = orc.state()
    A = env.num_actions
    n_term = 0
    for i, row in enumerate(rows):
        e = orc_mod.OracleEnv(dim, dyn)
        e.reset(table[st["plan_idx"][row]].reshape(-1), int(st["plan_idx"][row]))
        pos = st["pos"][row]
        e.set_state(st["grid"][row], int(pos[0]) if dim == 1 else pos, st["cb"][row], st["cs"][row])
        estimate, terminal, t = float(first[i]), bool(st["need_reset"][row]), 0
        n_term += terminal
        while (not terminal) and t < H:
            w = rng_spec.words(seed, 0, np.uint64(i), np.uint64(t))
            a, k = int(rng_spec.action_of(w, A)), int(rng_spec.step_size_of(w))
            _, _, r, terminal = e.transition(a, k, inplace=True)
            estimate += r * (gamma ** t)
            t += 1
        assert np.float64(est[i].item()).tobytes() == np.float64(estimate).tobytes(), (dim, dyn, i)
        assert int(steps[i]) == t
    assert n_term > 0 or dim == 1


@pytest.mark.gpu
@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_large_waves_of_2d_edges(dyn, f32):
    """A wave of 65 536 + 36 edges on a pool of 2^18 rows.  Round 5: gathered rows take k_edges2d -- every source record fetched once by
    five neighbouring lanes (16 bytes each), through LDS, out to its destination row the same way, rows through emit_tile: 0.075 against
    k_transition2d's 0.105 ms per 524 288 random-parent edges (round 3's 64-edge tiles with emit_tile but per-lane 4-byte loads were 6 %
    SLOWER than k_transition2d: what counts is the number of scattered lane requests) -- when m % 4 == 0 and the observations are 16-byte
    aligned.  Shared random parents from the lower half, distinct children in the upper half, some edges in place; against the oracle,
    and a second wave on the children; and m % 4 != 0 (k_transition2d) gives the same rows for the same edges."""
    import torch
    from snac_amd import BatchedDMPEnv, _lib

    pool, m = 1 << 18, 65536 + 36
    table = helpers.plan_table(2, dyn, "dense_train" if dyn else "p0")
    env = BatchedDMPEnv(2, dyn, pool, plans=table.reshape(len(table), 26, 26), seed=21, obs_dtype=torch.float32 if f32 else torch.float64)
    orc = helpers.oracle().OracleBatch(2, dyn, pool, table, seed=21)
    env.reset(); orc.reset()
    env.rollout(25, obs=None); orc.rollout(25, obs=None, nthreads=16)
    rng = np.random.default_rng(5)
    for wave in range(2):
        dst = (pool // 2 + rng.choice(pool // 2, m, replace=False)).astype(np.int32)
        src = rng.integers(0, pool // 2, m).astype(np.int32)
        inplace = rng.random(m) < 0.1
        src = np.where(inplace, dst, src).astype(np.int32)
        acts = rng.integers(0, 5, m).astype(np.int8)
        ks = rng.integers(1, 4, m).astype(np.int8) if wave == 0 else None
        o, r, d = env.transition(acts, ks, src, dst, t=wave)
        assert _lib.lib().snac_last_kernel() == b"k_edges2d"
        oo, ro, do = orc.transition(acts, ks, src, dst, t=wave)
        assert o.cpu().numpy().tobytes() == (oo.astype(np.float32) if f32 else oo).tobytes(), wave
        assert r.cpu().numpy().tobytes() == ro.tobytes() and np.array_equal(d.cpu().numpy().astype(np.uint8), do), wave
    assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()
    idx = np.arange(0, pool, 1031)
    cb = env.count_brick.cpu().numpy()
    assert [int(cb[i]) for i in idx] == [int(orc.b.contents.envs[int(i)].cb) for i in idx]
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(pool, -1), st["grid"])
    # the other kernel on the same edges: m - 2 of them (m % 4 != 0), rows and records compared on the device
    twin = env.fork(torch.arange(pool, device=env.device))
    dst = (pool // 2 + rng.choice(pool // 2, m, replace=False)).astype(np.int32)
    src = rng.integers(0, pool // 2, m).astype(np.int32)
    acts = rng.integers(0, 5, m).astype(np.int8)
    o1, r1, d1 = env.transition(acts, None, src, dst, t=7)
    o2, r2, d2 = twin.transition(acts[: m - 2], None, src[: m - 2], dst[: m - 2], t=7)
    assert _lib.lib().snac_last_kernel() == b"k_transition2d"
    assert torch.equal(o1[: m - 2], o2) and torch.equal(r1[: m - 2], r2) and torch.equal(d1[: m - 2], d2)
    keep = torch.from_numpy(dst[: m - 2].astype(np.int64)).to(env.device)
    assert torch.equal(env._grid[keep], twin._grid[keep]) and torch.equal(env._hdr[keep], twin._hdr[keep]) and torch.equal(env._episode[keep], twin._episode[keep])


@pytest.mark.gpu
@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_large_waves_of_3d_edges_through_lds(dyn, f32):
    """Round 4: 3D tree edges with gathered rows take k_edges3d (the wave's 32 source records through LDS, 16-byte pieces in and out,
    the rows through emit_tile) when m % 4 == 0 and the observations are 16-byte aligned.  A wave of 32 768 + 36 edges on a pool of
    2^17 rows -- shared random parents, distinct children, a tenth of the edges in place, a ragged last group of 4 -- against the oracle,
    a second wave on the children; and m % 4 != 0 (k_transition3d) gives the same rows for the same edges."""
    import torch
    from snac_amd import BatchedDMPEnv, _lib

    pool, m = 1 << 17, 32768 + 36
    table = helpers.plan_table(3, dyn, "dense_train" if dyn else "p1")
    env = BatchedDMPEnv(3, dyn, pool, plans=table.reshape(len(table), 26, 26), seed=22, obs_dtype=torch.float32 if f32 else torch.float64)
    orc = helpers.oracle().OracleBatch(3, dyn, pool, table, seed=22)
    env.reset(); orc.reset()
    env.rollout(25, obs=None); orc.rollout(25, obs=None, nthreads=16)
    rng = np.random.default_rng(6)
    for wave in range(2):
        dst = (pool // 2 + rng.choice(pool // 2, m, replace=False)).astype(np.int32)
        src = rng.integers(0, pool // 2, m).astype(np.int32)
        inplace = rng.random(m) < 0.1
        src = np.where(inplace, dst, src).astype(np.int32)
        acts = rng.integers(0, 8, m).astype(np.int8)
        ks = rng.integers(1, 4, m).astype(np.int8) if wave == 0 else None
        o, r, d = env.transition(acts, ks, src, dst, t=wave)
        assert _lib.lib().snac_last_kernel() == b"k_edges3d"
        oo, ro, do = orc.transition(acts, ks, src, dst, t=wave)
        assert o.cpu().numpy().tobytes() == (oo.astype(np.float32) if f32 else oo).tobytes(), wave
        assert r.cpu().numpy().tobytes() == ro.tobytes() and np.array_equal(d.cpu().numpy().astype(np.uint8), do), wave
    assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(pool, -1), st["grid"])
    # the other kernel on the same edges: m - 2 of them (m % 4 != 0), rows compared on the device
    twin = env.fork(torch.arange(pool, device=env.device))
    dst = (pool // 2 + rng.choice(pool // 2, m, replace=False)).astype(np.int32)
    src = rng.integers(0, pool // 2, m).astype(np.int32)
    acts = rng.integers(0, 8, m).astype(np.int8)
    o1, r1, d1 = env.transition(acts, None, src, dst, t=7)
    o2, r2, d2 = twin.transition(acts[: m - 2], None, src[: m - 2], dst[: m - 2], t=7)
    assert _lib.lib().snac_last_kernel() == b"k_transition3d"
    assert torch.equal(o1[: m - 2], o2) and torch.equal(r1[: m - 2], r2) and torch.equal(d1[: m - 2], d2)


@pytest.mark.gpu
def test_discounted_return_equals_the_python_loop_bit_for_bit():
    """snac_discounted_return (round 6: the sums of BatchedDMPEnv.evaluate on the device) against the reference's loop in python floats
    (script/MCTS/utils/mcts.py:100-110: `estimate += reward * (gamma**t)` while not terminal) on random reward / done arrays: rewards that
    do not sum exactly (-1, 5, 10, -100 times powers of 0.9 / 0.99: two roundings per step, no fused multiply-add), terminal leaves, leaves
    that never end, a horizon of zero, steps = NULL."""
    import ctypes as C

    import torch
    from snac_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(4)
    for H, m, gamma in ((37, 300, 0.9), (600, 1000, 0.99), (1, 5, 0.5), (0, 7, 0.9)):
        reward = rng.choice(np.array([-100.0, -1.0, 0.0, 1.0, 5.0, 10.0], np.float32), size=(H, m))
        done = (rng.random((H, m)) < 0.02)
        terminal = rng.random(m) < 0.1
        first = rng.integers(-1, 11, m).astype(np.float64)
        want, steps_want = first.copy(), np.zeros(m, np.int64)
        for i in range(m):
            estimate, term, t = float(first[i]), bool(terminal[i]), 0
            while (not term) and t < H:
                estimate += float(reward[t, i]) * (gamma ** t)
                term = bool(done[t, i])
                t += 1
            want[i], steps_want[i] = estimate, t
        r = torch.from_numpy(reward).cuda().contiguous() if H else torch.empty((0, m), dtype=torch.float32, device="cuda")
        d = torch.from_numpy(done.astype(np.uint8)).cuda().contiguous() if H else torch.empty((0, m), dtype=torch.uint8, device="cuda")
        tm = torch.from_numpy(terminal.astype(np.uint8)).cuda()
        gp = torch.tensor([gamma ** t for t in range(H)], dtype=torch.float64).cuda()
        est = torch.from_numpy(first).cuda()
        steps = torch.empty(m, dtype=torch.int64, device="cuda")
        vp = lambda t: C.c_void_p(t.data_ptr()) if t.numel() else None  # noqa: E731
        _lib.check(L.snac_discounted_return(H, m, vp(r), vp(d), vp(tm), vp(gp), vp(est), vp(steps), None))
        torch.cuda.synchronize()
        assert est.cpu().numpy().tobytes() == want.tobytes(), (H, m)
        assert np.array_equal(steps.cpu().numpy(), steps_want)
        est2 = torch.from_numpy(first).cuda()
        _lib.check(L.snac_discounted_return(H, m, vp(r), vp(d), None, vp(gp), vp(est2), None, None))   # no terminal leaves, no step counts
        torch.cuda.synchronize()
        want2 = first.copy()
        for i in range(m):
            estimate, term, t = float(first[i]), False, 0
            while (not term) and t < H:
                estimate += float(reward[t, i]) * (gamma ** t)
                term = bool(done[t, i])
                t += 1
            want2[i] = estimate
        assert est2.cpu().numpy().tobytes() == want2.tobytes()
    assert L.snac_discounted_return(-1, 4, None, None, None, None, None, None, None) != 0
    assert L.snac_discounted_return(3, 4, None, None, None, None, None, None, None) != 0
