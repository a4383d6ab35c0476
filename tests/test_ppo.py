"""The env copies under script/PPO of the reference (flat observations, 4-tuple step, `>` termination tests): goldens recorded
by tests/golden/make_golden_ppo.py replayed through the CPU oracle with the rule switches (CPU test), through the drop-in
classes on the HIP path from np.random.seed alone, and the rule bits of the batched API against the oracle (GPU tests)."""
import importlib.util
import os

import numpy as np
import pytest

import helpers

_Z = {}
RULES = {"1d_static": (0, 0), "1d_dynamic": (1, 0), "2d_static": (1, 0), "2d_dynamic": (0, 0), "3d_static": (1, 1), "3d_dynamic": (0, 0)}
FILES = {"ppo": {"1d_static": "DMP_Env_1D_static", "1d_dynamic": "DMP_Env_1D_dynamic_usedata_plan", "2d_static": "DMP_Env_2D_static",
                 "2d_dynamic": "DMP_Env_2d_dynamic_usedata_plan", "3d_static": "DMP_simulator_3d_static_circle",
                 "3d_dynamic": "DMP_simulator_3d_dynamic_triangle_usedata"},
         "sac": {"1d_static": "DMP_Env_1D_static", "1d_dynamic": "DMP_Env_1D_dynamic", "2d_static": "DMP_Env_2D_static",
                 "2d_dynamic": "DMP_Env_2D_dynamic", "3d_static": "DMP_simulator_3d_static_circle",
                 "3d_dynamic": "DMP_simulator_3d_dynamic_triangle_usedata"}}


def _file(suite):
    if suite not in _Z:
        _Z[suite] = np.load(os.path.join(helpers.GOLDEN, "traj_%s.npz" % suite))
    return _Z[suite]


def _names():
    """'ppo:<case>' / 'sac:<case>' (script/PPO and script/SAC/environments copies; tests/golden/make_golden_ppo.py)."""
    return ["%s:%s" % (suite, n) for suite in ("ppo", "sac") for n in _file(suite)["cases"].tolist()]


def _rec(name):
    suite, case = name.split(":")
    z = _file(suite)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(case + "/")}


def _replay(name, reset, step, state):
    rec = _rec(name)
    starts = dict((int(s), e) for e, s in enumerate(rec["ep_start"]))
    S = len(rec["actions"])
    for t in range(S):
        if t in starts:
            e = starts[t]
            o = np.asarray(reset(e, rec), np.float64).reshape(-1)
            want = np.concatenate([rec["ep_reset_win"][e].astype(np.float64), rec["ep_reset_sc"][e]])
            assert o[:len(want)].tobytes() == want.tobytes()
        o, r, d = step(int(rec["actions"][t]), int(rec["step_size"][t]), t, rec)
        o = np.asarray(o, np.float64).reshape(-1)
        want = np.concatenate([rec["win"][t].astype(np.float64), rec["sc"][t]])
        assert o[:len(want)].tobytes() == want.tobytes(), (name, t)
        assert r == rec["reward"][t] and bool(d) == bool(rec["done"][t]), (name, t)
        if (t + 1) in starts or t == S - 1:
            e = starts[t + 1] - 1 if (t + 1) in starts else len(rec["ep_start"]) - 1
            grid, iou = state()
            assert np.array_equal(np.asarray(grid).reshape(-1), rec["ep_final_grid"][e].astype(np.float64)), (name, e)
            assert np.float64(iou).tobytes() == np.float64(rec["ep_iou"][e]).tobytes()


@pytest.mark.parametrize("name", _names())
def test_oracle_replays_ppo_goldens(name):
    orc = helpers.oracle()
    fork = name.split(":")[1].split(".")[0]
    dim, dyn = int(fork[0]), fork.endswith("dynamic")
    env = orc.OracleEnv(dim, dyn).configure(obs_norm=0, rules_dyn=int(dyn)).set_rules(*RULES[fork])

    def reset(e, rec):
        o = env.reset(rec["ep_plan"][e].astype(np.int32), int(rec["ep_plan_idx"][e]))
        assert env.e.tb == rec["ep_total_brick"][e]
        return o

    _replay(name, reset, lambda a, k, t, rec: env.step(a, k), lambda: (env.grid.astype(np.float64), env.iou()))


def test_strict_rules_change_exactly_the_boundary_step():
    """brick_gt / time_gt move the terminal step by one and nothing else (oracle, all kinds)."""
    orc = helpers.oracle()
    for dim in (1, 2, 3):
        plan = orc.static_plan(dim, 0)
        drop = {1: 2, 2: 4, 3: 5}[dim]
        for bg, tg in ((0, 0), (1, 0), (0, 1), (1, 1)):
            env = orc.OracleEnv(dim, False).set_rules(bg, tg)
            env.reset(plan)
            n = 0
            while True:
                _, _, d = env.step(drop, 1)
                n += 1
                if d:
                    break
            assert n == min(env.e.tb + bg, env.e.total_step + tg), (dim, bg, tg, n)
            env.reset(plan)
            n = 0
            while True:
                _, _, d = env.step(0 if n % 2 else 1, 1)      # walk right / left: never blocked, never builds
                n += 1
                if d:
                    break
            assert n == env.e.total_step + tg


# ---- GPU ---------------------------------------------------------------------------------------------------------------
def _load(suite, fork):
    sub = os.path.join("PPO", fork) if suite == "ppo" else os.path.join("SAC", "environments")
    path = os.path.join(helpers.ROOT, "snac_amd", "script", sub, FILES[suite][fork] + ".py")
    spec = importlib.util.spec_from_file_location("%s_shim_%s" % (suite, fork), path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return getattr(mod, "deep_mobile_printing_%sd1r" % fork[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names())
def test_ppo_facades_on_hip(name):
    suite, case = name.split(":")
    fork, plan = case.split(".")[0], case.split(".")[1]
    dim, dyn = int(fork[0]), fork.endswith("dynamic")
    cls = _load(suite, fork)
    rec0 = _rec(name)
    np.random.seed(int(rec0["seed"]))
    if dyn:
        dens, split = plan.split("-")
        pre = "data_1d_dynamic_sin_envplan_500_" if dim == 1 else "data_%dd_dynamic_%s_envplan_500_" % (dim, dens)
        env = cls(data_path="/nonexistent/" + pre + split + ".pkl", random_choose_paln=True)
    else:
        env = cls(plan_choose=int(plan))
    W = helpers.DIMS[dim]["W"]
    D = W + 2 + ((30 if dim == 1 else 400) if dyn else 0)
    if suite == "ppo" or dim == 3:
        assert env.action_space.n == helpers.DIMS[dim]["A"] and env.observation_space.shape == (D,)
    else:
        assert not hasattr(env, "action_space") and not hasattr(env, "observation_space")

    def tail_ok(o, e, rec):
        if dyn:
            p = rec["ep_plan"][e].astype(np.float64)
            p = p if dim == 1 else p.reshape(26, 26)[3:23, 3:23].reshape(-1)
            assert np.array_equal(o[W + 2:], p)

    cur = {"e": -1}

    def reset(e, rec):
        o = env.reset()
        cur["e"] = e
        assert o.shape == (D,) and o.dtype == np.float64 and int(env.total_brick) == rec["ep_total_brick"][e]
        tail_ok(o, e, rec)
        return o

    def step(a, k, t, rec):
        ret = env.step(a)
        assert len(ret) == (4 if suite == "ppo" else 3) and (suite != "ppo" or ret[3] == {})
        o, r, d = ret[:3]
        assert o.shape == (D,) and env.step_size == k
        tail_ok(o, cur["e"], rec)
        return o, r, d

    def iou():
        if dim != 2:
            return env.iou()
        g = env.environment_memory[3:23, 3:23]
        p = env.plan[3:23, 3:23]
        return float(np.sum(np.logical_and(g, p)) / np.sum(np.logical_or(g, p)))

    _replay(name, reset, step, lambda: (env.environment_memory, iou()))
    if fork == "2d_static":
        assert env.conut_brick == env.count_brick              # the fork's spelling (:15)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,dyn,bg,tg", [(1, True, 1, 0), (2, False, 1, 0), (2, True, 1, 1), (3, False, 1, 1), (3, True, 0, 1), (1, False, 1, 1)])
def test_rule_bits_batched_vs_oracle(dim, dyn, bg, tg):
    """SNAC_RULE_BRICK_GT / SNAC_RULE_TIME_GT in the fused rollout (auto-reset, episodic sums) against the oracle, with a
    short time limit so that both boundaries are hit by many envs."""
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(dim, dyn, ("dense_train" if dim > 1 else "sin_train") if dyn else "p0")
    full = table.reshape((-1, 30) if dim == 1 else (-1, 26, 26))
    N, T, limit = 700, 260, 97
    env = BatchedDMPEnv(dim, dyn, N, plans=full, seed=9, brick_gt=bool(bg), time_gt=bool(tg), total_step=limit)
    orc = helpers.oracle().OracleBatch(dim, dyn, N, table, seed=9).set_rules(bg, tg).set_total_step(limit)
    env.reset(); orc.reset()
    rng = np.random.default_rng(dim * 10 + bg * 2 + tg)
    A = env.num_actions
    drop = {1: 2, 2: 4, 3: 5}[dim]
    acts = np.where(rng.random((T, N)) < 0.7, drop, rng.integers(0, A, (T, N))).astype(np.int8)
    o, r, d = env.rollout(T, actions=acts)
    oo, ro, do = orc.rollout(T, actions=acts)
    assert o.cpu().numpy().tobytes() == oo.tobytes()
    assert r.cpu().numpy().tobytes() == ro.tobytes() and np.array_equal(d.cpu().numpy().astype(np.uint8), do)
    s = orc.stats()
    assert env.episodic_stats() == dict(episodes=int(s["episodes"].sum()), return_sum=int(s["ret"].sum()), iou_fx_sum=int(s["iou_fx"].sum()))
    assert int(s["episodes"].sum()) > 0
    # the same inputs without the bits give a different trajectory (the bits are live)
    ref = BatchedDMPEnv(dim, dyn, N, plans=full, seed=9, total_step=limit)
    ref.reset()
    _, _, d0 = ref.rollout(T, actions=acts)
    if tg or (dim == 2 and dyn):                                 # other plans need > 97 bricks: only the time bit can fire there
        assert not torch.equal(d0, d)
