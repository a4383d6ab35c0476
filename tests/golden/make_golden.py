"""Capture golden input/output vectors from the imported reference (THIS container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference holds no tests or known-answer vectors for the env path, so
parity is pinned by trajectories recorded here from the reference's own classes:

  Env/1D/DMP_Env_1D_static.py:66-151, Env/1D/DMP_Env_1D_dynamic_usedata_plan.py:40-133,
  Env/2D/DMP_Env_2D_static.py:54-154,  Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:34-147,
  Env/3D/DMP_simulator_3d_static_circle.py:67-276,
  Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py:45-277

Every case seeds numpy's global MT19937 stream (`np.random.seed(seed)`), then steps the reference env
with a recorded action stream, calling reset() after each `done`.  Recorded per step: action, the
step_size the env drew, the observation (window cells as int16 + the two scalar slots as raw float64),
reward, done, position; per episode: plan index, total_brick, reset observation, final
environment_memory and IoU.  Outputs (data only, no reference text):

  tests/golden/traj_<dim>d_<static|dynamic>.npz   trajectories
  tests/golden/static_plans.npz                   the analytic / rasterised static plans
  tests/golden/digests.json                       sha256 of 100k-step streams + MT19937 known answers
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _refimport  # noqa: E402
from rng_spec import counter_actions  # noqa: E402  (tests/rng_spec.py: numpy statement of the counter RNG)

DIMS = {1: dict(A=3, W=5), 2: dict(A=5, W=49), 3: dict(A=8, W=49)}


def make_env(classes, dim, dyn, plan, random_choose=True):
    cls = classes[(dim, dyn)]
    if dyn:
        dens, split = plan
        return cls(data_path=_refimport.dataset_path(dim, dens, split), random_choose_paln=random_choose)
    return cls(plan_choose=plan)


def primary_obs(dim, dyn, obs):
    """-> (window int16[W], scalars f64[2], raw scalars f64[2] or None)."""
    W = DIMS[dim]["W"]
    if not dyn:
        o = np.asarray(obs, dtype=np.float64).reshape(-1)
        return o[:W], o[W:W + 2], None
    if dim == 1:
        raw = np.asarray(obs[0], dtype=np.float64).reshape(-1)
        nrm = np.asarray(obs[1], dtype=np.float64).reshape(-1)
        assert np.array_equal(raw[:W], nrm[:W])
        return nrm[:W], nrm[W:W + 2], raw[W:W + 2]
    o = np.asarray(obs[0], dtype=np.float64).reshape(-1)
    return o[:W], o[W:W + 2], None


def cur_iou(dim, env):
    if dim in (1, 3):
        return float(env.iou())
    # 2D: the caller-side boolean IoU, script/DQN/2d/DQN_2d_dynamic.py:63-71 (same formula as
    # Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:153-159)
    h = env.HALF_WINDOW_SIZE
    g = env.environment_memory[h:h + env.plan_height, h:h + env.plan_width]
    p = env.plan[h:h + env.plan_height, h:h + env.plan_width]
    inter = np.logical_and(g, p)
    union = np.logical_or(g, p)
    return float(np.sum(inter) / np.sum(union))


def run_case(classes, dim, dyn, plan, seed, actions, random_choose=True, n_steps=None):
    W = DIMS[dim]["W"]
    np.random.seed(seed)
    env = make_env(classes, dim, dyn, plan, random_choose)
    policy = None
    if callable(actions):  # online policy(env) -> action; the chosen actions are recorded
        policy, actions = actions, np.zeros(n_steps, np.int8)
    S = len(actions)
    rec = dict(
        actions=np.array(actions, np.int8), step_size=np.zeros(S, np.int8), win=np.zeros((S, W), np.int16),
        sc=np.zeros((S, 2), np.float64), reward=np.zeros(S, np.float64), done=np.zeros(S, np.uint8),
        pos=np.zeros((S, 2), np.int16), cb=np.zeros(S, np.int32), cs=np.zeros(S, np.int32))
    if dim == 1 and dyn:
        rec["sc_raw"] = np.zeros((S, 2), np.float64)
    ep = dict(start=[], plan_idx=[], total_brick=[], reset_win=[], reset_sc=[], final_grid=[], iou=[], length=[])

    def do_reset(t):
        obs = env.reset()
        w, sc, _ = primary_obs(dim, dyn, obs)
        ep["start"].append(t)
        if dyn:
            ep["plan_idx"].append(env.index_random if random_choose else (env.index_for_non_random - 1) % env.plan_dataset_len)
        else:
            ep["plan_idx"].append(-1)
        tb = float(env.total_brick)
        assert tb == int(tb)
        ep["total_brick"].append(int(tb))
        ep["reset_win"].append(w.astype(np.int16))
        ep["reset_sc"].append(sc.copy())

    def close_episode(t):
        g = np.asarray(env.environment_memory, np.float64)
        assert np.array_equal(g, np.round(g))
        ep["final_grid"].append(g.astype(np.int16).reshape(-1))
        ep["iou"].append(cur_iou(dim, env))
        ep["length"].append(t + 1 - ep["start"][-1])

    do_reset(0)
    for t in range(S):
        if policy is not None:
            rec["actions"][t] = policy(env)
        obs, reward, done = env.step(int(rec["actions"][t]))
        w, sc, raw = primary_obs(dim, dyn, obs)
        assert np.array_equal(w, np.round(w))
        rec["step_size"][t] = env.step_size
        rec["win"][t] = w.astype(np.int16)
        rec["sc"][t] = sc
        if raw is not None:
            rec["sc_raw"][t] = raw
        rec["reward"][t] = float(reward)
        rec["done"][t] = 1 if done else 0
        p = env.position_memory[-1]
        rec["pos"][t] = (p, 0) if dim == 1 else (p[0], p[1])
        rec["cb"][t] = env.conut_brick if dim == 1 else env.count_brick
        rec["cs"][t] = env.count_step
        if done or t == S - 1:
            close_episode(t)
            if t != S - 1:
                do_reset(t + 1)
    out = dict(rec)
    out["seed"] = np.int64(seed)
    out["random_choose"] = np.int8(random_choose)
    out["ep_start"] = np.asarray(ep["start"], np.int32)
    out["ep_len"] = np.asarray(ep["length"], np.int32)
    out["ep_plan_idx"] = np.asarray(ep["plan_idx"], np.int32)
    out["ep_total_brick"] = np.asarray(ep["total_brick"], np.int32)
    out["ep_reset_win"] = np.stack(ep["reset_win"])
    out["ep_reset_sc"] = np.stack(ep["reset_sc"])
    out["ep_final_grid"] = np.stack(ep["final_grid"])
    out["ep_iou"] = np.asarray(ep["iou"], np.float64)
    return out


def greedy_policy(dim, rng, eps=0.15):
    """A plan-following policy (uses only public attributes of the reference env) so that the goldens
    also cover the match rewards (10 / 5), high IoU and, in 3D, towers that reach the plan height."""
    A = DIMS[dim]["A"]

    def act(env):
        if rng.random() < eps:
            return int(rng.integers(A))
        g, plan = env.environment_memory, env.plan
        if dim == 1:
            p = env.position_memory[-1]
            if g[0, p] < plan[p - 2]:
                return 2
            return int(rng.integers(2))
        r, c = env.position_memory[-1]
        if dim == 2:
            if g[r, c] == 0 and plan[r, c] == 1:
                return 4
            return int(rng.integers(4))
        nb = [(r, c - 1), (r, c + 1), (r + 1, c), (r - 1, c)]
        free = [g[n] == 0 for n in nb]
        for i, n in enumerate(nb):
            if g[n] != -1 and g[n] < plan[n]:
                left = sum(free) - (1 if free[i] else 0)
                if left >= 1:
                    return 4 + i
        return int(rng.integers(4))

    return act


def mix_actions(rng, probs, n):
    probs = np.asarray(probs, np.float64)
    return rng.choice(len(probs), size=n, p=probs / probs.sum()).astype(np.int8)


MIXES = {
    1: dict(uniform=[1, 1, 1], drop=[0.05, 0.05, 0.9], walk=[0.45, 0.45, 0.1]),
    2: dict(uniform=[1, 1, 1, 1, 1], drop=[0.1, 0.1, 0.1, 0.1, 0.6], walk=[0.24, 0.24, 0.24, 0.24, 0.04]),
    3: dict(uniform=[1] * 8, refmix=[0.2] * 4 + [0.05] * 4,  # Env/3D/DMP_simulator_3d_static_circle.py:362
            build_right=[0.03, 0, 0, 0.03, 0.02, 0.9, 0, 0.02], moves=[1, 1, 1, 1, 0, 0, 0, 0],
            sparse_build=[0.24, 0.24, 0.24, 0.24, 0.01, 0.01, 0.01, 0.01]),
}
STEPS = {1: 2400, 2: 2600, 3: 2800}


def plans_for(dim, dyn):
    if dim == 1:
        return [("sin", "train"), ("sin", "val")] if dyn else [0, 1, 2]
    if dyn:
        return [("dense", "train"), ("sparse", "train"), ("dense", "test")]
    return [0, 1]


def plan_tag(plan):
    return plan if isinstance(plan, str) else ("p%d" % plan if isinstance(plan, int) else "%s_%s" % plan)


def long_digest(classes, dim, dyn, plan, seed, n_steps):
    """sha256 over a long seed-driven stream: step sizes and plan indices come from numpy's global
    MT19937 exactly as in the reference; actions from the counter RNG (tests/rng_spec.py)."""
    A, W = DIMS[dim]["A"], DIMS[dim]["W"]
    actions = counter_actions(seed, 0, n_steps, A)
    np.random.seed(seed)
    env = make_env(classes, dim, dyn, plan, True)
    h = hashlib.sha256()
    env.reset()
    n_ep = 1
    for t in range(n_steps):
        obs, reward, done = env.step(int(actions[t]))
        w, sc, _ = primary_obs(dim, dyn, obs)
        h.update(np.concatenate([w, sc]).astype("<f8").tobytes())
        h.update(np.float32(reward).tobytes())
        h.update(b"\x01" if done else b"\x00")
        if done:
            env.reset()
            n_ep += 1
    return dict(dim=dim, dynamic=bool(dyn), plan=plan_tag(plan), seed=seed, steps=n_steps, episodes=n_ep,
                sha256=h.hexdigest())


def main():
    classes = _refimport.load_ref_classes()
    # ---- static plans (Env/1D/DMP_Env_1D_static.py:34-55, Env/2D/DMP_Env_2D_static.py:31-52,
    #      Env/3D/DMP_simulator_3d_static_circle.py:42-65)
    sp = {}
    for pc in (0, 1, 2):
        e = classes[(1, False)](plan_choose=pc)
        y, area = e.create_plan()
        assert np.array_equal(y, np.round(y))
        sp["1d_p%d" % pc] = y.astype(np.int16)
        sp["1d_p%d_tb" % pc] = np.int32(area)
    for pc in (0, 1):
        e = classes[(2, False)](plan_choose=pc)
        p, area = e.create_plan()
        sp["2d_p%d" % pc] = p.astype(np.uint8)
        sp["2d_p%d_tb" % pc] = np.int32(area)
        e = classes[(3, False)](plan_choose=pc)
        p, area = e.create_plan()
        sp["3d_p%d" % pc] = p.astype(np.uint8)
        sp["3d_p%d_tb" % pc] = np.int32(area)
    np.savez_compressed(os.path.join(HERE, "static_plans.npz"), **sp)
    print("static plans:", {k: int(v) for k, v in sp.items() if k.endswith("_tb")})

    # ---- trajectories
    for dim in (1, 2, 3):
        for dyn in (False, True):
            out = {}
            names = []
            seed = 100 * dim + (50 if dyn else 0)
            for plan in plans_for(dim, dyn):
                for mixname, probs in MIXES[dim].items():
                    seed += 1
                    rng = np.random.default_rng(7000 + seed)
                    acts = mix_actions(rng, probs, STEPS[dim])
                    name = "%s.%s" % (plan_tag(plan), mixname)
                    r = run_case(classes, dim, dyn, plan, seed, acts)
                    names.append(name)
                    for k, v in r.items():
                        out["%s/%s" % (name, k)] = v
                    print("%dd %s %-24s seed=%d episodes=%d done=%d rewards=%s" % (
                        dim, "dyn" if dyn else "sta", name, seed, len(r["ep_start"]), int(r["done"].sum()),
                        sorted(set(r["reward"].tolist()))))
            for plan in plans_for(dim, dyn)[:2]:
                seed += 1
                rng = np.random.default_rng(7000 + seed)
                name = "%s.greedy" % plan_tag(plan)
                r = run_case(classes, dim, dyn, plan, seed, greedy_policy(dim, rng), n_steps=STEPS[dim])
                names.append(name)
                for k, v in r.items():
                    out["%s/%s" % (name, k)] = v
                print("%dd %s %-24s seed=%d episodes=%d done=%d rewards=%s iou=%s" % (
                    dim, "dyn" if dyn else "sta", name, seed, len(r["ep_start"]), int(r["done"].sum()),
                    sorted(set(r["reward"].tolist())), np.round(r["ep_iou"][:4], 3)))
            if dyn:  # sequential plan order (random_choose_paln=False), one case
                plan = plans_for(dim, dyn)[-1]
                seed += 1
                rng = np.random.default_rng(7000 + seed)
                acts = mix_actions(rng, MIXES[dim]["uniform" if dim == 3 else "drop"], STEPS[dim])
                name = "%s.sequential" % plan_tag(plan)
                r = run_case(classes, dim, dyn, plan, seed, acts, random_choose=False)
                names.append(name)
                for k, v in r.items():
                    out["%s/%s" % (name, k)] = v
                print("%dd dyn %-24s seed=%d episodes=%d" % (dim, name, seed, len(r["ep_start"])))
            out["cases"] = np.array(names)
            fn = os.path.join(HERE, "traj_%dd_%s.npz" % (dim, "dynamic" if dyn else "static"))
            np.savez_compressed(fn, **out)
            print("wrote", fn, os.path.getsize(fn))

    # ---- long-stream digests + MT19937 known answers
    dig = dict(streams=[], mt19937={})
    for dim in (1, 2, 3):
        for dyn in (False, True):
            plan = plans_for(dim, dyn)[0]
            d = long_digest(classes, dim, dyn, plan, 12345, 100000)
            dig["streams"].append(d)
            print(d)
    for s in (0, 1, 12345):
        np.random.seed(s)
        a = [int(np.random.randint(1, 4)) for _ in range(40)]
        b = [int(np.random.randint(0, 400)) for _ in range(10)]
        c = np.random.randint(3, size=16).tolist()
        dig["mt19937"][str(s)] = dict(randint_1_4=a, then_randint_0_400=b, then_randint_3_size16=c)
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(dig, f, indent=1)


if __name__ == "__main__":
    main()
