"""Golden vectors for the reference's MCTS env variants (THIS container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mcts.py

  Env/1D/DMP_Env_1D_static_MCTS.py        deep_mobile_printing_1d1r_MCTS(plan_choose)            transition() edits the given grid
  Env/1D/DMP_Env_1D_static_MCTS_test.py   deep_mobile_printing_1d1r_MCTS_obs_test(plan_choose)   transition() copies
  Env/1D/DMP_Env_1D_dynamic_MCTS.py       deep_mobile_printing_1d1r_MCTS_obs(data_path, ...)     transition() copies
  Env/2D/DMP_ENV_2D_static_MCTS.py        deep_mobile_printing_2d1r_MCTS(plan_choose)            edits
  Env/2D/DMP_ENV_2D_static_MCTS_test.py   deep_mobile_printing_2d1r_MCTS_test(plan_choose)       edits
  Env/2D/DMP_ENV_2D_dynamic_MCTS.py       deep_mobile_printing_2d1r(data_path, ...)              edits
  Env/3D/DMP_simulator_3d_static_circle_MCTS.py, ..._MCTS_test.py   deep_mobile_printing_3d1r(plan_choose)   edits
  Env/3D/DMP_simulator_3d_dynamic_triangle_MCTS.py                  deep_mobile_printing_3d1r(data_path, ...)  edits; the
        brick-limit test of transition() reads the ENV's count_brick, not the state's (:258)

These classes return (state, obs) from reset() and (state, obs, reward, done) from step(), state = (position,
environment_memory copy, count_brick, count_step), and add the functional transition(state, action) used by
script/MCTS/utils/mcts_Qvalue*.py.  Each case seeds numpy's global stream and runs a recorded mix of operations:

  op 0  step(a)                       -> new node
  op 1  transition(node j, a)         -> new node (j: any earlier node of the episode)
  op 2  transition(crafted state, a)  a copy of node j with count_step / count_brick moved next to their limits
  op 3  like op 1 with env.count_brick raised to total_brick for the call (3D dynamic: the :258 quirk)

Recorded per op: the input state, the step size drawn, the output state, obs, reward, done, and the input grid as it
is AFTER the call (equal to the output grid where the class edits in place).  Output: tests/golden/traj_mcts.npz.
"""
import importlib
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _refimport  # noqa: E402
import make_golden as mg  # noqa: E402

# variant -> (dim, dynamic, module, class)
VARIANTS = {
    "1d.static": (1, False, "DMP_Env_1D_static_MCTS", "deep_mobile_printing_1d1r_MCTS"),
    "1d.test": (1, False, "DMP_Env_1D_static_MCTS_test", "deep_mobile_printing_1d1r_MCTS_obs_test"),
    "1d.dynamic": (1, True, "DMP_Env_1D_dynamic_MCTS", "deep_mobile_printing_1d1r_MCTS_obs"),
    "2d.static": (2, False, "DMP_ENV_2D_static_MCTS", "deep_mobile_printing_2d1r_MCTS"),
    "2d.test": (2, False, "DMP_ENV_2D_static_MCTS_test", "deep_mobile_printing_2d1r_MCTS_test"),
    "2d.dynamic": (2, True, "DMP_ENV_2D_dynamic_MCTS", "deep_mobile_printing_2d1r"),
    "3d.static": (3, False, "DMP_simulator_3d_static_circle_MCTS", "deep_mobile_printing_3d1r"),
    "3d.test": (3, False, "DMP_simulator_3d_static_circle_MCTS_test", "deep_mobile_printing_3d1r"),
    "3d.dynamic": (3, True, "DMP_simulator_3d_dynamic_triangle_MCTS", "deep_mobile_printing_3d1r"),
}
CASES = [
    ("1d.static", 0, "uniform"), ("1d.static", 1, "drop"), ("1d.static", 2, "uniform"), ("1d.test", 0, "drop"),
    ("1d.dynamic", ("sin", "train"), "uniform"), ("1d.dynamic", ("sin", "test"), "drop"),
    ("2d.static", 0, "uniform"), ("2d.static", 1, "drop"), ("2d.test", 0, "drop"),
    ("2d.dynamic", ("dense", "train"), "drop"), ("2d.dynamic", ("sparse", "test"), "uniform"),
    ("3d.static", 0, "refmix"), ("3d.static", 1, "uniform"), ("3d.test", 1, "sparse_build"),
    ("3d.dynamic", ("dense", "train"), "uniform"), ("3d.dynamic", ("sparse", "val"), "refmix"),
    ("3d.dynamic", ("dense", "test"), "build_right"),
]
N_OPS = 700


def unpack(dim, state):
    pos, grid, cb, cs = state
    p = (int(pos), 0) if dim == 1 else (int(pos[0]), int(pos[1]))
    return p, np.asarray(grid), int(cb), int(cs)


def run(variant, plan, mix, seed):
    dim, dyn, mod, cname = VARIANTS[variant]
    cls = getattr(importlib.import_module(mod), cname)
    cells = 34 if dim == 1 else 676
    D = 7 if dim == 1 else 51
    rng = np.random.default_rng(12000 + seed)            # choices of this script; the env draws from np.random
    acts = mg.mix_actions(rng, mg.MIXES[dim][mix], N_OPS)
    draws = []
    orig = np.random.randint

    def logged(*a, **k):
        v = orig(*a, **k)
        draws.append(v)
        return v

    np.random.randint = logged
    try:
        np.random.seed(seed)
        if dyn:
            env = cls(data_path=_refimport.dataset_path(dim, *plan), random_choose_paln=True)
        else:
            env = cls(plan_choose=plan)
        rec = dict(op=np.zeros(N_OPS, np.int8), parent=np.zeros(N_OPS, np.int32), action=acts.astype(np.int8),
                   step_size=np.zeros(N_OPS, np.int8), gate_cb=np.zeros(N_OPS, np.int32), episode=np.zeros(N_OPS, np.int32),
                   in_pos=np.zeros((N_OPS, 2), np.int16), in_cb=np.zeros(N_OPS, np.int32), in_cs=np.zeros(N_OPS, np.int32),
                   in_grid=np.zeros((N_OPS, cells), np.int16), in_grid_after=np.zeros((N_OPS, cells), np.int16),
                   out_pos=np.zeros((N_OPS, 2), np.int16), out_cb=np.zeros(N_OPS, np.int32), out_cs=np.zeros(N_OPS, np.int32),
                   out_grid=np.zeros((N_OPS, cells), np.int16), obs=np.zeros((N_OPS, D)), reward=np.zeros(N_OPS),
                   done=np.zeros(N_OPS, np.uint8), aliased=np.zeros(N_OPS, np.uint8))
        ep_plan_idx, ep_tb, ep_reset_obs, ep_plan = [], [], [], []
        nodes = []

        def reset():
            del nodes[:]
            state, obs = env.reset()
            nodes.append(state)
            ep_plan_idx.append(int(getattr(env, "index_random", 0)) if dyn else 0)
            ep_tb.append(int(env.total_brick))
            ep_reset_obs.append(np.asarray(obs, np.float64).reshape(-1))
            ep_plan.append(np.asarray(env.plan).astype(np.int16).reshape(-1))
            p, g, cb, cs = unpack(dim, state)
            assert (p, cb, cs) == ((2, 0) if dim == 1 else (3, 3), 0, 0)

        reset()
        live = True                                       # the env itself may still be stepped (no done yet)
        for t in range(N_OPS):
            a = int(acts[t])
            u = rng.random()
            op = 0 if (live and u < 0.35) else (1 if u < 0.75 else (2 if u < 0.93 else 3))
            if not live and op == 0:
                op = 1
            rec["op"][t] = op
            rec["episode"][t] = len(ep_tb) - 1
            n0 = len(draws)
            if op == 0:
                # the env's own fields: env.state may hold a grid that an in-place transition() has edited since
                p, g, cb, cs = unpack(dim, (env.position_memory[-1], env.environment_memory,
                                            env.conut_brick if dim == 1 else env.count_brick, env.count_step))
                rec["in_pos"][t], rec["in_cb"][t], rec["in_cs"][t], rec["in_grid"][t] = p, cb, cs, g.reshape(-1)
                rec["parent"][t] = -1
                rec["gate_cb"][t] = cb
                state, obs, r, d = env.step(a)
                rec["in_grid_after"][t] = rec["in_grid"][t]
            else:
                j = int(rng.integers(0, len(nodes)))
                rec["parent"][t] = j
                src = nodes[j]
                if op == 2:                               # crafted copy: next to the time limit and/or the brick limit
                    pos, grid, cb, cs = src
                    pos = list(pos) if dim != 1 else pos
                    which = int(rng.integers(0, 3))
                    if which in (0, 2):
                        cs = int(env.total_step) - 1 - int(rng.integers(0, 2))
                    if which in (1, 2):
                        cb = int(env.total_brick) - 1 - int(rng.integers(0, 2))
                    src = (pos, np.array(grid, copy=True), cb, cs)
                saved = env.count_brick if dim != 1 else env.conut_brick
                if op == 3:
                    if dim == 1:
                        env.conut_brick = int(env.total_brick)
                    else:
                        env.count_brick = int(env.total_brick)
                rec["gate_cb"][t] = int(env.count_brick if dim != 1 else env.conut_brick)
                p, g, cb, cs = unpack(dim, src)
                rec["in_pos"][t], rec["in_cb"][t], rec["in_cs"][t], rec["in_grid"][t] = p, cb, cs, g.reshape(-1)
                state, obs, r, d = env.transition(src, a)
                if dim == 1:
                    env.conut_brick = saved
                else:
                    env.count_brick = saved
                rec["in_grid_after"][t] = np.asarray(src[1]).reshape(-1)
                rec["aliased"][t] = 1 if state[1] is src[1] else 0
            assert len(draws) == n0 + 1 and 1 <= draws[-1] <= 3
            rec["step_size"][t] = draws[-1]
            p, g, cb, cs = unpack(dim, state)
            rec["out_pos"][t], rec["out_cb"][t], rec["out_cs"][t], rec["out_grid"][t] = p, cb, cs, g.reshape(-1)
            rec["obs"][t] = np.asarray(obs, np.float64).reshape(-1)
            rec["reward"][t] = float(r)
            rec["done"][t] = 1 if d else 0
            nodes.append(state)
            if op == 0 and d:
                live = False
            if (not live and rng.random() < 0.1) or len(nodes) > 120:
                reset()
                live = True
        rec.update(ep_plan_idx=np.asarray(ep_plan_idx, np.int32), ep_total_brick=np.asarray(ep_tb, np.int32),
                   ep_reset_obs=np.stack(ep_reset_obs), ep_plan=np.stack(ep_plan), seed=np.int64(seed),
                   total_step=np.int32(env.total_step))
        return rec
    finally:
        np.random.randint = orig


def main():
    _refimport.install_gym_stub()
    _refimport.install_cv2_stub()
    _refimport.load_ref_classes()                         # matplotlib backend + sys.path of Env/{1D,2D,3D}
    out, names = {}, []
    seed = 900
    for variant, plan, mix in CASES:
        seed += 1
        name = "%s.%s.%s" % (variant, plan if not isinstance(plan, tuple) else "-".join(plan), mix)
        r = run(variant, plan, mix, seed)
        names.append(name)
        for k, v in r.items():
            out["%s/%s" % (name, k)] = v
        ops = r["op"]
        print("%-34s episodes %3d ops %s done %3d rewards %s aliased %d/%d" % (
            name, len(r["ep_total_brick"]), np.bincount(ops, minlength=4).tolist(), int(r["done"].sum()),
            sorted(set(r["reward"].tolist())), int(r["aliased"].sum()), int((ops > 0).sum())))
    out["cases"] = np.array(names)
    fn = os.path.join(HERE, "traj_mcts.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
