"""Golden vectors for the hindsight relabel flow of the reference's DRQN_hindsight scripts (THIS container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_relabel.py

Per case: one episode on the canonical static env (seeded, plan-following policy with exploration), its recorded
actions and drawn step sizes, its final environment_memory; then exactly what
script/DRQN_hindsight/2d/DRQN_hindsight_2D_static.py:245-252 (1d: :242-245, 3d: :239-246) does with the hindsight env:
reset(), overwrite the plan with the final grid, replay step(action, step_size) and keep the rewards.
Output: tests/golden/relabel_static.npz.
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

HMODS = {1: "DMP_Env_1D_static_hindsight_replay", 2: "DMP_Env_2D_static_hindsight_replay",
         3: "DMP_simulator_3d_static_circle_hindsight_replay"}


def main():
    classes = _refimport.load_ref_classes()
    out, names = {}, []
    seed = 700
    for dim in (1, 2, 3):
        hcls = getattr(importlib.import_module(HMODS[dim]), "deep_mobile_printing_%dd1r_hindsight" % dim)
        for pc in ((0, 2) if dim == 1 else (0, 1)):
            for rep in range(3):
                seed += 1
                np.random.seed(seed)
                rng = np.random.default_rng(9000 + seed)
                env = classes[(dim, False)](plan_choose=pc)
                policy = mg.greedy_policy(dim, rng, eps=0.3 if dim != 3 else 0.1)
                env.reset()
                acts, ks, rews = [], [], []
                while True:
                    a = policy(env)
                    _, r, d = env.step(a)
                    acts.append(a); ks.append(env.step_size); rews.append(float(r))
                    if d:
                        break
                h = env.HALF_WINDOW_SIZE
                henv = hcls(plan_choose=pc)
                henv.reset()
                if dim == 1:
                    henv.plan = env.environment_memory[0, h:h + env.plan_width]
                else:
                    henv.plan[h:h + env.plan_height, h:h + env.plan_width] = env.environment_memory[h:h + env.plan_height, h:h + env.plan_width]
                    henv.input_plan = henv.plan[h:h + env.plan_height, h:h + env.plan_width]
                hr, hd = [], []
                for a, k in zip(acts, ks):
                    _, r, d = henv.step(a, k)
                    hr.append(float(r)); hd.append(1 if d else 0)
                name = "%dd.p%d.%d" % (dim, pc, rep)
                names.append(name)
                rec = dict(actions=np.asarray(acts, np.int8), step_size=np.asarray(ks, np.int8), reward=np.asarray(rews),
                           final_grid=np.asarray(env.environment_memory, np.float64).astype(np.int16).reshape(-1),
                           total_brick=np.int32(env.total_brick), hindsight_reward=np.asarray(hr), hindsight_done=np.asarray(hd, np.uint8),
                           seed=np.int64(seed))
                for k2, v in rec.items():
                    out["%s/%s" % (name, k2)] = v
                print(name, "len", len(acts), "reward", sum(rews), "-> hindsight", sum(hr), "done at end", hd[-1], "tb", int(env.total_brick))
    out["cases"] = np.array(names)
    fn = os.path.join(HERE, "relabel_static.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
