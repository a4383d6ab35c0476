"""Golden vectors for the reference's L-Net env variants (THIS container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_lnet.py

  Env/1D/DMP_Env_1D_static_Lnet.py                obs (1, 8) = [window 5, count_brick, count_step, position]
  Env/2D/DMP_Env_2D_static_Lnet.py                obs = [(1, 51) with frame value 2 and normalised scalars, position]
  Env/3D/DMP_simulator_3d_static_circle_Lnet.py   obs = [(1, 51) normalised scalars, position]; dynamic rules, T = 1300

Each case seeds numpy's global stream (the envs draw their step sizes from it) and steps a recorded action stream,
resetting after every done.  Output: tests/golden/traj_lnet.npz (fields as in traj_*.npz).
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

MODS = {1: "DMP_Env_1D_static_Lnet", 2: "DMP_Env_2D_static_Lnet", 3: "DMP_simulator_3d_static_circle_Lnet"}


def split_obs(dim, obs):
    W = mg.DIMS[dim]["W"]
    if dim == 1:
        o = np.asarray(obs, np.float64).reshape(-1)
        assert o.size == 8
        return o[:W], o[W:W + 2], int(o[7])
    o = np.asarray(obs[0], np.float64).reshape(-1)
    return o[:W], o[W:W + 2], tuple(obs[1])


def run(cls, dim, pc, seed, probs, n_steps):
    W = mg.DIMS[dim]["W"]
    rng = np.random.default_rng(8000 + seed)
    acts = mg.mix_actions(rng, probs, n_steps)
    np.random.seed(seed)
    env = cls(plan_choose=pc)
    rec = dict(actions=acts, step_size=np.zeros(n_steps, np.int8), win=np.zeros((n_steps, W), np.int16), sc=np.zeros((n_steps, 2)),
               reward=np.zeros(n_steps), done=np.zeros(n_steps, np.uint8), pos=np.zeros((n_steps, 2), np.int16))
    starts, finals, ious, tbs, rwin, rsc = [], [], [], [], [], []

    def reset(t):
        w, sc, _ = split_obs(dim, env.reset())
        starts.append(t); tbs.append(int(env.total_brick)); rwin.append(w.astype(np.int16)); rsc.append(sc.copy())

    reset(0)
    for t in range(n_steps):
        obs, r, d = env.step(int(acts[t]))
        w, sc, p = split_obs(dim, obs)
        rec["step_size"][t] = env.step_size
        rec["win"][t] = w.astype(np.int16)
        rec["sc"][t] = sc
        rec["reward"][t] = float(r)
        rec["done"][t] = 1 if d else 0
        rec["pos"][t] = (p, 0) if dim == 1 else p
        pm = env.position_memory[-1]
        assert (pm == p) if dim == 1 else (tuple(pm) == tuple(p))
        if d or t == n_steps - 1:
            finals.append(np.asarray(env.environment_memory).astype(np.int16).reshape(-1))
            ious.append(mg.cur_iou(dim, env) if dim != 2 else _iou2(env))
            if t != n_steps - 1:
                reset(t + 1)
    rec.update(ep_start=np.asarray(starts, np.int32), ep_total_brick=np.asarray(tbs, np.int32), ep_final_grid=np.stack(finals),
               ep_iou=np.asarray(ious), ep_reset_win=np.stack(rwin), ep_reset_sc=np.stack(rsc), seed=np.int64(seed))
    return rec


def _iou2(env):
    # interior cells only ever hold 0 / 1; the frame value 2 lies outside the compared region
    h = env.HALF_WINDOW_SIZE
    g = env.environment_memory[h:h + env.plan_height, h:h + env.plan_width]
    p = env.plan[h:h + env.plan_height, h:h + env.plan_width]
    return float(np.sum(np.logical_and(g, p)) / np.sum(np.logical_or(g, p)))


def main():
    _refimport.load_ref_classes()
    out, names = {}, []
    seed = 500
    for dim in (1, 2, 3):
        cls = getattr(importlib.import_module(MODS[dim]), "deep_mobile_printing_%dd1r" % dim)
        mixes = {k: v for k, v in mg.MIXES[dim].items() if k in ("uniform", "drop", "refmix", "build_right", "sparse_build")}
        for pc in ((0, 1, 2) if dim == 1 else (0, 1)):
            for mix, probs in mixes.items():
                seed += 1
                name = "%dd.p%d.%s" % (dim, pc, mix)
                r = run(cls, dim, pc, seed, probs, 2000)
                names.append(name)
                for k, v in r.items():
                    out["%s/%s" % (name, k)] = v
                print(name, "episodes", len(r["ep_start"]), "rewards", sorted(set(r["reward"].tolist())), "win values", np.unique(r["win"]).tolist()[:6])
    out["cases"] = np.array(names)
    fn = os.path.join(HERE, "traj_lnet.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
