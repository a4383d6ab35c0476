"""Golden vectors for the reference's 1D DYNAMIC hindsight-replay env (THIS container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hindsight_dynamic.py

  Env/1D/DMP_Env_1D_dynamic_hindsight_replay.py   deep_mobile_printing_1d1r_hindsight(): every reset() draws a random sin
      curve from numpy's global stream (create_plan :29-42: uniform, randint, uniform), step(action, step_size)

The 2D / 3D dynamic hindsight files rasterise triangles with cv2 inside reset() (absent here; the number of RNG draws
depends on the rasteriser), so they cannot be run, let alone pinned, in this container.
Each case seeds numpy's global stream; actions and step sizes come from the counter RNG (tests/rng_spec.py).
Output: tests/golden/traj_hindsight_dynamic_1d.npz (per-step / per-episode fields as in traj_*.npz, plus ep_plan, ep_one_hot).
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
import rng_spec  # noqa: E402


def run(cls, seed, n_steps, drop_heavy):
    w = rng_spec.words(seed, rng_spec.STREAM_STEP, np.uint64(7), np.arange(n_steps, dtype=np.uint64))
    acts = rng_spec.action_of(w, 3)
    if drop_heavy:
        acts = np.where(np.arange(n_steps) % drop_heavy != drop_heavy - 1, 2, acts).astype(np.int8)
    ks = rng_spec.step_size_of(w)
    np.random.seed(seed)
    env = cls()
    rec = dict(actions=acts.astype(np.int8), step_size=ks.astype(np.int8), win=np.zeros((n_steps, 5), np.int16),
               sc=np.zeros((n_steps, 2)), reward=np.zeros(n_steps), done=np.zeros(n_steps, np.uint8), pos=np.zeros((n_steps, 2), np.int16))
    starts, finals, ious, tbs, rwin, rsc, plans, hots = [], [], [], [], [], [], [], []

    def reset(t):
        obs = env.reset()
        o = np.asarray(obs[0], np.float64).reshape(-1)
        assert obs[1] is env.plan
        starts.append(t); tbs.append(float(env.total_brick)); rwin.append(o[:5].astype(np.int16)); rsc.append(o[5:].copy())
        plans.append(np.asarray(env.plan, np.float64).copy()); hots.append(np.asarray(env.one_hot, np.float64))

    reset(0)
    for t in range(n_steps):
        obs, r, d = env.step(int(acts[t]), int(ks[t]))
        o = np.asarray(obs[0], np.float64).reshape(-1)
        assert obs[1] is env.plan
        rec["win"][t] = o[:5].astype(np.int16)
        rec["sc"][t] = o[5:]
        rec["reward"][t] = float(r)
        rec["done"][t] = 1 if d else 0
        rec["pos"][t] = (env.position_memory[-1], 0)
        if d or t == n_steps - 1:
            finals.append(np.asarray(env.environment_memory).astype(np.int16).reshape(-1))
            ious.append(mg.cur_iou(1, env))
            if t != n_steps - 1:
                reset(t + 1)
    rec.update(ep_start=np.asarray(starts, np.int32), ep_total_brick=np.asarray(tbs), ep_final_grid=np.stack(finals),
               ep_iou=np.asarray(ious), ep_reset_win=np.stack(rwin), ep_reset_sc=np.stack(rsc), ep_plan=np.stack(plans),
               ep_one_hot=np.stack(hots), seed=np.int64(seed))
    return rec


def main():
    _refimport.load_ref_classes()
    cls = getattr(importlib.import_module("DMP_Env_1D_dynamic_hindsight_replay"), "deep_mobile_printing_1d1r_hindsight")
    out, names = {}, []
    for seed, heavy in ((31, 0), (32, 4), (33, 16)):
        name = "1d.s%d.%s" % (seed, "drop%d" % heavy if heavy else "uniform")
        r = run(cls, seed, 4000, heavy)
        names.append(name)
        for k, v in r.items():
            out["%s/%s" % (name, k)] = v
        print(name, "episodes", len(r["ep_start"]), "total_brick", r["ep_total_brick"][:6], "rewards", sorted(set(r["reward"].tolist())),
              "plan range", r["ep_plan"].min(), r["ep_plan"].max())
    out["cases"] = np.array(names)
    fn = os.path.join(HERE, "traj_hindsight_dynamic_1d.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
