"""Golden vectors for the reference's static hindsight-replay env variants (THIS container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hindsight.py

  Env/1D/DMP_Env_1D_static_hindsight_replay.py      deep_mobile_printing_1d1r_hindsight.step(action, step_size)
  Env/2D/DMP_Env_2D_static_hindsight_replay.py      deep_mobile_printing_2d1r_hindsight.step(action, step_size)
  Env/3D/DMP_simulator_3d_static_circle_hindsight_replay.py   deep_mobile_printing_3d1r_hindsight.step(action, step_size)

These take the step size from the caller instead of drawing it (the reference's own precedent for the build's explicit
`step_size` input).  Step sizes and actions here come from the counter RNG (tests/rng_spec.py), so the file needs no
numpy RNG state.  The dynamic hindsight variants synthesise plans with cv2 (absent here) and are not captured.
Output: tests/golden/traj_hindsight_static.npz (same per-step / per-episode fields as traj_*.npz).
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

MODS = {1: "DMP_Env_1D_static_hindsight_replay", 2: "DMP_Env_2D_static_hindsight_replay",
        3: "DMP_simulator_3d_static_circle_hindsight_replay"}


def run(cls, dim, plan_choose, seed, n_steps, drop_heavy):
    A, W = mg.DIMS[dim]["A"], mg.DIMS[dim]["W"]
    w = rng_spec.words(seed, rng_spec.STREAM_STEP, np.uint64(plan_choose), np.arange(n_steps, dtype=np.uint64))
    acts = rng_spec.action_of(w, A)
    if drop_heavy:  # every second action becomes the drop / a build so that count_brick >= total_brick is reached
        acts = np.where(np.arange(n_steps) % 2 == 0, A - 1 if dim != 3 else 5, acts).astype(np.int8)
    ks = rng_spec.step_size_of(w)
    env = cls(plan_choose=plan_choose)
    rec = dict(actions=acts.astype(np.int8), step_size=ks.astype(np.int8), win=np.zeros((n_steps, W), np.int16),
               sc=np.zeros((n_steps, 2)), reward=np.zeros(n_steps), done=np.zeros(n_steps, np.uint8), pos=np.zeros((n_steps, 2), np.int16))
    starts, finals, ious, tbs, rwin, rsc = [], [], [], [], [], []

    def reset(t):
        o = np.asarray(env.reset(), np.float64).reshape(-1)
        starts.append(t)
        tbs.append(int(env.total_brick))
        rwin.append(o[:W].astype(np.int16))
        rsc.append(o[W:].copy())

    reset(0)
    for t in range(n_steps):
        obs, r, d = env.step(int(acts[t]), int(ks[t]))
        o = np.asarray(obs, np.float64).reshape(-1)
        rec["win"][t] = o[:W].astype(np.int16)
        rec["sc"][t] = o[W:]
        rec["reward"][t] = float(r)
        rec["done"][t] = 1 if d else 0
        p = env.position_memory[-1]
        rec["pos"][t] = (p, 0) if dim == 1 else (p[0], p[1])
        if d or t == n_steps - 1:
            finals.append(np.asarray(env.environment_memory).astype(np.int16).reshape(-1))
            ious.append(mg.cur_iou(dim, env))
            if t != n_steps - 1:
                reset(t + 1)
    rec.update(ep_start=np.asarray(starts, np.int32), ep_total_brick=np.asarray(tbs, np.int32), ep_final_grid=np.stack(finals),
               ep_iou=np.asarray(ious), ep_reset_win=np.stack(rwin), ep_reset_sc=np.stack(rsc), seed=np.int64(seed))
    return rec


def main():
    _refimport.load_ref_classes()  # installs the gym stub and the Env paths
    out, names = {}, []
    for dim in (1, 2, 3):
        cls = getattr(importlib.import_module(MODS[dim]), "deep_mobile_printing_%dd1r_hindsight" % dim)
        for pc in ((0, 1, 2) if dim == 1 else (0, 1)):
            for heavy in (False, True):
                name = "%dd.p%d.%s" % (dim, pc, "drop" if heavy else "uniform")
                r = run(cls, dim, pc, 900 + 10 * dim + pc, 1800, heavy)
                names.append(name)
                for k, v in r.items():
                    out["%s/%s" % (name, k)] = v
                print(name, "episodes", len(r["ep_start"]), "rewards", sorted(set(r["reward"].tolist())))
    out["cases"] = np.array(names)
    fn = os.path.join(HERE, "traj_hindsight_static.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
