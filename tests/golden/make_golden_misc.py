"""Golden vectors for the remaining Env/ modules (THIS container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_misc.py

  Env/3D/DMP_simulator_3d_dynamic_triangle_hindsight_replay.py   deep_mobile_printing_3d1r_hindsight(data_path, ...):
        the dataset class with step(action, step_size); reset() -> [obs with raw counters, input_plan] (:71-73), step() ->
        the canonical normalised [obs, input_plan, position] (:199-228)
  Env/1D/DMP_Env_1D_static_test.py, Env/2D/DMP_Env_2D_static_test.py, Env/3D/DMP_simulator_3d_static_circle_test.py
        the canonical static classes with another render() (1D also spells count_brick correctly); imported by the
        test_*.py scripts (script/DRQN_hindsight/*/test_*, script/Handcraft_SLAM/test_slam_*_static.py)
(Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py rasterises a throw-away triangle with cv2 inside reset(): not runnable here.)
Output: tests/golden/traj_misc.npz, fields as in traj_*.npz.
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


def run(env, dim, dyn, acts, ks, seed, hindsight):
    n_steps, W = len(acts), mg.DIMS[dim]["W"]
    rec = dict(actions=acts.astype(np.int8), step_size=np.zeros(n_steps, np.int8), win=np.zeros((n_steps, W), np.int16),
               sc=np.zeros((n_steps, 2)), reward=np.zeros(n_steps), done=np.zeros(n_steps, np.uint8), pos=np.zeros((n_steps, 2), np.int16))
    starts, finals, ious, tbs, pidx, rwin, rsc = [], [], [], [], [], [], []

    def reset(t):
        obs = env.reset()
        if hindsight:
            assert len(obs) == 2 and obs[1] is env.input_plan
            obs = obs[0]
        o = np.asarray(obs, np.float64).reshape(-1)
        starts.append(t); tbs.append(int(env.total_brick)); rwin.append(o[:W].astype(np.int16)); rsc.append(o[W:].copy())
        pidx.append(int(env.index_random) if dyn else 0)

    reset(0)
    for t in range(n_steps):
        if hindsight:
            obs, r, d = env.step(int(acts[t]), int(ks[t]))
            assert len(obs) == 3 and obs[1] is env.input_plan and list(obs[2]) == list(env.position_memory[-1])
            obs = obs[0]
        else:
            obs, r, d = env.step(int(acts[t]))
        o = np.asarray(obs, np.float64).reshape(-1)
        rec["step_size"][t] = env.step_size
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
    rec.update(ep_start=np.asarray(starts, np.int32), ep_total_brick=np.asarray(tbs, np.int32), ep_plan_idx=np.asarray(pidx, np.int32),
               ep_final_grid=np.stack(finals), ep_iou=np.asarray(ious), ep_reset_win=np.stack(rwin), ep_reset_sc=np.stack(rsc),
               seed=np.int64(seed))
    return rec


def main():
    _refimport.load_ref_classes()
    out, names = {}, []

    def add(name, r):
        names.append(name)
        for k, v in r.items():
            out["%s/%s" % (name, k)] = v
        print("%-28s episodes %3d rewards %s" % (name, len(r["ep_start"]), sorted(set(r["reward"].tolist()))))

    cls = getattr(importlib.import_module("DMP_simulator_3d_dynamic_triangle_hindsight_replay"), "deep_mobile_printing_3d1r_hindsight")
    for seed, dens, split, mix in ((41, "dense", "train", "uniform"), (42, "sparse", "val", "refmix"), (43, "dense", "test", "build_right")):
        n = 2200
        w = rng_spec.words(seed, rng_spec.STREAM_STEP, np.uint64(3), np.arange(n, dtype=np.uint64))
        ks = rng_spec.step_size_of(w)
        acts = mg.mix_actions(np.random.default_rng(seed), mg.MIXES[3][mix], n)
        np.random.seed(seed)
        env = cls(data_path=_refimport.dataset_path(3, dens, split), random_choose_paln=True)
        add("3dhd.%s-%s.%s" % (dens, split, mix), run(env, 3, True, acts, ks, seed, True))
    mods = {1: "DMP_Env_1D_static_test", 2: "DMP_Env_2D_static_test", 3: "DMP_simulator_3d_static_circle_test"}
    for dim in (1, 2, 3):
        cls = getattr(importlib.import_module(mods[dim]), "deep_mobile_printing_%dd1r" % dim)
        for pc, mix in ((0, "uniform"), (1, "drop" if dim != 3 else "build_right")):
            seed = 50 + 2 * dim + pc
            n = 1500
            acts = mg.mix_actions(np.random.default_rng(seed), mg.MIXES[dim][mix], n)
            np.random.seed(seed)
            env = cls(plan_choose=pc)
            add("test%dd.p%d.%s" % (dim, pc, mix), run(env, dim, False, acts, None, seed, False))
    out["cases"] = np.array(names)
    fn = os.path.join(HERE, "traj_misc.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
