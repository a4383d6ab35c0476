"""Golden vectors for the env copies under script/Rainbow/env (THIS container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rainbow.py

  Env1D.py  Env1DStatic(args)                     (1, 7) raw counters; brick test `>` (:135); args.uniform_step -> step size 1, no draw
            Env1DDynamic(args, data_path, ...)    (1, 7); brick test `>` (:317); uniform_step likewise
  Env2D.py  Env2DStatic(args)                     (1, 51); brick test `>` (:166); uniform_step
            Env2DDynamic(args, data_path, ...)    (451, 1) = [window 49, count_brick, count_step, input_plan 400] as a column
  Env3D.py  Env3DStatic(args)                     (1, 51); brick test `>` (:234) and time test `>` (:257); uniform_step
            Env3DDynamic(args, data_path, ...)    (451,)
`args` carries plan_choose, half_window_size, history_length, uniform_step.  The modules import cv2 (stubbed, unused) and
common.utils of script/Rainbow.  Output: tests/golden/traj_rainbow.npz (fields as in traj_*.npz plus `uniform`).
"""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _refimport  # noqa: E402
import make_golden as mg  # noqa: E402

CASES = [
    (1, False, 0, "uniform", False, 1600), (1, False, 2, "drop", True, 1600), (1, True, ("sin", "train"), "drop", False, 1500),
    (1, True, ("sin", "val"), "uniform", True, 1600),
    (2, False, 0, "drop", False, 1200), (2, False, 1, "uniform", True, 1400), (2, True, ("dense", "train"), "drop", False, 1200),
    (2, True, ("sparse", "test"), "uniform", False, 1000),
    (3, False, 0, "build_right", True, 2000), (3, False, 1, "moves", False, 2800), (3, False, 1, "refmix", False, 1200),
    (3, True, ("dense", "train"), "uniform", False, 1200), (3, True, ("sparse", "val"), "build_right", False, 1500),
]


def load(dim):
    path = os.path.join(_refimport.REF, "script", "Rainbow", "env", "Env%dD.py" % dim)
    spec = importlib.util.spec_from_file_location("rainbow_env%dd" % dim, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run(mod, dim, dyn, plan, mix, uniform, n_steps, seed):
    W = mg.DIMS[dim]["W"]
    args = types.SimpleNamespace(plan_choose=None if dyn else plan, half_window_size=2 if dim == 1 else 3, history_length=4,
                                 uniform_step=uniform)
    cls = getattr(mod, "Env%dD%s" % (dim, "Dynamic" if dyn else "Static"))
    rng = np.random.default_rng(17000 + seed)
    acts = mg.mix_actions(rng, mg.MIXES[dim][mix], n_steps)
    np.random.seed(seed)
    env = cls(args, data_path=_refimport.dataset_path(dim, *plan), random_choose_paln=True) if dyn else cls(args)
    assert env.action_space() == mg.DIMS[dim]["A"] and env.get_features() == W + 1
    tail = dyn and dim != 1
    shape = {(1, False): (1, 7), (1, True): (1, 7), (2, False): (1, 51), (2, True): (451, 1), (3, False): (1, 51), (3, True): (451,)}[(dim, dyn)]
    rec = dict(actions=acts.astype(np.int8), step_size=np.zeros(n_steps, np.int8), win=np.zeros((n_steps, W), np.int16),
               sc=np.zeros((n_steps, 2)), reward=np.zeros(n_steps), done=np.zeros(n_steps, np.uint8), pos=np.zeros((n_steps, 2), np.int16))
    starts, finals, ious, tbs, pidx, rwin, rsc, plans = [], [], [], [], [], [], [], []

    def split(obs):
        o = np.asarray(obs)
        assert o.shape == shape and o.dtype == np.float64, (o.shape, shape)
        o = o.reshape(-1)
        if tail:
            assert np.array_equal(o[W + 2:], np.asarray(env.input_plan, np.float64).reshape(-1))
        return o[:W].astype(np.int16), o[W:W + 2].copy()

    def reset(t):
        w, sc = split(env.reset())
        starts.append(t); tbs.append(int(env.total_brick)); rwin.append(w); rsc.append(sc)
        pidx.append(int(env.index_random) if dyn else 0)
        plans.append(np.asarray(env.plan).astype(np.int16).reshape(-1))

    reset(0)
    for t in range(n_steps):
        obs, r, d = env.step(int(acts[t]))
        rec["win"][t], rec["sc"][t] = split(obs)
        rec["step_size"][t] = env.step_size
        rec["reward"][t] = float(r)
        rec["done"][t] = 1 if d else 0
        p = env.position_memory[-1]
        rec["pos"][t] = (p, 0) if dim == 1 else (p[0], p[1])
        if d or t == n_steps - 1:
            finals.append(np.asarray(env.environment_memory).astype(np.int16).reshape(-1))
            ious.append(float(env._iou()))
            if t != n_steps - 1:
                reset(t + 1)
    rec.update(ep_start=np.asarray(starts, np.int32), ep_total_brick=np.asarray(tbs, np.int32), ep_plan_idx=np.asarray(pidx, np.int32),
               ep_final_grid=np.stack(finals), ep_iou=np.asarray(ious), ep_reset_win=np.stack(rwin), ep_reset_sc=np.stack(rsc),
               ep_plan=np.stack(plans), seed=np.int64(seed), uniform=np.int8(uniform))
    return rec


def main():
    _refimport.install_gym_stub()
    _refimport.install_cv2_stub()
    _refimport.load_ref_classes()
    sys.path.insert(0, os.path.join(_refimport.REF, "script", "Rainbow"))
    mods = {d: load(d) for d in (1, 2, 3)}
    out, names = {}, []
    seed = 2700
    for dim, dyn, plan, mix, uniform, n in CASES:
        seed += 1
        name = "%dd_%s.%s.%s%s" % (dim, "dynamic" if dyn else "static", plan if not isinstance(plan, tuple) else "-".join(plan), mix,
                                  ".k1" if uniform else "")
        r = run(mods[dim], dim, dyn, plan, mix, uniform, n, seed)
        names.append(name)
        for k, v in r.items():
            out["%s/%s" % (name, k)] = v
        lens = np.diff(np.append(r["ep_start"], n))
        print("%-40s episodes %3d lengths %s step sizes %s rewards %s" % (name, len(lens), lens[:4].tolist(), np.unique(r["step_size"]).tolist(),
                                                                        sorted(set(r["reward"].tolist()))))
    out["cases"] = np.array(names)
    fn = os.path.join(HERE, "traj_rainbow.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
