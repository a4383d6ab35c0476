"""Golden vectors for the env copies under script/PPO and script/SAC/environments (THIS container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ppo.py          # -> traj_ppo.npz
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ppo.py sac      # -> traj_sac.npz

The SAC copies (script/SAC/environments/DMP_*.py) are the PPO ones with a 3-tuple step() (no info) and, for 1D / 2D, without
gym spaces; same flat observations and the same `>` termination tests (1D dynamic :94, 2D static :133, 3D static :216,:232).

stable-baselines flavoured forks of the six canonical classes: gym spaces, flat observations, 4-tuple step():

  script/PPO/1d_static/DMP_Env_1D_static.py                         obs (7,)
  script/PPO/1d_dynamic/DMP_Env_1D_dynamic_usedata_plan.py          obs (37,)  = [window 5, count_brick, count_step, plan 30]; brick test `>` (:93)
  script/PPO/2d_static/DMP_Env_2D_static.py                         obs (51,); brick test `>` (:137)
  script/PPO/2d_dynamic/DMP_Env_2d_dynamic_usedata_plan.py          obs (451,) = [window 49, count_brick, count_step, input_plan 400]
  script/PPO/3d_static/DMP_simulator_3d_static_circle.py            obs (51,); brick test `>` (:205), time test `>` (:221)
  script/PPO/3d_dynamic/DMP_simulator_3d_dynamic_triangle_usedata.py  obs (451,)
Every variant returns the raw counters and `info = {}`.  The files pass `dtype=np.int` to spaces.Box, an alias numpy removed
in 1.24: the capture sets `np.int = int` for the import (the only concession; the env code itself runs unmodified).
Each case seeds numpy's global stream and steps a recorded action stream, resetting after every done.
Output: tests/golden/traj_ppo.npz (fields as in traj_*.npz; `win` / `sc` are the first obs_dim entries of the flat vector,
the plan tail is checked against ep_plan here and not stored per step).
"""
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _refimport  # noqa: E402
import make_golden as mg  # noqa: E402

FORKS = {
    "1d_static": (1, False, "DMP_Env_1D_static.py", "deep_mobile_printing_1d1r"),
    "1d_dynamic": (1, True, "DMP_Env_1D_dynamic_usedata_plan.py", "deep_mobile_printing_1d1r"),
    "2d_static": (2, False, "DMP_Env_2D_static.py", "deep_mobile_printing_2d1r"),
    "2d_dynamic": (2, True, "DMP_Env_2d_dynamic_usedata_plan.py", "deep_mobile_printing_2d1r"),
    "3d_static": (3, False, "DMP_simulator_3d_static_circle.py", "deep_mobile_printing_3d1r"),
    "3d_dynamic": (3, True, "DMP_simulator_3d_dynamic_triangle_usedata.py", "deep_mobile_printing_3d1r"),
}
SAC_FORKS = {
    "1d_static": (1, False, "DMP_Env_1D_static.py", "deep_mobile_printing_1d1r"),
    "1d_dynamic": (1, True, "DMP_Env_1D_dynamic.py", "deep_mobile_printing_1d1r"),
    "2d_static": (2, False, "DMP_Env_2D_static.py", "deep_mobile_printing_2d1r"),
    "2d_dynamic": (2, True, "DMP_Env_2D_dynamic.py", "deep_mobile_printing_2d1r"),
    "3d_static": (3, False, "DMP_simulator_3d_static_circle.py", "deep_mobile_printing_3d1r"),
    "3d_dynamic": (3, True, "DMP_simulator_3d_dynamic_triangle_usedata.py", "deep_mobile_printing_3d1r"),
}
SUITE = "ppo"
CASES = [
    ("1d_static", 0, "uniform", 1600), ("1d_static", 2, "drop", 1600),
    ("1d_dynamic", ("sin", "train"), "drop", 2400), ("1d_dynamic", ("sin", "val"), "uniform", 1600),
    ("2d_static", 0, "drop", 1800), ("2d_static", 1, "uniform", 1400),
    ("2d_dynamic", ("dense", "train"), "drop", 1800), ("2d_dynamic", ("sparse", "test"), "uniform", 1400),
    ("3d_static", 0, "build_right", 2400), ("3d_static", 1, "moves", 2800), ("3d_static", 1, "refmix", 1500),
    ("3d_dynamic", ("dense", "train"), "uniform", 1500), ("3d_dynamic", ("sparse", "val"), "build_right", 2200),
]


def install_box_stub():
    import gym

    class Box(object):
        def __init__(self, low, high, dtype=None):
            self.low, self.high, self.dtype, self.shape = np.asarray(low), np.asarray(high), dtype, np.asarray(low).shape

    gym.spaces.Box = Box


def load(fork):
    dim, dyn, fn, cname = (FORKS if SUITE == "ppo" else SAC_FORKS)[fork]
    path = os.path.join(_refimport.REF, "script", "PPO", fork, fn) if SUITE == "ppo" else os.path.join(
        _refimport.REF, "script", "SAC", "environments", fn)
    spec = importlib.util.spec_from_file_location("%s_%s" % (SUITE, fork), path)
    mod = importlib.util.module_from_spec(spec)
    had = hasattr(np, "int")
    if not had:
        np.int = int
    try:
        spec.loader.exec_module(mod)
        return getattr(mod, cname), (lambda: None if had else delattr(np, "int"))
    except Exception:
        if not had:
            del np.int
        raise


def run(fork, plan, mix, n_steps, seed):
    dim, dyn, _, _ = FORKS[fork]
    W = mg.DIMS[dim]["W"]
    cls, restore = load(fork)
    rng = np.random.default_rng(15000 + seed)
    acts = mg.mix_actions(rng, mg.MIXES[dim][mix], n_steps)
    np.random.seed(seed)
    env = cls(data_path=_refimport.dataset_path(dim, *plan), random_choose_paln=True) if dyn else cls(plan_choose=plan)
    restore()
    D = W + 2 + ((30 if dim == 1 else 400) if dyn else 0)
    if SUITE == "ppo" or dim == 3:
        assert env.action_space.n == mg.DIMS[dim]["A"] and env.observation_space.shape == (D,)
    else:
        assert not hasattr(env, "action_space")
    rec = dict(actions=acts.astype(np.int8), step_size=np.zeros(n_steps, np.int8), win=np.zeros((n_steps, W), np.int16),
               sc=np.zeros((n_steps, 2)), reward=np.zeros(n_steps), done=np.zeros(n_steps, np.uint8), pos=np.zeros((n_steps, 2), np.int16))
    starts, finals, ious, tbs, pidx, rwin, rsc, plans = [], [], [], [], [], [], [], []

    def split(obs):
        o = np.asarray(obs)
        assert o.shape == (D,) and o.dtype == np.float64
        if dyn:
            tail = np.asarray(env.plan if dim == 1 else env.input_plan, np.float64).reshape(-1)
            assert np.array_equal(o[W + 2:], tail)
        return o[:W].astype(np.int16), o[W:W + 2].copy()

    def reset(t):
        w, sc = split(env.reset())
        starts.append(t); tbs.append(int(env.total_brick)); rwin.append(w); rsc.append(sc)
        pidx.append(int(env.index_random) if dyn else 0)
        plans.append(np.asarray(env.plan).astype(np.int16).reshape(-1))

    reset(0)
    for t in range(n_steps):
        ret = env.step(int(acts[t]))
        assert len(ret) == (4 if SUITE == "ppo" else 3) and (SUITE != "ppo" or ret[3] == {})
        obs, r, d = ret[:3]
        rec["win"][t], rec["sc"][t] = split(obs)
        rec["step_size"][t] = env.step_size
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
               ep_plan=np.stack(plans), seed=np.int64(seed))
    return rec


def main():
    global SUITE
    SUITE = "sac" if len(sys.argv) > 1 and sys.argv[1] == "sac" else "ppo"
    _refimport.install_gym_stub()
    _refimport.install_cv2_stub()
    install_box_stub()
    _refimport.load_ref_classes()
    out, names = {}, []
    seed = 700 if SUITE == "ppo" else 1700
    for fork, plan, mix, n in CASES:
        seed += 1
        name = "%s.%s.%s" % (fork, plan if not isinstance(plan, tuple) else "-".join(plan), mix)
        r = run(fork, plan, mix, n, seed)
        names.append(name)
        for k, v in r.items():
            out["%s/%s" % (name, k)] = v
        lens = np.diff(np.append(r["ep_start"], n))
        print("%-36s episodes %3d lengths %s tb %s rewards %s" % (name, len(lens), lens[:5].tolist(), r["ep_total_brick"][:3].tolist(),
                                                                 sorted(set(r["reward"].tolist()))))
    out["cases"] = np.array(names)
    fn = os.path.join(HERE, "traj_%s.npz" % SUITE)
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
