"""Golden vectors for Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py (THIS container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hindsight_2d_dynamic.py

deep_mobile_printing_2d1r_hindsight(data_path, random_choose_paln): the 2D dataset class with step(action, step_size), raw
counters in every observation, and a reset() that first draws a THROW-AWAY random triangle with cv2 (create_plan, :37-59: two
np.random.randint(0, 20, size=3) per attempt, cv2.polylines (+ cv2.fillPoly when dense), redrawn while the area is <= 50 / 20)
before it takes the dataset plan (:61-72).  cv2 is not installed here, so the class is imported with a `cv2` stand-in whose
polylines / fillPoly draw with THE BUILD'S restatement of cv2's rules (oracle/snac_oracle.c orc_raster_triangle -- the one that
reproduces all 2000 cv2-drawn plans the reference ships, tests/test_plan_generators.py).  What the recording therefore pins is the
class's own logic: how reset() consumes np.random (the number of redraws depends on the rasterised area), the order of the index
draw behind it, the sequential plan order, step(action, step_size), the observation format.  The rasteriser itself is the build's:
"pinned modulo rasteriser" (tests/golden/README.md).
Output: tests/golden/traj_hindsight_dynamic_2d.npz, fields as in traj_misc.npz plus per-episode `ep_draws` (np.random words the
reset consumed, measured with a counting RandomState twin) and `rng_after` (one more randint of the global stream at the end).
"""
import importlib
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refimport  # noqa: E402
import make_golden as mg  # noqa: E402
import rng_spec  # noqa: E402
from oracle import snac_oracle  # noqa: E402


def install_cv2_standin():
    """cv2.polylines(img, [pts], isClosed=True, color) / cv2.fillPoly(img, [pts], color) for ONE triangle on a 20 x 20 image:
    the cells the build's rasteriser sets (outline / outline + interior) take `color`.  Anything else is a loud error."""
    cv2 = types.ModuleType("cv2")

    def _tri(pts_list):
        assert len(pts_list) == 1
        p = np.asarray(pts_list[0]).reshape(-1, 2)
        assert p.shape == (3, 2)
        return p[:, 0], p[:, 1]                                   # x = column, y = row

    def polylines(img, pts, isClosed, color, *a, **kw):
        assert isClosed and not a and not kw and img.shape[:2] == (20, 20)
        mask, _ = snac_oracle.raster_triangle(*_tri(pts), 1)     # the outline alone
        img[mask != 0] = color
        return img

    def fillPoly(img, pts, color, *a, **kw):
        assert not a and not kw and img.shape[:2] == (20, 20)
        mask, _ = snac_oracle.raster_triangle(*_tri(pts), 0)     # outline + interior
        img[mask != 0] = color
        return img

    cv2.polylines, cv2.fillPoly = polylines, fillPoly
    sys.modules["cv2"] = cv2


def run(env, acts, ks, seed):
    n_steps, W = len(acts), 49
    rec = dict(actions=acts.astype(np.int8), step_size=ks.astype(np.int8), win=np.zeros((n_steps, W), np.int16), sc=np.zeros((n_steps, 2)),
               reward=np.zeros(n_steps), done=np.zeros(n_steps, np.uint8), pos=np.zeros((n_steps, 2), np.int16))
    starts, finals, ious, tbs, pidx, rwin, rsc, thrown = [], [], [], [], [], [], [], []

    def reset(t):
        obs = env.reset()
        assert len(obs) == 3 and obs[1] is env.input_plan and list(obs[2]) == [3, 3]
        o = np.asarray(obs[0], np.float64).reshape(-1)
        starts.append(t); tbs.append(int(env.total_brick)); rwin.append(o[:W].astype(np.int16)); rsc.append(o[W:].copy())
        pidx.append(int(env.index_random) if env.random_choose_paln else (env.index_for_non_random - 1) % env.plan_dataset_len)

    reset(0)
    for t in range(n_steps):
        obs, r, d = env.step(int(acts[t]), int(ks[t]))
        assert len(obs) == 3 and obs[1] is env.input_plan and list(obs[2]) == list(env.position_memory[-1]) and env.step_size == ks[t]
        o = np.asarray(obs[0], np.float64).reshape(-1)
        rec["win"][t] = o[:W].astype(np.int16)
        rec["sc"][t] = o[W:]
        rec["reward"][t] = float(r)
        rec["done"][t] = 1 if d else 0
        rec["pos"][t] = env.position_memory[-1]
        if d or t == n_steps - 1:
            finals.append(np.asarray(env.environment_memory).astype(np.int16).reshape(-1))
            ious.append(mg.cur_iou(2, env))
            if t != n_steps - 1:
                reset(t + 1)
    rec.update(ep_start=np.asarray(starts, np.int32), ep_total_brick=np.asarray(tbs, np.int32), ep_plan_idx=np.asarray(pidx, np.int32),
               ep_final_grid=np.stack(finals), ep_iou=np.asarray(ious), ep_reset_win=np.stack(rwin), ep_reset_sc=np.stack(rsc),
               seed=np.int64(seed), rng_after=np.int64(np.random.randint(0, 1 << 30)))
    return rec


def main():
    install_cv2_standin()
    _refimport.load_ref_classes()                                 # gym stub + sys.path
    cls = getattr(importlib.import_module("DMP_Env_2D_dynamic_hindsight_replay_usedata"), "deep_mobile_printing_2d1r_hindsight")
    out, names = {}, []
    for seed, dens, split, mix, rnd in ((61, "dense", "train", "uniform", True), (62, "sparse", "val", "drop", True),
                                        (63, "dense", "test", "uniform", False), (64, "sparse", "train", "walk", True)):
        n = 2400
        w = rng_spec.words(seed, rng_spec.STREAM_STEP, np.uint64(2), np.arange(n, dtype=np.uint64))
        ks = rng_spec.step_size_of(w)
        key = mix if mix in mg.MIXES[2] else "uniform"
        acts = mg.mix_actions(np.random.default_rng(seed), mg.MIXES[2][key], n)
        np.random.seed(seed)
        env = cls(data_path=_refimport.dataset_path(2, dens, split), random_choose_paln=rnd)
        r = run(env, acts, ks, seed)
        r["random_choose"] = np.int64(1 if rnd else 0)
        name = "2dhd.%s-%s.%s%s" % (dens, split, key, "" if rnd else ".seq")
        names.append(name)
        for k, v in r.items():
            out["%s/%s" % (name, k)] = v
        print("%-34s episodes %3d rewards %s rng_after %d" % (name, len(r["ep_start"]), sorted(set(r["reward"].tolist())), int(r["rng_after"])))
    out["cases"] = np.array(names)
    fn = os.path.join(HERE, "traj_hindsight_dynamic_2d.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn))


if __name__ == "__main__":
    main()
