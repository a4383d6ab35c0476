"""Plan tables: the static plans of the reference as integer tables, the converted plan datasets, and
the packing into the device layout of include/snac_hip.h.  Host-side numpy only."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data", "plans.npz")

# Env/1D/DMP_Env_1D_static.py:34-55 -- 0: round(10 sin(2 pi x / 30) + 20), 1: round(100 N(x; 0, 3) + 17) on
# linspace(-18, 18, 30), 2: steps of 25 / 15.  Integer results, independent of libm.
_PLAN_1D = (
    (20, 22, 24, 26, 27, 29, 30, 30, 30, 30, 29, 27, 26, 24, 22, 20, 18, 16, 14, 13, 11, 10, 10, 10, 10, 11, 13, 14, 16, 18),
    (17,) * 9 + (18, 19, 22, 25, 28, 30, 30, 28, 25, 22, 19, 18) + (17,) * 9,
    ((25,) * 5 + (15,) * 5) * 3,
)
# Env/2D/DMP_Env_2D_static.py:31-52 -- CirclePolygon((12.5, 12.5), r).contains_point((i, j)) on the 26x26 grid:
# 0 dense r = 7; 1 sparse ring r_out = 8 minus r_in = 7.  (first row, 26-bit row masks, bit 25 - col)
_PLAN_2D = (
    (6, (30720, 130560, 261888, 524160, 524160, 1048512, 1048512, 1048512, 1048512, 524160, 524160, 261888, 130560, 30720)),
    (5, (64512, 231168, 393600, 786624, 524352, 1572960, 1048608, 1048608, 1048608, 1048608, 1572960, 524352, 786624,
         393600, 231168, 64512)),
)


def static_plan(dim, plan_choose):
    """The bordered plan array exactly as the reference's create_plan() returns it (float64)."""
    if dim == 1:
        if plan_choose not in (0, 1, 2):
            raise ValueError('0: Sin, 1: Gaussian, 2: Step')
        return np.asarray(_PLAN_1D[plan_choose], np.float64)
    if plan_choose not in (0, 1):
        raise ValueError('0: Dense circle, 1: Sparse circle')
    first, masks = _PLAN_2D[plan_choose]
    plan = np.zeros((26, 26), np.float64)
    for i, m in enumerate(masks):
        for c in range(26):
            plan[first + i, c] = (m >> (25 - c)) & 1
    if dim == 3:
        plan = plan * 6  # plan * self.z, Env/3D/DMP_simulator_3d_static_circle.py:63
    return plan


def random_sin_plan(plan_width=30, plan_height=20):
    """The random sin-curve plan generator of Env/1D/DMP_Env_1D_dynamic_hindsight_replay.py:29-42 (create_plan): the same
    numpy calls in the same order on numpy's global stream, so a seeded script sees the same plans.
    -> (plan float64[30], area, [k_1, k_2, phase])"""
    k_1 = np.random.uniform(3, 12)
    k_2 = np.random.randint(1, 4)
    phase = np.random.uniform(-1, 1) * np.pi
    x = np.arange(plan_width)
    y = np.round((k_1 * np.sin(2 * np.pi / plan_width * (k_2 * x + phase)) + plan_height))
    return y, sum(y), [k_1, k_2, phase]


def dataset(dim, density="dense", split="train"):
    """Converted copy of Env/<dim>D/data_*_envplan_500_<split>.pkl -> float64 array [P, ...] like np.asarray(joblib.load(..))."""
    key = "1d_sin_%s" % split if dim == 1 else "%dd_%s_%s" % (dim, density, split)
    with np.load(DATA) as z:
        return z[key].astype(np.float64)


def load_plan_file(path):
    """A plan dataset as the reference's constructors take it (`data_path`): a joblib pickle of a list of arrays,
    or an .npy / .npz (first array) holding [P, ...]."""
    if path.endswith(".npy"):
        return np.asarray(np.load(path), np.float64)
    if path.endswith(".npz"):
        with np.load(path) as z:
            return np.asarray(z[z.files[0]], np.float64)
    import joblib

    return np.asarray(joblib.load(path), np.float64)


def pack_plans(dim, plans):
    """plans [P, 30] (1D) or [P, 26, 26] (2D/3D), integer-valued -> (device-layout table, total_brick int16[P]).

    total_brick: sum(plan) (Env/1D/..static.py:53, Env/3D/..triangle_usedata.py:49), floored at 30 for 2D only
    (Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:45-46)."""
    plans = np.asarray(plans)
    if not np.array_equal(plans, np.round(plans)):
        raise ValueError("plans must be integer-valued")
    plans = plans.astype(np.int64)
    P = len(plans)
    if P < 1 or P > 32767:
        raise ValueError("number of plans out of range")
    if dim == 1:
        if plans.shape != (P, 30) or plans.min() < 0 or plans.max() > 32767:
            raise ValueError("1D plans must be [P, 30] heights")
        out = np.zeros((P, 32), np.int16)
        out[:, :30] = plans
        tb = plans.sum(axis=1)
    else:
        if plans.shape != (P, 26, 26):
            raise ValueError("%dD plans must be [P, 26, 26]" % dim)
        inner = plans[:, 3:23, 3:23]
        if plans.sum() != inner.sum() or plans.min() < 0:
            raise ValueError("plan cells outside the 20x20 interior must be 0")
        tb = inner.reshape(P, -1).sum(axis=1)
        if dim == 2:
            if inner.max() > 1:
                raise ValueError("2D plans must be 0/1")
            out = (inner.astype(np.uint64) << np.arange(20, dtype=np.uint64)[None, None, :]).sum(axis=2).astype(np.uint32)
            tb = np.maximum(tb, 30)
        else:
            if inner.max() > 32767:
                raise ValueError("3D plan heights too large")
            out = inner.reshape(P, 400).astype(np.int16)
    if tb.max() > 32767:
        raise ValueError("total_brick too large")
    return np.ascontiguousarray(out), tb.astype(np.int16)
