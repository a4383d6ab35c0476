"""Long rollouts of the 3D layout variants on k_rollout3db's variant forms against the oracle: whole episodes by the time limit (1300 / 1000
ticks) and many episodes ended by boxed-in agents, several launches in a row on the same batch, ragged last blocks, both row types.

    gpurun -- 'SNAC_3D_BLOCK_VAR_PLAN_F64=4 SNAC_3D_BLOCK_VAR_PLAN_F32=4 python tools/var3d_soak.py'      (the env overrides: rows with the plan tail at 1060 envs)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import helpers  # noqa: E402
from snac_amd import BatchedDMPEnv, _lib  # noqa: E402

CASES = [(1060, (1400, 777), dict(obs_tail=("record",)), False), (1060, (1400, 777), dict(obs_tail=("position", "record"), obs_scalars="raw"), True),
         (68, (3000, 1111), dict(obs_tail=("position",), obs_scalars="norm"), False), (1060, (700, 333), dict(layout="ppo"), False),
         (132, (1500, 500), dict(obs_tail=("position", "plan", "record")), True)]

for n, launches, kw, f32 in CASES:
    for dyn in (True, False):
        table = helpers.plan_table(3, dyn, "dense_train" if dyn else "p1")
        env = BatchedDMPEnv(3, dyn, n, plans=table.reshape(len(table), 26, 26), seed=77, obs_dtype=torch.float32 if f32 else torch.float64, **kw)
        orc = helpers.oracle().OracleBatch(3, dyn, n, table, seed=77)
        orc.configure(obs_norm={None: dyn, "raw": False, "norm": True}[env.obs_scalars], frame=env.frame_value, tail=env.obs_tail)
        cast = (lambda a: a.astype(np.float32)) if f32 else (lambda a: a)
        assert env.reset().cpu().numpy().tobytes() == cast(orc.reset()).tobytes()
        t0 = 0
        for TT in launches:
            og, rg, dg = env.rollout(TT)
            k = _lib.lib().snac_last_kernel().decode()
            oc, rc, dc = orc.rollout(TT, t0=t0, nthreads=16)
            assert og.cpu().numpy().tobytes() == cast(oc).tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes() and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), (kw, dyn, TT)
            t0 += TT
        s, e = orc.stats(), env.episodic_stats()
        assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
        print("soak 3D %s n=%d %s %s: %s ticks on %s, %d episodes: identical to the oracle" % ("dyn" if dyn else "sta", n, "f32" if f32 else "f64", kw, "+".join(map(str, launches)), k, e["episodes"]), flush=True)
