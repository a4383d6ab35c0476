import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import helpers
from snac_amd import BatchedDMPEnv, _lib
for kind, n, T, kw in ((2, 128, 3000, dict(layout="ppo")), (2, 256, 3000, dict(obs_tail=("position", "record"), frame_value=2, obs_scalars="raw")),
                       (1, 256, 3000, dict(layout="ppo")), (1, 512, 3000, dict(layout="lnet1d")), (2, 128, 3000, dict(obs_tail=("position", "plan", "record")))):
    for dyn in (True, False):
        tag = ("sin_train" if kind == 1 else "dense_train") if dyn else "p1"
        table = helpers.plan_table(kind, dyn, tag)
        full = table.reshape(len(table), 30) if kind == 1 else table.reshape(len(table), 26, 26)
        env = BatchedDMPEnv(kind, dyn, n, plans=full, seed=77, **kw)
        orc = helpers.oracle().OracleBatch(kind, dyn, n, table, seed=77)
        orc.configure(obs_norm={None: dyn, "raw": False, "norm": True}[env.obs_scalars], frame=env.frame_value, tail=env.obs_tail)
        assert env.reset().cpu().numpy().tobytes() == orc.reset().tobytes()
        t0 = 0
        for TT in (T, 777):
            og, rg, dg = env.rollout(TT)
            k = _lib.lib().snac_last_kernel().decode()
            oc, rc, dc = orc.rollout(TT, t0=t0, nthreads=16)
            assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes() and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), (kind, kw, dyn, TT)
            t0 += TT
        s, e = orc.stats(), env.episodic_stats()
        assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
        print("soak %dD %s n=%d %s: %d + 777 ticks on %s, %d episodes: identical to the oracle" % (kind, "dyn" if dyn else "sta", n, kw, T, k, e["episodes"]))
