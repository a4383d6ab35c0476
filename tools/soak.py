"""Long-run parity soak on the GPU box (not part of the test suite): many episodes deep, every observation / reward /
done of every env-step compared with the CPU oracle, all six env types.  Prints one line per env type.

    gpurun -- python tools/soak.py [envs] [ticks]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import helpers  # noqa: E402
from snac_amd import BatchedDMPEnv  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    chunk = 500
    nt = max(1, min(64, len(os.sched_getaffinity(0))))
    for dim, dyn in ((1, False), (1, True), (2, False), (2, True), (3, False), (3, True)):
        tag = ("sin_train" if dim == 1 else "dense_train") if dyn else "p0"
        table = helpers.plan_table(dim, dyn, tag)
        full = table.reshape(len(table), 30) if dim == 1 else table.reshape(len(table), 26, 26)
        env = BatchedDMPEnv(dim, dyn, n, plans=full, seed=2024, env_id_base=10**9)
        orc = helpers.oracle().OracleBatch(dim, dyn, n, table, seed=2024, env_id_base=10**9)
        assert env.reset().cpu().numpy().tobytes() == orc.reset().tobytes()
        buf = torch.empty((chunk, n, env.obs_dim), dtype=torch.float64, device=env.device)
        t0 = time.time()
        t = 0
        while t < total:
            og, rg, dg = env.rollout(chunk, out=buf)
            oc, rc, dc = orc.rollout(chunk, t0=t, nthreads=nt)
            assert og.cpu().numpy().tobytes() == oc.tobytes(), (dim, dyn, t)
            assert rg.cpu().numpy().tobytes() == rc.tobytes(), (dim, dyn, t)
            assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc), (dim, dyn, t)
            t += chunk
        s = orc.stats()
        e = env.episodic_stats()
        assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
        print("soak %dD %-7s %d envs x %d ticks = %.2e env-steps, %d episodes: every obs / reward / done identical to the oracle (%.0f s)" % (
            dim, "dynamic" if dyn else "static", n, total, n * total, e["episodes"], time.time() - t0), flush=True)
        # the per-tick API (single-step kernel, auto-reset) and tree-search edges on the same batch
        ticks = max(200, total // 10)
        for i in range(ticks):
            og, rg, dg = env.step(auto_reset=True)
            oc, rc, dc = orc.step(t + i, auto_reset=True, nthreads=nt)
            assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes(), (dim, dyn, "step", i)
            assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
        rng = np.random.default_rng(dim)
        waves = 200
        for w in range(waves):
            m = n // 2
            src = rng.integers(0, n // 2, m).astype(np.int32)
            dst = (n // 2 + rng.permutation(n // 2)[:m]).astype(np.int32)
            acts = rng.integers(0, env.num_actions, m).astype(np.int8)
            og, rg, dg = env.transition(acts, None, src, dst, t=w)
            oc, rc, dc = orc.transition(acts, None, src, dst, t=w)
            assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes(), (dim, dyn, "edge", w)
            assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
        s = orc.stats()
        e = env.episodic_stats()
        assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
        print("     + %d step() ticks with auto-reset and %d transition waves of %d edges: identical (%.0f s)" % (
            ticks, waves, n // 2, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
