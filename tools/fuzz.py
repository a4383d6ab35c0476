"""Randomised parity fuzz on the GPU box (not part of the test suite): random env kind, batch size (small, and around the
dispatch thresholds of the specialised kernels), time limit, rule bits, observation dtype and seed; then a random sequence of
operations -- fused rollouts (counter RNG or explicit inputs), per-tick step() with and without auto-reset, step_scalar, waves of
tree-search edges (gathered / scattered and in place; 2D also on node records), masked resets, and for batches of up to 256 envs steps through the
resident wavefronts (the mailbox) -- each compared with the CPU oracle bit for bit, the full state at the end.  Prints one line per trial; stops at the first difference.

    gpurun -- python tools/fuzz.py [trials] [seed]
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
from snac_amd import BatchedDMPEnv, trajmem  # noqa: E402

SIZES = [1, 2, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 100, 129, 255, 300, 1000, 2047, 2048, 2049, 3584, 3585, 4095, 4096, 4100, 8191, 8192, 8193, 8200, 11260, 11264, 11268, 15356, 15360, 15872, 15880, 16380, 16384, 16388, 16390, 17408, 19456, 19460,
         24576, 30716, 30720, 32764, 32768, 32772, 33000, 38912, 38916, 40956, 40960, 45052, 45056, 45060, 49152, 49153, 65536, 65540, 65600, 66000]


def same(a, b, what, ctx):
    if a.tobytes() != b.tobytes():
        bad = np.flatnonzero(a.reshape(-1) != b.reshape(-1))
        raise AssertionError("%s differs at %d of %d values (first flat index %d): %r" % (what, len(bad), a.size, bad[0] if len(bad) else -1, ctx))


def trial(rng, idx):
    dim = int(rng.integers(1, 4))
    dyn = bool(rng.integers(0, 2))
    n = int(rng.choice(SIZES)) if rng.random() < 0.7 else int(rng.integers(1, 600))
    if dim == 3 and n > 20000:
        n = int(rng.choice([16384, 16390, 8200]))                # the oracle's 3D rollouts are the slow part
    f32 = rng.random() < 0.25
    total_step = int(rng.choice([0, 0, 0, 23, 60, 150, 400]))
    bgt, tgt = bool(rng.integers(0, 2)) and rng.random() < 0.3, bool(rng.integers(0, 2)) and rng.random() < 0.3
    seed = int(rng.integers(0, 2 ** 62))
    base = int(rng.integers(0, 2 ** 40))
    tag = (("sin_train", "sin_val", "sin_test")[int(rng.integers(0, 3))] if dim == 1 else
           ("dense_train", "sparse_train", "dense_test", "sparse_val")[int(rng.integers(0, 4))]) if dyn else "p%d" % int(rng.integers(0, 3 if dim == 1 else 2))
    table = helpers.plan_table(dim, dyn, tag)
    full = table.reshape(len(table), 30) if dim == 1 else table.reshape(len(table), 26, 26)
    ctx = dict(trial=idx, dim=dim, dyn=dyn, n=n, f32=f32, total_step=total_step, brick_gt=bgt, time_gt=tgt, seed=seed, base=base, plans=tag)
    lay = {}
    if rng.random() < 0.25:                                       # an observation-layout variant: flags of the kernels
        tails = [t for t in ("position", "plan", "record") if rng.random() < 0.5]
        if n > 5000 and "plan" in tails:
            if dim in (2, 3) and rng.random() < 0.5:
                pass                                               # 451-value rows at N >= 65 536 take k_rollout2d's row groups, 3D ones from 10 240 envs k_rollout3db's plan rows: a few ticks of them
            else:
                tails.remove("plan")                              # 400 more values per row: keep the oracle's share small
        lay = dict(obs_tail=tuple(tails), frame_value=int(rng.choice([-1, 2])) if dim != 3 else -1, obs_scalars=str(rng.choice(["raw", "norm"])))
    ctx["layout"] = lay
    env = BatchedDMPEnv(dim, dyn, n, plans=full, seed=seed, env_id_base=base, total_step=total_step or None,
                        obs_dtype=torch.float32 if f32 else torch.float64, brick_gt=bgt, time_gt=tgt, **lay)
    orc = helpers.oracle().OracleBatch(dim, dyn, n, table, seed=seed, env_id_base=base)
    if lay:
        orc.configure(obs_norm={None: dyn, "raw": False, "norm": True}[env.obs_scalars], frame=env.frame_value, tail=env.obs_tail)
        assert orc.obs_dim == env.obs_dim
    if total_step:
        orc.set_total_step(total_step)
    orc.set_rules(bgt, tgt)
    cast = (lambda a: a.astype(np.float32)) if f32 else (lambda a: a)
    same(env.reset().cpu().numpy(), cast(orc.reset()), "reset obs", ctx)
    A = helpers.DIMS[dim]["A"]
    budget = 3_000_000 if dim == 3 else 6_000_000                # env-steps of oracle work per trial
    if lay and "plan" in lay.get("obs_tail", ()) and n > 5000:
        budget = 300_000
    t = 0
    ops = []
    for _ in range(int(rng.integers(2, 7))):
        op = str(rng.choice(["rollout", "rollout", "rollout_x", "steps", "steps_x", "scalar", "edges", "reset"] + (["mailbox", "mailbox"] if n <= 256 else [])))
        ops.append(op)
        ctx["ops"] = ops
        if op in ("rollout", "rollout_x"):
            T = int(min(rng.integers(1, 400), max(1, budget // n)))
            a = k = None
            if op == "rollout_x":
                bias = rng.random()
                a = np.where(rng.random((T, n)) < bias, A - 1, rng.integers(0, A, size=(T, n))).astype(np.int8)
                k = rng.integers(1, 4, size=(T, n)).astype(np.int8)
            tiled = rng.random() < 0.3                                # the tile-major trajectory layout holds the same rows
            out = None
            if rng.random() < 0.3:                                    # the observations into trajectory memory (snac_traj_alloc)
                shape = ((n + 63) // 64, T, 64, env.obs_dim) if tiled else (T, n, env.obs_dim)
                out = trajmem.traj_empty(shape, env.obs_dtype, env.device)
                ops[-1] += "/vmm"
            og, rg, dg = env.rollout(T, actions=None if a is None else torch.from_numpy(a), step_size=None if k is None else torch.from_numpy(k),
                                     obs="tiled" if tiled else "all", out=out)
            if tiled:
                og = env.untile(og)
                ops[-1] += "/tiled"
            oc, rc, dc = orc.rollout(T, t0=t, actions=a, step_size=k, nthreads=16)
            same(og.cpu().numpy(), cast(oc), "rollout obs", ctx)
            same(rg.cpu().numpy(), rc, "rollout reward", ctx)
            same(dg.cpu().numpy().view(np.uint8), dc, "rollout done", ctx)
            t += T
        elif op in ("steps", "steps_x"):
            m = int(min(rng.integers(1, 60), max(1, budget // (4 * n))))
            ar = bool(rng.integers(0, 2))
            for _ in range(m):
                a = k = None
                if op == "steps_x":
                    a = rng.integers(0, A, size=n).astype(np.int8)
                    k = rng.integers(1, 4, size=n).astype(np.int8)
                og, rg, dg = env.step(None if a is None else torch.from_numpy(a), None if k is None else torch.from_numpy(k), auto_reset=ar)
                oc, rc, dc = orc.step(t, a, k, auto_reset=ar, nthreads=16)
                same(og.cpu().numpy(), cast(oc), "step obs", ctx)
                same(rg.cpu().numpy(), rc, "step reward", ctx)
                same(dg.cpu().numpy().view(np.uint8), dc, "step done", ctx)
                t += 1
        elif op == "scalar":
            for _ in range(int(rng.integers(1, 20))):
                a, k = int(rng.integers(0, A)), int(rng.integers(1, 4))
                og = env.step_scalar(a, k, auto_reset=True)
                oc, _, _ = orc.step(t, np.full(n, a, np.int8), np.full(n, k, np.int8), auto_reset=True, nthreads=16)
                same(og.cpu().numpy(), cast(oc), "step_scalar obs", ctx)
                t += 1
        elif op == "edges" and n >= 2:
            for w in range(int(rng.integers(1, 4))):
                n_dst = int(rng.integers(1, max(2, n // 2)))
                dst = rng.choice(n, n_dst, replace=False).astype(np.int32)
                free = np.setdiff1d(np.arange(n), dst)
                inplace = rng.random(n_dst) < 0.3
                src = np.where(inplace | (len(free) == 0), dst, rng.choice(free if len(free) else dst, n_dst)).astype(np.int32)
                acts = rng.integers(0, A, n_dst).astype(np.int8)
                ks = rng.integers(1, 4, n_dst).astype(np.int8) if rng.random() < 0.5 else None
                if dim == 2 and not lay and rng.random() < 0.5:      # the same wave on node records (one 128-byte record per node): pack, step, unpack
                    from snac_amd import NodePool2D

                    pool = NodePool2D(env, n)
                    pool.load()
                    o, r, d = pool.transition(acts, ks, src, dst, t=w)
                    pool.store()
                    ops[-1] = "edges/nodes"
                else:
                    o, r, d = env.transition(acts, ks, src, dst, t=w)
                oo, ro, do = orc.transition(acts, ks, src, dst, t=w)
                same(o.cpu().numpy(), cast(oo), "edge obs", ctx)
                same(r.cpu().numpy(), ro, "edge reward", ctx)
                same(d.cpu().numpy().view(np.uint8), do, "edge done", ctx)
        elif op == "mailbox":
            # the resident wavefront (snac_mailbox_step / _step_n: an env per lane), in between whatever else the trial does to the batch;
            # short idle times so that waves also leave and come back inside a trial
            rows = env.mailbox_open(idle_us=int(rng.choice([30, 200, 2000]))).numpy()
            rew, don = env.mailbox_outputs()
            for _ in range(int(rng.integers(1, 40))):
                a = rng.integers(0, A, size=n).astype(np.int8)
                k = rng.integers(1, 4, size=n).astype(np.int8)
                if n == 1 and rng.random() < 0.5:
                    env.mailbox_step(int(a[0]), int(k[0]))
                    single = True
                else:
                    env.mailbox_step_n(a, k)
                    single = False
                oc, rc, dc = orc.step(t, a, k, auto_reset=False, nthreads=16)
                same(rows.copy(), cast(oc), "mailbox rows", ctx)
                if not (single and "record" in lay.get("obs_tail", ())):      # (a single env with the record tail has them in its row)
                    same(rew.copy(), rc, "mailbox reward", ctx)
                    same(don.copy(), dc, "mailbox done", ctx)
                t += 1
                if rng.random() < 0.1:
                    time.sleep(0.001)                                 # long enough for a 30 / 200 us wave to leave
        elif op == "reset":
            mask = (rng.random(n) < 0.4).astype(np.uint8)
            pidx = rng.integers(0, len(table), n).astype(np.int16)
            same(env.reset(mask=mask, plan_idx=pidx).cpu().numpy(), cast(orc.reset(mask=mask, plan_idx=pidx.astype(np.int32))), "masked reset obs", ctx)
    st = orc.state()
    same(env.environment_memory().cpu().numpy().reshape(n, -1), st["grid"].astype(np.float64), "grid", ctx)
    for name, key in (("count_brick", "cb"), ("count_step", "cs"), ("plan_idx", "plan_idx"), ("episode", "episode"), ("episode_return", "ep_return"), ("total_brick", "tb")):
        same(getattr(env, name).cpu().numpy().astype(np.int64), st[key].astype(np.int64), name, ctx)
    same(env.iou().cpu().numpy(), orc.iou(), "iou", ctx)
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum())), ctx
    return ctx, t, e["episodes"]


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for i in range(trials):
        ctx, t, eps = trial(rng, i)
        print("fuzz %3d ok  %dD %-3s n=%-6d %s T=%-4d rules=%d%d  ticks=%-4d episodes=%-7d ops=%s" % (
            i, ctx["dim"], "dyn" if ctx["dyn"] else "sta", ctx["n"], "f32" if ctx["f32"] else "f64", ctx["total_step"], ctx["brick_gt"], ctx["time_gt"],
            t, eps, ",".join(ctx["ops"]) + (" layout=%s" % (ctx["layout"],) if ctx["layout"] else "")), flush=True)
    print("fuzz: %d trials identical to the oracle (seed %d, %.0f s)" % (trials, seed, time.time() - t0))


if __name__ == "__main__":
    main()
