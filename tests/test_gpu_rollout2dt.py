"""GPU: the time-parallel 2D rollout kernel (k_rollout2dt, round 4: one wavefront per env, lane = tick, blocks of 4 (up to 1024
envs) or 8 stepper waves and as many writer waves, compact rows through a double-buffered staging tile, the ticks to expand a queue
both kinds of wave draw from) against the CPU oracle, and
its borders in the dispatch: float64 rows up to 15 872 and from 16 385 to 19 456 envs, float32 rows below 30 720, batches whose
per-tick runs are not 16-byte pieces (odd N, unaligned outputs) up to 8192 -- the tile kernel / k_rollout2d beyond.  Every test
names the kernel it expects (snac_last_kernel).  Ragged blocks and blocks with idle waves, canonical and tile-major layouts,
tiny / odd tick counts, episodes that end by count_brick and by the time limit (several per chunk of 64 ticks, on and round the chunk
border), the `>` rule bits, both dtypes, the record outputs and explicit inputs, replay rings."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _kernel():
    from snac_amd import _lib

    return _lib.lib().snac_last_kernel().decode()


def _pair(dyn, n, seed, total_step=None, obs_dtype=None, brick_gt=False, time_gt=False, tag=None):
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(2, dyn, tag or ("dense_train" if dyn else "p0"))
    env = BatchedDMPEnv(2, dyn, n, plans=table.reshape(len(table), 26, 26), seed=seed, env_id_base=91, total_step=total_step,
                        obs_dtype=obs_dtype or torch.float64, brick_gt=brick_gt, time_gt=time_gt)
    orc = helpers.oracle().OracleBatch(2, dyn, n, table, seed=seed, env_id_base=91)
    if total_step:
        orc.set_total_step(total_step)
    orc.set_rules(brick_gt, time_gt)
    o = orc.reset()
    assert env.reset().cpu().numpy().tobytes() == (o.astype(np.float32) if obs_dtype == torch.float32 else o).tobytes()
    return env, orc


def _compare(env, orc, T, t0, f32=False, kernel="k_rollout2dt", **kw):
    og, rg, dg = env.rollout(T, **kw)
    assert _kernel() == kernel
    oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16, **kw)
    assert og.cpu().numpy().tobytes() == (oc.astype(np.float32) if f32 else oc).tobytes(), "observations"
    assert rg.cpu().numpy().tobytes() == rc.tobytes(), "rewards"
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc), "done flags"


def _end_state(env, orc):
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(env.num_envs, -1), st["grid"])
    for name, key in (("count_brick", "cb"), ("count_step", "cs"), ("plan_idx", "plan_idx"), ("episode", "episode"), ("episode_return", "ep_return")):
        assert np.array_equal(getattr(env, name).cpu().numpy(), st[key]), name
    assert np.array_equal(env.need_reset.cpu().numpy().astype(np.uint8), st["need_reset"])
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
    assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()


@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("n", [1, 3, 13, 1001, 1024, 1026, 2046, 4100, 8191, 11260])
def test_batch_shapes_and_tick_counts(dyn, n):
    """n = 1 / 3 / 13 / 1001: blocks of 4 envs, the last one ragged, odd N (rows leave element by element); 1024: the last batch in
    blocks of 4; 1026 / 2046: blocks of 8, a last block of 2 / 6 envs and idle stepper waves (which still draw ticks to expand); 4100: a
    last block of 4; 8191: the largest odd batch of this kernel; 11 260: the last one below k_rollout2db's range.  Launches of 1, 2, 37 and 80 steps; the time limit of 60 ends an episode
    in every launch of 80."""
    env, orc = _pair(dyn, n, seed=5, total_step=60)
    t0 = 0
    for T in (1, 2, 37, 80):
        _compare(env, orc, T, t0)
        t0 += T
    _end_state(env, orc)


@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("total_step,time_gt", [(1, False), (5, False), (7, True), (63, False), (64, False), (65, True)])
def test_segments_inside_a_chunk(dyn, total_step, time_gt):
    """An episode that ends inside a chunk of 64 ticks splits it into segments (the reset, with its new plan row, happens in the
    wave's uniform state; the bricks dropped before it must not show in the windows behind it): time limits of 1 (every lane its own
    episode), 5 and 7 (a dozen segments per chunk), and 63 / 64 / 65 (ends on, just before and just behind the chunk border);
    launches of 1, 64, 65 and 200 ticks; both dtypes; explicit inputs on the last launch."""
    import torch

    for f32 in (False, True):
        env, orc = _pair(dyn, 37, seed=8, total_step=total_step, obs_dtype=torch.float32 if f32 else None, time_gt=time_gt)
        t0 = 0
        for T in (1, 64, 65, 200):
            _compare(env, orc, T, t0, f32)
            t0 += T
        rng = np.random.default_rng(total_step)
        acts = rng.integers(0, 5, size=(70, 37)).astype(np.int8)
        ks = rng.integers(1, 4, size=(70, 37)).astype(np.int8)
        _compare(env, orc, 70, t0, f32, actions=acts, step_size=ks)
        _end_state(env, orc)


@pytest.mark.parametrize("rules", [(False, False), (True, False), (False, True), (True, True)], ids=str)
def test_episodes_end_by_bricks_and_by_time(rules):
    """Dense dataset plans hold ~150-200 bricks: with a time limit of 3000 a fifth of the steps drop one and episodes end at
    count_brick >= (>) total_brick long before it; with 45 they end by time.  Sparse plans (total_brick at its floor of 30) end by
    bricks within a few chunks."""
    for tag, total_step, T in (("dense_train", 3000, 2500), ("dense_train", 45, 700), ("sparse_train", 600, 900)):
        env, orc = _pair(True, 72, seed=9, total_step=total_step, brick_gt=rules[0], time_gt=rules[1], tag=tag)
        _compare(env, orc, T, 0)
        _end_state(env, orc)
        assert env.episodic_stats()["episodes"] > 0


def test_a_whole_episode_of_float32_rows():
    import torch

    env, orc = _pair(True, 200, seed=2, obs_dtype=torch.float32)
    _compare(env, orc, 600, 0, f32=True)
    _compare(env, orc, 33, 600, f32=True)
    _end_state(env, orc)


BORDERS = [
    # n, float32 rows, kernel of the canonical output, kernel of an output that is not 16-byte aligned.  From 11 264 envs (float32: 15 360) to
    # 38 912 (45 056) the canonical rows are k_rollout2db's (round 5: blocks of 64 envs, of 128 from 16 384 / 16 385, of 256 above 32 768): tests/test_gpu_rollout2d_block.py
    (8191, False, "k_rollout2dt", "k_rollout2dt"),                  # odd N: element by element, still this kernel up to 8192
    (8193, False, "k_rollout", "k_rollout"),
    (8200, True, "k_rollout2dt", "k_rollout"),                      # whole pieces / an unaligned output of more than 8192 envs
    (11260, False, "k_rollout2dt", "k_rollout"),
    (11264, False, "k_rollout2db", "k_rollout"),
    (11266, False, "k_rollout2dt", "k_rollout"),                    # N % 4 != 0: not the block kernel's
    (15356, True, "k_rollout2dt", "k_rollout"),
    (15360, True, "k_rollout2db", "k_rollout"),
    (32768, False, "k_rollout2db", "k_rollout"),                    # blocks of 128 envs
    (32772, False, "k_rollout2db", "k_rollout"),                    # ... of 256 envs, up to 38 912 (float32: 45 056)
    (38916, False, "k_rollout2d", "k_rollout"),
    (45056, True, "k_rollout2db", "k_rollout"),
    (45060, True, "k_rollout2d", "k_rollout"),
]


def _tiled_kernel(n, f32, kern):
    return "k_rollout2dt" if kern == "k_rollout" else kern             # (8193 envs: tile-major runs are whole pieces whatever N)


@pytest.mark.parametrize("n,f32,kern,kern_unaligned", BORDERS, ids=lambda v: str(v))
def test_both_sides_of_every_border(n, f32, kern, kern_unaligned):
    """The same rows on either side of each dispatch border, canonical and tile-major ([env // 64, t, env % 64]), and into an output
    that is not 16-byte aligned; records and episodic sums of the three batches equal."""
    import torch

    dt = torch.float32 if f32 else torch.float64
    T = 70
    env, orc = _pair(True, n, seed=3, total_step=40, obs_dtype=dt)
    twin = env.fork(torch.arange(n, device=env.device))
    third = env.fork(torch.arange(n, device=env.device))
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    want = (oc.astype(np.float32) if f32 else oc).tobytes()
    og, rg, dg = env.rollout(T)
    assert _kernel() == kern
    assert og.cpu().numpy().tobytes() == want and rg.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
    ot, rt, dtt = twin.rollout(T, obs="tiled")
    assert _kernel() == _tiled_kernel(n, f32, kern)
    assert twin.untile(ot).cpu().numpy().tobytes() == want and torch.equal(rt, rg) and torch.equal(dtt, dg)
    raw = torch.empty(T * n * 51 + 1, dtype=dt, device=env.device)
    ou, ru, du = third.rollout(T, out=raw[1:].view(T, n, 51))
    assert _kernel() == kern_unaligned
    assert ou.data_ptr() % 16 != 0 and ou.cpu().numpy().tobytes() == want and torch.equal(ru, rg) and torch.equal(du, dg)
    _end_state(env, orc)
    assert torch.equal(env._hdr, twin._hdr) and torch.equal(env._hdr, third._hdr) and torch.equal(env._grid, third._grid)
    assert torch.equal(env._stats, twin._stats) and torch.equal(env._stats, third._stats)


@pytest.mark.parametrize("n", [40, 9000])
def test_record_outputs_fed_back_as_explicit_inputs(n):
    """The record outputs (action taken, step size used, plan row, first-step flag) of a counter-RNG rollout, fed back as explicit
    inputs (the EXPL instantiation), reproduce observations, rewards and done flags -- which equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(2, True, "dense_train")
    T = 130
    a = BatchedDMPEnv(2, True, n, plans=table.reshape(len(table), 26, 26), seed=21, total_step=50)
    b = BatchedDMPEnv(2, True, n, plans=table.reshape(len(table), 26, 26), seed=21, total_step=50)
    orc = helpers.oracle().OracleBatch(2, True, n, table, seed=21, env_id_base=0)
    orc.set_total_step(50)
    orc.reset()
    a.reset()
    b.reset()
    rec = {"actions": torch.empty((T, n), dtype=torch.int8, device="cuda"), "step_size": torch.empty((T, n), dtype=torch.int8, device="cuda"),
           "plan_idx": torch.empty((T, n), dtype=torch.int16, device="cuda"), "first": torch.empty((T, n), dtype=torch.uint8, device="cuda")}
    oa, ra, da = a.rollout(T, record=rec)
    assert _kernel() == "k_rollout2dt"
    ob, rb, db = b.rollout(T, actions=rec["actions"], step_size=rec["step_size"])
    assert _kernel() == "k_rollout2dt"
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    assert oa.cpu().numpy().tobytes() == oc.tobytes() and ra.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(da.cpu().numpy().view(np.uint8), dc)
    first = rec["first"].cpu().numpy()
    done = da.cpu().numpy()
    assert first[0].all() and np.array_equal(first[1:], done[:-1].astype(np.uint8))   # auto-reset: a step opens an episode iff the last one ended one
    assert np.array_equal(rec["plan_idx"][-1].cpu().numpy(), a.plan_idx.cpu().numpy())   # an env is reset by its NEXT step
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid)


def test_replay_rings_filled_by_the_time_parallel_kernel():
    """ReplayRing.collect on a 2D batch whose rollouts run on k_rollout2dt in blocks of 8 envs (a ragged last block): the tick ring
    and the tile-major ring (launches that write at an offset of the ring and wrap) hold the same rows, records and samples; the last
    launch's rows equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    n = 4100
    table = helpers.plan_table(2, True, "dense_train")
    envs = [BatchedDMPEnv(2, True, n, plans=table.reshape(len(table), 26, 26), seed=12, total_step=40) for _ in range(2)]
    orc = helpers.oracle().OracleBatch(2, True, n, table, seed=12)
    orc.set_total_step(40)
    orc.reset()
    orc.rollout(13, t0=0, obs=None, nthreads=16)
    rings = []
    for e, layout in zip(envs, ("ticks", "tiled")):
        e.reset()
        e.rollout(13, obs=None)                                       # attach in mid-episode
        rings.append(ReplayRing(e, 100, layout=layout))
    t0 = 13
    for T in (70, 90, 100, 7):                                        # 64-tick chunks that straddle the ring's end
        for r in rings:
            r.collect(T)
            assert _kernel() == "k_rollout2dt"
        oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
        t0 += T
    a, b = rings
    for slot in range(100):
        assert torch.equal(a.obs_at(slot), b.obs_at(slot)), slot
    for name in ("reward", "done", "action", "step_size", "plan_idx", "first"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert a.head == b.head == (70 + 90 + 100 + 7) % 100
    for i in range(7):
        assert a.obs_at((a.head - 7 + i) % 100).cpu().numpy().tobytes() == oc[i].tobytes(), i
    ga, gb = torch.Generator(device="cuda"), torch.Generator(device="cuda")
    ga.manual_seed(3); gb.manual_seed(3)
    sa, sb = a.sample(300, generator=ga), b.sample(300, generator=gb)
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    assert torch.equal(envs[0]._hdr, envs[1]._hdr) and torch.equal(envs[0]._grid, envs[1]._grid)


VARIANTS = [
    dict(layout="lnet2d"),                                           # 51 + position, frame value 2, normalised scalars on a static plan
    dict(layout="ppo"),                                              # 451 values: window, raw counters, the 400 plan cells
    dict(obs_tail=("record",)),                                      # 59
    dict(obs_tail=("position", "plan", "record"), frame_value=2, obs_scalars="raw"),   # 461
]


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("kw", VARIANTS, ids=["lnet2d", "ppo", "record", "all"])
def test_layout_variants(kw, f32):
    """The observation layouts of the reference's env copies (snac_env_desc.frame_value / obs_scalars / obs_tail) on the time-parallel
    kernel: the steppers file the record's values with every compact row, the writer waves assemble rows of any length (the plan
    tail from a per-writer copy of each env's plan row, refilled when a tick's row differs -- time limit 4: every env starts over, on
    a new plan row, sixteen times per chunk).  Against the oracle configured the same way (launches of 1, 5, 70 and 130 ticks, explicit
    inputs), against the tile kernel's rows for a twin of the batch (an output that is not 16-byte aligned forces it) over a whole
    episode at the reference's own time limit, and the tile-major output."""
    import torch
    from snac_amd import BatchedDMPEnv

    dyn = kw.get("layout") != "lnet2d"
    n = 1000
    table = helpers.plan_table(2, dyn, "dense_train" if dyn else "p0")
    dt = torch.float32 if f32 else torch.float64
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    for total_step in (4, None):
        env = BatchedDMPEnv(2, dyn, n, plans=table.reshape(len(table), 26, 26), seed=4, total_step=total_step, obs_dtype=dt, **kw)
        orc = helpers.oracle().OracleBatch(2, dyn, n, table, seed=4)
        norm = {None: dyn, "raw": False, "norm": True}[env.obs_scalars]
        orc.configure(obs_norm=norm, frame=env.frame_value, tail=env.obs_tail)
        if total_step:
            orc.set_total_step(total_step)
        assert env.reset().cpu().numpy().tobytes() == cast(orc.reset()).tobytes()
        t0 = 0
        for T in (1, 5, 70, 130):
            og, rg, dg = env.rollout(T)
            assert _kernel() == "k_rollout2dt"
            oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
            assert og.cpu().numpy().tobytes() == cast(oc).tobytes(), ("observations", total_step, T)
            assert rg.cpu().numpy().tobytes() == rc.tobytes() and np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
            t0 += T
        rng = np.random.default_rng(1)
        acts, ks = rng.integers(0, 5, size=(70, n)).astype(np.int8), rng.integers(1, 4, size=(70, n)).astype(np.int8)
        og, rg, dg = env.rollout(70, actions=torch.from_numpy(acts).to(env.device), step_size=torch.from_numpy(ks).to(env.device))
        assert _kernel() == "k_rollout2dt"
        oc, rc, dc = orc.rollout(70, t0=t0, actions=acts, step_size=ks, nthreads=16)
        assert og.cpu().numpy().tobytes() == cast(oc).tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
        del og, oc
        _end_state(env, orc)
    # a whole episode against the tile kernel, on the device
    twin = env.fork(torch.arange(n, device=env.device))
    T = 600
    o1, r1, d1 = env.rollout(T)
    assert _kernel() == "k_rollout2dt"
    raw = torch.empty(T * n * env.obs_dim + 1, dtype=dt, device=env.device)
    o2, r2, d2 = twin.rollout(T, out=raw[1:].view(T, n, env.obs_dim))
    assert _kernel() == "k_rollout" and o2.data_ptr() % 16 != 0
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    assert torch.equal(env._hdr, twin._hdr) and torch.equal(env._grid, twin._grid)     # (a fork starts its episodic sums at zero)
    # tile-major output of the same layout
    tw2 = env.fork(torch.arange(n, device=env.device))
    ot, _, _ = env.rollout(70, obs="tiled")
    assert _kernel() == "k_rollout2dt"
    on, _, _ = tw2.rollout(70)
    assert torch.equal(env.untile(ot), on)


@pytest.mark.parametrize("n,kw,kern", [(6144, dict(obs_tail=("record",)), "k_rollout2dt"), (6148, dict(obs_tail=("record",)), "k_rollout2db"),
                                       (49152, dict(layout="ppo"), "k_rollout2dt"), (49156, dict(layout="ppo"), "k_rollout2d"),
                                       (1002, dict(layout="ppo"), "k_rollout")], ids=str)
def test_layout_variants_at_the_borders(n, kw, kern):
    """Short rows up to 6144 envs, rows with the plan tail up to 49 152, whole groups of four envs only: 20 ticks on either side
    against the oracle."""
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(2, True, "dense_train")
    env = BatchedDMPEnv(2, True, n, plans=table.reshape(len(table), 26, 26), seed=6, total_step=7, obs_dtype=torch.float32, **kw)
    orc = helpers.oracle().OracleBatch(2, True, n, table, seed=6)
    norm = {None: True, "raw": False, "norm": True}[env.obs_scalars]
    orc.configure(obs_norm=norm, frame=env.frame_value, tail=env.obs_tail)
    orc.set_total_step(7)
    assert env.reset().cpu().numpy().tobytes() == orc.reset().astype(np.float32).tobytes()
    og, rg, dg = env.rollout(20)
    assert _kernel() == kern
    oc, rc, dc = orc.rollout(20, t0=0, nthreads=16)
    assert og.cpu().numpy().tobytes() == oc.astype(np.float32).tobytes()
    assert rg.cpu().numpy().tobytes() == rc.tobytes() and np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
    _end_state(env, orc)


@pytest.mark.parametrize("kind,kw", [(2, dict(layout="ppo")), (2, dict(obs_tail=("record",))), (1, dict(layout="ppo"))], ids=str)
def test_replay_rings_of_layout_variants(kind, kw):
    """Rings that collect rows of a layout variant through the time-parallel kernels (launches that write at an offset of a tile-major
    ring and wrap): the tick ring and the tile-major ring hold the same rows and records; the last launch's rows equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    n = 1000
    table = helpers.plan_table(kind, True, "sin_train" if kind == 1 else "dense_train")
    full = table.reshape(len(table), 30) if kind == 1 else table.reshape(len(table), 26, 26)
    envs = [BatchedDMPEnv(kind, True, n, plans=full, seed=12, total_step=40, **kw) for _ in range(2)]
    orc = helpers.oracle().OracleBatch(kind, True, n, table, seed=12)
    orc.configure(obs_norm={None: True, "raw": False, "norm": True}[envs[0].obs_scalars], frame=envs[0].frame_value, tail=envs[0].obs_tail)
    orc.set_total_step(40)
    orc.reset()
    rings = []
    for e, layout in zip(envs, ("ticks", "tiled")):
        e.reset()
        rings.append(ReplayRing(e, 100, layout=layout))
    t0 = 0
    for T in (70, 90, 100, 7):
        for r in rings:
            r.collect(T)
            assert _kernel() == ("k_rollout1dt" if kind == 1 else "k_rollout2dt")
        oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
        t0 += T
    a, b = rings
    for slot in range(100):
        assert torch.equal(a.obs_at(slot), b.obs_at(slot)), slot
    for name in ("reward", "done", "action", "step_size", "plan_idx", "first"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for i in range(7):
        assert a.obs_at((a.head - 7 + i) % 100).cpu().numpy().tobytes() == oc[i].tobytes(), i
