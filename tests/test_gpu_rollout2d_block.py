"""GPU: the block-cooperative 2D rollout kernel (k_rollout2db, round 5: per 64 envs a stepper wave -- lane = env, the boards its alone --
that publishes each tick's window codes, scalar slots, reward and done, and eight writer waves that assemble and store the rows; blocks of
64 envs with one stepper below 16 384 envs, of 128 envs with two steppers up to 32 768, of 256 envs with four above) against the CPU oracle.
It takes the canonical 2D rollouts of 11 264 .. 38 912 envs (float32 rows: 15 360 .. 45 056; N % 4 = 0, 16-byte aligned output) that write every row: full
blocks and ragged last blocks (a last stepper without envs, a last writer with 4 rows), float64 and float32 rows, dataset and static plans,
[T][N][D] and tile-major outputs, launches of 1 / 2 / 37 steps, explicit actions / step sizes, the `>` rule bits, time limits of 1 .. 3,
the record outputs -- and, bit for bit, what the tile kernel writes for the same batch (an unaligned output selects it).  The layout variants
without the plan tail (rows of 51 .. 61 values: the L-Net rows with frame cells 2 and normalised scalar slots, raw slots, position and
record tails) take the same kernel from 6148 to 32 768 envs (blocks of 128 envs from 16 388)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

N1 = 12288                                                        # blocks of 64 envs
N2 = 16384                                                        # blocks of 128 envs


def _kernel():
    from snac_amd import _lib

    return _lib.lib().snac_last_kernel().decode()


def _pair(dyn, n, seed, tag=None, total_step=None, f32=False, brick_gt=False, time_gt=False, base=0):
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(2, dyn, tag or ("dense_train" if dyn else "p1"))
    env = BatchedDMPEnv(2, dyn, n, plans=table.reshape(len(table), 26, 26), seed=seed, env_id_base=base, total_step=total_step,
                        obs_dtype=torch.float32 if f32 else torch.float64, brick_gt=brick_gt, time_gt=time_gt)
    orc = helpers.oracle().OracleBatch(2, dyn, n, table, seed=seed, env_id_base=base)
    if total_step:
        orc.set_total_step(total_step)
    orc.set_rules(brick_gt, time_gt)
    o = orc.reset()
    assert env.reset().cpu().numpy().tobytes() == (o.astype(np.float32) if f32 else o).tobytes()
    return env, orc


def _compare(env, orc, T, t0, f32=False, actions=None, step_size=None):
    import torch

    a = None if actions is None else torch.from_numpy(actions).to(env.device)
    k = None if step_size is None else torch.from_numpy(step_size).to(env.device)
    og, rg, dg = env.rollout(T, actions=a, step_size=k)
    assert _kernel() == "k_rollout2db"
    oc, rc, dc = orc.rollout(T, t0=t0, actions=actions, step_size=step_size, nthreads=16)
    want = oc.astype(np.float32) if f32 else oc
    got = og.cpu().numpy()
    if got.tobytes() != want.tobytes():
        bad = np.argwhere(got != want)
        raise AssertionError("observations: %d values differ, first at (t, env, value) %s: %r != %r" % (len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))
    assert rg.cpu().numpy().tobytes() == rc.tobytes(), "rewards"
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc), "done flags"


def _end_state(env, orc):
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
    assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(env.num_envs, -1), st["grid"])


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("n", [N1 + 3072, N1 + 3072 + 36, N2 + 128 + 36, N2 + 128 + 68, N2 + 4, 32768 + 256 + 36, 32768 + 256 + 200])
def test_blocks_dtypes_and_launch_lengths(dyn, n, f32):
    """15 360 envs: full blocks of 64; + 36: a last block of 36 envs (four full writer waves, one with 4 envs, three idle).  16 548 envs:
    blocks of 128 envs with a last block of 36 (its second stepper has no envs); + 32: a last block of 68 (the second stepper has 4);
    16 388: a last block of 4 envs; 33 060 / 33 224 envs: blocks of 256 envs (four steppers) with a last block of 36 (three steppers without
    envs) / of 200 (the fourth stepper has 8).  Launches of 1, 2 and 37 steps with a time limit of 30: every launch of 37 has envs that start over."""
    env, orc = _pair(dyn, n, seed=5, f32=f32, base=11, total_step=30)
    t0 = 0
    for T in (1, 2, 37):
        _compare(env, orc, T, t0, f32)
        t0 += T
    _end_state(env, orc)
    _compare(env, orc, 3, t0, f32)                                # the records written back by the launches above carry on


@pytest.mark.parametrize("n", [N1 + 36, N2 + 36])
@pytest.mark.parametrize("total_step,time_gt", [(1, False), (1, True), (2, False), (3, False)])
def test_envs_that_start_over_every_tick(total_step, time_gt, n):
    """A time limit of 1: every env starts over at every tick -- the stepper clears a board and fetches a plan row through the scalar cache
    for each of its 64 lanes, every tick."""
    env, orc = _pair(True, n, seed=9, total_step=total_step, time_gt=time_gt)
    rng = np.random.default_rng(total_step)
    acts = rng.choice(np.arange(5, dtype=np.int8), size=(12, n), p=[0.1, 0.1, 0.1, 0.1, 0.6])   # drops mostly
    _compare(env, orc, 12, 0, actions=acts)
    _compare(env, orc, 13, 12)
    _end_state(env, orc)


@pytest.mark.parametrize("rules", [(False, False), (True, False), (False, True), (True, True)], ids=str)
def test_episodes_end_by_bricks_and_by_time(rules):
    """Sparse plans (total_brick floored at 30) and drop-heavy explicit actions: episodes end at count_brick >= (>) total_brick and by the
    time limit 45 (> with the rule bit)."""
    n, T = N2 + 36, 120
    env, orc = _pair(True, n, seed=9, tag="sparse_train", total_step=45, brick_gt=rules[0], time_gt=rules[1])
    rng = np.random.default_rng(3)
    acts = rng.choice(np.arange(5, dtype=np.int8), size=(T, n), p=[0.1, 0.1, 0.1, 0.1, 0.6])
    _compare(env, orc, T, 0, actions=acts)                        # explicit actions, counter-RNG step sizes
    _end_state(env, orc)
    assert env.episodic_stats()["episodes"] >= 2 * n


@pytest.mark.parametrize("n", [N1 + 36, N2 + 36])
def test_explicit_inputs_loaded_two_ticks_ahead(n):
    """actions only, step sizes only, both; out-of-range step sizes are clamped into {1, 2, 3}; launches of one and two steps have
    nothing (or one tick) to prefetch."""
    env, orc = _pair(True, n, seed=2, total_step=50)
    rng = np.random.default_rng(7)
    t0 = 0
    for T, use_a, use_k in ((1, True, True), (2, True, True), (23, True, False), (23, False, True), (40, True, True)):
        acts = rng.integers(0, 5, size=(T, n)).astype(np.int8) if use_a else None
        ks = rng.integers(0, 6, size=(T, n)).astype(np.int8) if use_k else None
        orc_k = None if ks is None else np.clip(ks, 1, 3)
        og, rg, dg = env.rollout(T, actions=acts, step_size=ks)
        assert _kernel() == "k_rollout2db"
        oc, rc, dc = orc.rollout(T, t0=t0, actions=acts, step_size=orc_k, nthreads=16)
        assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
        assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
        t0 += T
    _end_state(env, orc)


@pytest.mark.parametrize("n", [N1 + 36, N2 + 36 + 64])
@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
def test_tile_major_output_record_and_the_tile_kernel(f32, n):
    """rollout(obs="tiled") holds the same rows at [env // 64, t, env % 64] (a block of 128 envs writes two tiles); the record outputs
    (action, step size, plan row, first-step flag), the rows and the final records equal what the tile kernel gives for an identical
    batch -- which an output that is not 16-byte aligned selects."""
    import torch

    if f32:
        n += 3072                                                 # float32 rows take the block kernel from 15 360 envs
    T = 33
    dt = torch.float32 if f32 else torch.float64
    a, orc = _pair(True, n, seed=4, total_step=20, f32=f32)
    b = a.fork(torch.arange(n, device=a.device))
    kinds = {"actions": torch.int8, "step_size": torch.int8, "plan_idx": torch.int16, "first": torch.uint8}
    ra = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    rb = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    ot, rt, dtt = a.rollout(T, obs="tiled", record=ra)
    assert _kernel() == "k_rollout2db"
    raw = torch.empty(T * n * 51 + 1, dtype=dt, device=a.device)
    ob, rwb, db = b.rollout(T, out=raw[1:].view(T, n, 51), record=rb)
    assert ob.data_ptr() % 16 != 0 and _kernel() == "k_rollout"
    assert torch.equal(a.untile(ot), ob) and torch.equal(rt, rwb) and torch.equal(dtt, db)
    for k in kinds:
        assert torch.equal(ra[k], rb[k]), k
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    assert ob.cpu().numpy().tobytes() == (oc.astype(np.float32) if f32 else oc).tobytes()
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid) and torch.equal(a._stats, b._stats) and torch.equal(a._episode, b._episode)


def test_a_pending_reset_carried_into_the_next_launch():
    """A launch that ends on a done step leaves the flag in the header: the next launch starts the env over at its first tick."""
    env, orc = _pair(True, N2, seed=6, total_step=5)
    t0 = 0
    for T in (5, 1, 4, 5, 7):                                     # launches that end exactly on the time limit, and ones that do not
        _compare(env, orc, T, t0)
        t0 += T
    _end_state(env, orc)


def test_a_whole_episode_and_more_against_the_oracle():
    """600 + 75 ticks of the dataset class at its own time limit: the boolean IoU kept incrementally over whole episodes, the plan rows
    fetched at every new episode."""
    env, orc = _pair(True, N2 + 36, seed=3)
    _compare(env, orc, 600, 0)
    _compare(env, orc, 75, 600)
    _end_state(env, orc)


def test_replay_rings_filled_by_the_block_kernel():
    """ReplayRing.collect on a 2D batch of block-kernel size: the tick ring and the tile-major ring (launches that write at an offset of
    the ring and wrap) hold the same rows, records and samples; the last launch's rows equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    n = N2 + 36
    table = helpers.plan_table(2, True, "dense_train")
    envs = [BatchedDMPEnv(2, True, n, plans=table.reshape(len(table), 26, 26), seed=12, total_step=40) for _ in range(2)]
    orc = helpers.oracle().OracleBatch(2, True, n, table, seed=12)
    orc.set_total_step(40)
    orc.reset()
    orc.rollout(13, t0=0, obs=None, nthreads=16)
    rings = []
    for e, layout in zip(envs, ("ticks", "tiled")):
        e.reset()
        e.rollout(13, obs=None)                                   # attach in mid-episode
        rings.append(ReplayRing(e, 48, layout=layout))
    t0 = 13
    for T in (20, 30, 48, 7):
        for r in rings:
            r.collect(T)
            assert _kernel() == "k_rollout2db"
        oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
        t0 += T
    a, b = rings
    for slot in range(48):
        assert torch.equal(a.obs_at(slot), b.obs_at(slot)), slot
    for name in ("reward", "done", "action", "step_size", "plan_idx", "first"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert a.head == b.head == (20 + 30 + 48 + 7) % 48
    for i in range(7):
        slot = (a.head - 7 + i) % 48
        assert a.obs_at(slot).cpu().numpy().tobytes() == oc[i].tobytes(), i


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("P", [2000, 32767])
def test_generated_plan_tables_of_any_size(P, f32):
    """Tables from generate_plans() (2000 rows; 32 767: the rows then miss the scalar cache): a time limit of 25, so every env picks a new row
    of the big table in every launch of 37 -- the stepper fetches it through the scalar cache; blocks of 128 envs, a ragged last block."""
    import torch
    from snac_amd import BatchedDMPEnv

    n = N2 + 3072 + 36
    env = BatchedDMPEnv(2, True, n, plans=np.zeros((P, 26, 26)), seed=8, total_step=25, obs_dtype=torch.float32 if f32 else torch.float64)
    env.generate_plans(0, P, sparse=False, seed=77, id_base=5)
    env._sync_plans_full()
    orc = helpers.oracle().OracleBatch(2, True, n, env.plans_full.reshape(P, -1).astype(np.int32), seed=8)
    orc.set_total_step(25)
    o = orc.reset()
    assert env.reset().cpu().numpy().tobytes() == (o.astype(np.float32) if f32 else o).tobytes()
    t0 = 0
    for T in (1, 37, 2):
        _compare(env, orc, T, t0, f32)
        t0 += T
    _end_state(env, orc)
    assert len(set(env.plan_idx.cpu().numpy().tolist())) > min(P, 400) // 2   # the batch really draws from all over the table


# ---- the layout variants without the plan tail (k_roll2dbv.hip)
def _vpair(dyn, n, seed, kw, total_step=None, f32=False, base=0):
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(2, dyn, "dense_train" if dyn else "p1")
    env = BatchedDMPEnv(2, dyn, n, plans=table.reshape(len(table), 26, 26), seed=seed, env_id_base=base, total_step=total_step,
                        obs_dtype=torch.float32 if f32 else torch.float64, **kw)
    orc = helpers.oracle().OracleBatch(2, dyn, n, table, seed=seed, env_id_base=base)
    if total_step:
        orc.set_total_step(total_step)
    orc.configure(obs_norm={None: dyn, "raw": False, "norm": True}[env.obs_scalars], frame=env.frame_value, tail=env.obs_tail)
    assert orc.obs_dim == env.obs_dim
    o = orc.reset()
    assert env.reset().cpu().numpy().tobytes() == (o.astype(np.float32) if f32 else o).tobytes()
    return env, orc


VARIANTS = [dict(layout="lnet2d"), dict(obs_tail=("record",)), dict(obs_tail=("position", "record"), obs_scalars="raw", frame_value=2), dict(obs_tail=("position",)),
            dict(obs_scalars="raw"), dict(obs_scalars="norm")]


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("n", [6148 + 36, N2 + 128 + 36])
@pytest.mark.parametrize("kw", VARIANTS, ids=["lnet2d", "record", "pos_record_raw_f2", "position", "raw", "norm"])
def test_variant_rows_without_the_plan_tail(kw, n, dyn, f32):
    """Rows of 51 / 53 / 59 / 61 values on blocks of 64 envs (6184 envs: a last block of 40) and of 128 envs (16 548: a last block of 36);
    launches of 1, 2 and 37 steps with a time limit of 30."""
    if kw == dict(obs_scalars="norm" if dyn else "raw"):
        pytest.skip("the canonical layout of this class")
    env, orc = _vpair(dyn, n, 5, kw, total_step=30, f32=f32, base=11)
    t0 = 0
    for T in (1, 2, 37):
        _compare(env, orc, T, t0, f32)
        t0 += T
    _end_state(env, orc)
    _compare(env, orc, 3, t0, f32)


@pytest.mark.parametrize("n,kernel", [(6144, "k_rollout2dt"), (6148, "k_rollout2db"), (16384, "k_rollout2db"), (32768, "k_rollout2db"), (32772, "k_rollout2d")])
def test_variant_rows_either_side_of_the_thresholds(n, kernel):
    import torch

    env, orc = _vpair(True, n, 6, dict(obs_tail=("record",)), total_step=7)
    og, rg, dg = env.rollout(12)
    assert _kernel() == kernel
    oc, rc, dc = orc.rollout(12, t0=0, nthreads=16)
    assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
    _end_state(env, orc)


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
def test_variant_rows_tile_major_explicit_inputs_and_the_tile_kernel(f32):
    import torch

    n, T = N2 + 100, 33
    dt = torch.float32 if f32 else torch.float64
    a, orc = _vpair(True, n, 4, dict(obs_tail=("position", "record"), obs_scalars="raw", frame_value=2), total_step=20, f32=f32)
    b = a.fork(torch.arange(n, device=a.device))
    rng = np.random.default_rng(5)
    acts, ks = rng.integers(0, 5, size=(T, n)).astype(np.int8), rng.integers(1, 4, size=(T, n)).astype(np.int8)
    ta, tk = torch.from_numpy(acts).cuda(), torch.from_numpy(ks).cuda()
    ot, rt, dtt = a.rollout(T, obs="tiled", actions=ta, step_size=tk)
    assert _kernel() == "k_rollout2db"
    raw = torch.empty(T * n * 61 + 1, dtype=dt, device=a.device)
    ob, rwb, db = b.rollout(T, out=raw[1:].view(T, n, 61), actions=ta, step_size=tk)
    assert ob.data_ptr() % 16 != 0 and _kernel() == "k_rollout"
    assert torch.equal(a.untile(ot), ob) and torch.equal(rt, rwb) and torch.equal(dtt, db)
    oc, rc, dc = orc.rollout(T, t0=0, actions=acts, step_size=ks, nthreads=16)
    assert ob.cpu().numpy().tobytes() == (oc.astype(np.float32) if f32 else oc).tobytes()
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid) and torch.equal(a._stats, b._stats) and torch.equal(a._episode, b._episode)
