"""GPU: the block-cooperative 3D rollout kernel (k_rollout3db, round 3: 64 envs per block, one stepper wave with an env per lane,
eight writer waves that own the height maps, one barrier per tick, the stepper's reads patched for the tick they lag) against the
CPU oracle.  The kernel takes 3D rollouts of N >= 4096 envs ( N % 4 = 0, 16-byte aligned output) that write every observation: full
blocks and a ragged last block (down to one writer wave with 4 envs), float64 and float32 rows, static and dataset plans,
[T][N][D] and tile-major outputs, launches of 1 / 2 / 37 steps, explicit actions / step sizes, the `>` rule bits, time limits of
1 .. 3 (an env starts over every tick: the stepper then reads no map at all for it, or a map that is a tick behind), the record
outputs -- and, bit for bit, the rows k_rollout3d writes for the same batch (forced by an unaligned output)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

N0 = 6144


def _pair(dyn, n, seed, tag=None, total_step=None, obs_dtype=None, brick_gt=False, time_gt=False, base=0):
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(3, dyn, tag or ("dense_train" if dyn else "p1"))
    env = BatchedDMPEnv(3, dyn, n, plans=table.reshape(len(table), 26, 26), seed=seed, env_id_base=base, total_step=total_step,
                        obs_dtype=obs_dtype or torch.float64, brick_gt=brick_gt, time_gt=time_gt)
    orc = helpers.oracle().OracleBatch(3, dyn, n, table, seed=seed, env_id_base=base)
    if total_step:
        orc.set_total_step(total_step)
    orc.set_rules(brick_gt, time_gt)
    o = orc.reset()
    assert env.reset().cpu().numpy().tobytes() == (o.astype(np.float32) if obs_dtype == torch.float32 else o).tobytes()
    return env, orc


def _compare(env, orc, T, t0, f32=False, actions=None, step_size=None):
    import torch

    a = None if actions is None else torch.from_numpy(actions).to(env.device)
    k = None if step_size is None else torch.from_numpy(step_size).to(env.device)
    og, rg, dg = env.rollout(T, actions=a, step_size=k)
    oc, rc, dc = orc.rollout(T, t0=t0, actions=actions, step_size=step_size, nthreads=16)
    want = oc.astype(np.float32) if f32 else oc
    assert og.cpu().numpy().tobytes() == want.tobytes(), "observations"
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
@pytest.mark.parametrize("n", [N0, N0 + 36, N0 + 64 + 4])
def test_blocks_dtypes_and_launch_lengths(dyn, n, f32):
    """n = 6144: full blocks only; + 36: a last block of 36 envs (four full writer waves, one with 4 envs, three idle); + 68: a last
    block of 4 envs.  Launches of 1, 2 and 37 steps; random 3D agents box themselves in every ~22 steps, so every launch of 37 has
    envs that start over, some of them twice."""
    import torch

    env, orc = _pair(dyn, n, seed=5, obs_dtype=torch.float32 if f32 else None, base=11)
    t0 = 0
    for T in (1, 2, 37):
        _compare(env, orc, T, t0, f32)
        t0 += T
    _end_state(env, orc)
    _compare(env, orc, 3, t0, f32)                                # the records written back by the launches above carry on


@pytest.mark.parametrize("total_step,time_gt", [(1, False), (1, True), (2, False), (3, False)])
def test_envs_that_start_over_every_tick(total_step, time_gt):
    """A time limit of 1: every env starts over at every tick, so the stepper never reads a map (cells by coordinates only) and the
    writers clear every map every tick; 2 (or 1 with the `>` rule): an env's second step reads a map that may not be cleared yet
    and carries the cell of its first step only as a patch; 3: the third step is the first to read its own episode's map."""
    n = N0 + 36
    env, orc = _pair(True, n, seed=9, total_step=total_step, time_gt=time_gt)
    rng = np.random.default_rng(total_step)
    acts = rng.choice(np.arange(8, dtype=np.int8), size=(24, n), p=[0.05, 0.05, 0.05, 0.05, 0.2, 0.2, 0.2, 0.2])   # builds mostly
    _compare(env, orc, 24, 0, actions=acts)
    _compare(env, orc, 25, 24)
    _end_state(env, orc)


@pytest.mark.parametrize("rules", [(False, False), (True, False), (False, True), (True, True)], ids=str)
def test_episodes_end_by_bricks_boxed_in_and_by_time(rules):
    """Sparse plans and build-heavy explicit actions: episodes end at count_brick >= (>) total_brick, boxed in (-100) and by the
    time limit 45 (> with the rule bit)."""
    n, T = N0 + 36, 120
    env, orc = _pair(True, n, seed=9, tag="sparse_train", total_step=45, brick_gt=rules[0], time_gt=rules[1])
    rng = np.random.default_rng(3)
    acts = rng.choice(np.arange(8, dtype=np.int8), size=(T, n), p=[0.1, 0.1, 0.1, 0.1, 0.15, 0.15, 0.15, 0.15])
    _compare(env, orc, T, 0, actions=acts)                        # explicit actions, counter-RNG step sizes
    _end_state(env, orc)
    assert env.episodic_stats()["episodes"] > 2 * n


def test_explicit_inputs_loaded_two_ticks_ahead():
    """actions only, step sizes only, both; out-of-range step sizes are clamped into {1, 2, 3}; launches of one and two steps have
    nothing (or one tick) to prefetch."""
    n = N0 + 36
    env, orc = _pair(True, n, seed=2, total_step=50)
    rng = np.random.default_rng(7)
    t0 = 0
    for T, use_a, use_k in ((1, True, True), (2, True, True), (23, True, False), (23, False, True), (40, True, True)):
        acts = rng.integers(0, 8, size=(T, n)).astype(np.int8) if use_a else None
        ks = rng.integers(0, 6, size=(T, n)).astype(np.int8) if use_k else None
        orc_k = None if ks is None else np.clip(ks, 1, 3)
        og, rg, dg = env.rollout(T, actions=acts, step_size=ks)
        oc, rc, dc = orc.rollout(T, t0=t0, actions=acts, step_size=orc_k, nthreads=16)
        assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
        assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
        t0 += T
    _end_state(env, orc)


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
def test_tile_major_output_record_and_the_eight_env_kernel(f32):
    """rollout(obs="tiled") holds the same rows at [env // 64, t, env % 64]; the record outputs (action, step size, plan row,
    first-step flag), the rows and the final records equal what k_rollout3d (8 envs per wave) gives for an identical batch -- which
    an output that is not 16-byte aligned selects."""
    import torch

    n, T = N0 + 36, 33
    dt = torch.float32 if f32 else torch.float64
    a, orc = _pair(True, n, seed=4, total_step=20, obs_dtype=dt)
    b = a.fork(torch.arange(n, device=a.device))
    kinds = {"actions": torch.int8, "step_size": torch.int8, "plan_idx": torch.int16, "first": torch.uint8}
    ra = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    rb = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    ot, rt, dtt = a.rollout(T, obs="tiled", record=ra)
    raw = torch.empty(T * n * 51 + 1, dtype=dt, device=a.device)
    ob, rwb, db = b.rollout(T, out=raw[1:].view(T, n, 51), record=rb)
    assert ob.data_ptr() % 16 != 0
    assert torch.equal(a.untile(ot), ob) and torch.equal(rt, rwb) and torch.equal(dtt, db)
    for k in kinds:
        assert torch.equal(ra[k], rb[k]), k
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    assert ob.cpu().numpy().tobytes() == (oc.astype(np.float32) if f32 else oc).tobytes()
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid) and torch.equal(a._stats, b._stats) and torch.equal(a._episode, b._episode)


def test_a_pending_reset_carried_into_the_next_launch():
    """A launch that ends on a done step leaves the flag in the header: the next launch starts the env over at its first tick (the
    stepper applies the scalars before the loop, the writers clear the map at tick 0)."""
    n = N0
    env, orc = _pair(True, n, seed=6, total_step=5)
    t0 = 0
    for T in (5, 1, 4, 5, 7):                                     # launches that end exactly on the time limit, and ones that do not
        _compare(env, orc, T, t0)
        t0 += T
    _end_state(env, orc)


def test_replay_rings_filled_by_the_block_kernel():
    """ReplayRing.collect on a 3D batch of block-kernel size: the tick ring ([ring_ticks][N][51] rows + record outputs) and the
    tile-major ring (snac_rollout_tiled: a launch writes its ticks at an offset of the ring, wrapping) hold the same rows, records
    and samples; the tick ring's rows equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    n = N0 + 36
    table = helpers.plan_table(3, True, "dense_train")
    envs = [BatchedDMPEnv(3, True, n, plans=table.reshape(len(table), 26, 26), seed=12, total_step=40) for _ in range(2)]
    orc = helpers.oracle().OracleBatch(3, True, n, table, seed=12)
    orc.set_total_step(40)
    orc.reset()
    orc.rollout(13, t0=0, obs=None, nthreads=16)
    rings = []
    for e, layout in zip(envs, ("ticks", "tiled")):
        e.reset()
        e.rollout(13, obs=None)                                       # attach in mid-episode
        rings.append(ReplayRing(e, 48, layout=layout))
    t0 = 13
    for T in (20, 30, 48, 7):
        for r in rings:
            r.collect(T)
        oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
        t0 += T
    a, b = rings
    for slot in range(48):
        assert torch.equal(a.obs_at(slot), b.obs_at(slot)), slot
    for name in ("reward", "done", "action", "step_size", "plan_idx", "first"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    # the last launch wrote 7 ticks: its rows are the oracle's last 7
    assert a.head == b.head == (20 + 30 + 48 + 7) % 48                # the ring slot after the last written one
    for i in range(7):
        slot = (a.head - 7 + i) % 48
        assert a.obs_at(slot).cpu().numpy().tobytes() == oc[i].tobytes(), i
    ga, gb = torch.Generator(device="cuda"), torch.Generator(device="cuda")
    ga.manual_seed(3); gb.manual_seed(3)
    sa, sb = a.sample(300, generator=ga), b.sample(300, generator=gb)
    assert all(torch.equal(sa[k], sb[k]) for k in sa)


def test_header_total_brick_survives_a_reset_onto_the_same_plan_in_every_3d_kernel():
    """A static batch whose headers carry a total_brick other than the plan row's (set_plan_row(update_tb=True) after reset,
    import_states with total_brick, an old snapshot): K::reset keeps it when an env starts over on the SAME plan row.  Round 3's
    k_rollout3d / k_rollout3db reloaded it from the table, so results depended on which kernel a call was dispatched to: obs="all"
    (the block kernel), an unaligned output (the eight-env kernel) and obs=None (the tile kernel) must agree."""
    import torch
    from snac_amd import _lib

    n, T = 8192, 90
    env, _ = _pair(False, n, seed=3, total_step=40)
    twins = [env.fork(torch.arange(n, device=env.device)) for _ in range(2)]
    for e in [env] + twins:
        e._hdr.view(torch.int16)[:, 4] = 23                        # total_brick 23 instead of the plan's 360
    o1, r1, d1 = env.rollout(T)
    assert _lib.lib().snac_last_kernel() == b"k_rollout3db"
    raw = torch.empty(T * n * 51 + 1, dtype=torch.float64, device=env.device)
    o2, r2, d2 = twins[0].rollout(T, out=raw[1:].view(T, n, 51))
    assert _lib.lib().snac_last_kernel() == b"k_rollout3d"
    _, r3, d3 = twins[1].rollout(T, obs=None)
    assert _lib.lib().snac_last_kernel() == b"k_rollout"
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(r1, r3) and torch.equal(d1, d3)
    for tw in twins:
        assert torch.equal(env._hdr, tw._hdr) and torch.equal(env._stats, tw._stats) and torch.equal(env._grid, tw._grid)
    assert int(env.total_brick.min()) == 23 and int(env.episodic_stats()["episodes"]) > n
