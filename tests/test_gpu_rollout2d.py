"""GPU: the staged 2D rollout kernel (k_rollout2d, round 3: lane-per-env window extraction, rows transposed through an LDS
staging tile, 16-byte-per-lane stores; the plan table, its popcounts and total_brick in LDS; incremental boolean IoU; reciprocal
observation scalars; prefetched explicit inputs) against the CPU oracle.  The kernel takes 2D rollouts on tiles of 64 envs
(N >= 65 536) that write every observation: full tiles and a ragged last tile, float64 (two staged halves) and float32, static and
dataset plans, [T][N][D] and tile-major outputs, launches of 1 / 2 / 37 steps, explicit actions / step sizes, the `>` rule bits,
short time limits (many resets per launch), the record outputs -- and, bit for bit, the rows the tile kernel writes for the same
batch (forced by an unaligned output, which the 16-byte stores do not take)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

N0 = 65536


def _pair(dyn, n, seed, tag=None, total_step=None, obs_dtype=None, brick_gt=False, time_gt=False, base=0):
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(2, dyn, tag or ("dense_train" if dyn else "p0"))
    env = BatchedDMPEnv(2, dyn, n, plans=table.reshape(len(table), 26, 26), seed=seed, env_id_base=base, total_step=total_step,
                        obs_dtype=obs_dtype or torch.float64, brick_gt=brick_gt, time_gt=time_gt)
    orc = helpers.oracle().OracleBatch(2, dyn, n, table, seed=seed, env_id_base=base)
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


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("n", [N0, N0 + 36, N0 + 64 + 4])
def test_tiles_dtypes_and_launch_lengths(dyn, n, f32):
    """n = 65 536: full tiles only; + 36: a ragged last tile that ends inside the first staged half of a float64 tile; + 68: one that
    is a lone 4-env tile in a block of its own.  Launches of 1, 2 and 37 steps; time limit 30, so every env resets in every launch of 37."""
    import torch

    env, orc = _pair(dyn, n, seed=5, total_step=30, obs_dtype=torch.float32 if f32 else None, base=11)
    t0 = 0
    for T in (1, 2, 37):
        _compare(env, orc, T, t0, f32)
        t0 += T
    _end_state(env, orc)
    _compare(env, orc, 3, t0, f32)                                # the records written back by the launches above carry on


@pytest.mark.parametrize("n", [45056 + 36, 49152 + 4])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_the_staged_kernel_from_32769_envs(dyn, n):
    """Above 45 056 envs both row types take the staged kernel (round 4: from 32 769 / 32 768 envs; since round 5 the block kernel k_rollout2db
    has the batches up to 38 912 / 45 056 envs, tests/test_gpu_rollout2d_block.py; round 3: float64 only from 65 536) --
    float32 and float64 rows of the same batch against the oracle and against each other, a ragged last tile, the tile-major output."""
    import torch
    from snac_amd import _lib

    env, orc = _pair(dyn, n, seed=6, total_step=30, obs_dtype=torch.float32)
    ref, _ = _pair(dyn, n, seed=6, total_step=30)
    t0 = 0
    for T in (1, 37):
        og, rg, dg = env.rollout(T)
        assert _lib.lib().snac_last_kernel() == b"k_rollout2d"
        o2, r2, d2 = ref.rollout(T)
        assert _lib.lib().snac_last_kernel() == b"k_rollout2d"
        oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
        assert og.cpu().numpy().tobytes() == oc.astype(np.float32).tobytes() and o2.cpu().numpy().tobytes() == oc.tobytes()
        assert rg.cpu().numpy().tobytes() == rc.tobytes() and torch.equal(rg, r2) and torch.equal(dg, d2)
        t0 += T
    _end_state(env, orc)
    assert torch.equal(env._hdr, ref._hdr) and torch.equal(env._grid, ref._grid) and torch.equal(env._stats, ref._stats)
    twin = ref.fork(torch.arange(n, device=ref.device))
    ot, _, _ = ref.rollout(5, obs="tiled")
    assert _lib.lib().snac_last_kernel() == b"k_rollout2d"
    on, _, _ = twin.rollout(5)
    assert torch.equal(ref.untile(ot), on)


@pytest.mark.parametrize("rules", [(False, False), (True, False), (False, True), (True, True)], ids=str)
def test_episodes_end_by_bricks_and_by_time(rules):
    """Sparse plans (total_brick floored at 30) and drop-heavy explicit actions: episodes end at count_brick >= (>) total_brick
    within ~40 steps; the time limit 45 (> with the rule bit) ends the rest."""
    n, T = N0 + 36, 120
    env, orc = _pair(True, n, seed=9, tag="sparse_train", total_step=45, brick_gt=rules[0], time_gt=rules[1])
    rng = np.random.default_rng(3)
    acts = rng.choice(np.arange(5, dtype=np.int8), size=(T, n), p=[0.05, 0.05, 0.05, 0.05, 0.8])
    _compare(env, orc, T, 0, actions=acts)                        # explicit actions, counter-RNG step sizes
    _end_state(env, orc)
    e = env.episodic_stats()
    assert e["episodes"] > 2 * n


def test_explicit_inputs_prefetched_a_tick_ahead():
    """actions only, step sizes only, both; out-of-range step sizes are clamped into {1, 2, 3}; a launch of one step has nothing to prefetch."""
    n = N0 + 36
    env, orc = _pair(True, n, seed=2, total_step=50)
    rng = np.random.default_rng(7)
    t0 = 0
    for T, use_a, use_k in ((1, True, True), (23, True, False), (23, False, True), (40, True, True)):
        acts = rng.integers(0, 5, size=(T, n)).astype(np.int8) if use_a else None
        ks = rng.integers(0, 6, size=(T, n)).astype(np.int8) if use_k else None
        if ks is not None:
            orc_k = np.clip(ks, 1, 3)
        og, rg, dg = env.rollout(T, actions=acts, step_size=ks)
        oc, rc, dc = orc.rollout(T, t0=t0, actions=acts, step_size=None if ks is None else orc_k, nthreads=16)
        assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
        assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
        t0 += T
    _end_state(env, orc)


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
def test_tile_major_output_and_record(f32):
    """rollout(obs="tiled") holds the same rows at [env // 64, t, env % 64]; the record outputs (action, step size, plan row,
    first-step flag) equal what the tile kernel records for an identical batch."""
    import torch
    from snac_amd import BatchedDMPEnv

    n, T = N0 + 36, 33
    dt = torch.float32 if f32 else torch.float64
    a, orc = _pair(True, n, seed=4, total_step=20, obs_dtype=dt)
    b = a.fork(torch.arange(n, device=a.device))
    kinds = {"actions": torch.int8, "step_size": torch.int8, "plan_idx": torch.int16, "first": torch.uint8}
    ra = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    rb = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    ot, rt, dtt = a.rollout(T, obs="tiled", record=ra)
    # the same batch through the tile kernel: an output that is not 16-byte aligned
    raw = torch.empty(T * n * 51 + 1, dtype=dt, device=a.device)
    ob, rwb, db = b.rollout(T, out=raw[1:].view(T, n, 51), record=rb)
    assert ob.data_ptr() % 16 != 0
    assert torch.equal(a.untile(ot), ob) and torch.equal(rt, rwb) and torch.equal(dtt, db)
    for k in kinds:
        assert torch.equal(ra[k], rb[k]), k
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    assert ob.cpu().numpy().tobytes() == (oc.astype(np.float32) if f32 else oc).tobytes()
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid) and torch.equal(a._stats, b._stats) and torch.equal(a._episode, b._episode)


def test_header_total_brick_survives_a_reset_onto_the_same_plan():
    """A static batch whose headers carry a total_brick other than the plan row's (import_states with total_brick): the tile kernel
    keeps it across auto-resets onto the same row -- so does the staged kernel, whose plan metadata lies in LDS."""
    import torch

    n = N0
    env, orc = _pair(False, n, seed=3, total_step=25)
    other = env.fork(torch.arange(n, device=env.device))
    for e in (env, other):
        e._hdr.view(torch.int16)[:, 4] = 17                        # total_brick 17 instead of the plan's
    o1, r1, d1 = env.rollout(60)
    raw = torch.empty(60 * n * 51 + 1, dtype=torch.float64, device=env.device)
    o2, r2, d2 = other.rollout(60, out=raw[1:].view(60, n, 51))
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    assert torch.equal(env._hdr, other._hdr) and torch.equal(env._stats, other._stats)
    assert int(env.total_brick.min()) == 17


def _generated(n, P, seed, total_step=None, obs_dtype=None, sparse=False):
    """A batch and its oracle twin on a table of P freshly generated plans (snac_make_plans; the table the oracle gets is read back
    from the device, the generator itself is pinned in tests/test_plan_generators.py)."""
    import torch
    from snac_amd import BatchedDMPEnv

    env = BatchedDMPEnv(2, True, n, plans=np.zeros((P, 26, 26)), seed=seed, total_step=total_step, obs_dtype=obs_dtype or torch.float64)
    env.generate_plans(0, P, sparse=sparse, seed=77, id_base=5)
    env._sync_plans_full()
    orc = helpers.oracle().OracleBatch(2, True, n, env.plans_full.reshape(P, -1).astype(np.int32), seed=seed)
    if total_step:
        orc.set_total_step(total_step)
    o = orc.reset()
    assert env.reset().cpu().numpy().tobytes() == (o.astype(np.float32) if obs_dtype == torch.float32 else o).tobytes()
    return env, orc


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("P", [513, 2000, 32767])
def test_plan_tables_beyond_the_lds_table_stay_on_the_staged_kernel(P, f32):
    """More than 512 plan rows (what generate_plans() is for) no longer fall back to the tile kernel: each wave keeps its lanes' current
    plan rows in LDS and an env that starts over fetches its new row through the scalar cache.  Time limit 25: every env picks a new
    row of the big table in every launch of 37; a ragged last tile; both dtypes; explicit inputs on the same path."""
    import torch

    n = N0 + 36
    env, orc = _generated(n, P, seed=8, total_step=25, obs_dtype=torch.float32 if f32 else None)
    t0 = 0
    for T in (1, 37, 2):
        _compare(env, orc, T, t0, f32)
        t0 += T
    rng = np.random.default_rng(3)
    acts = rng.integers(0, 5, size=(30, n)).astype(np.int8)
    ks = rng.integers(1, 4, size=(30, n)).astype(np.int8)
    _compare(env, orc, 30, t0, f32, actions=acts, step_size=ks)
    _end_state(env, orc)
    assert len(set(env.plan_idx.cpu().numpy().tolist())) > min(P, 400) // 2   # the batch really draws from all over the table


def test_a_full_episode_length_on_a_generated_table_of_2000_plans():
    """The verdict's case: N = 65 536, 2000 generated plans, the reference's own time limit (600): 120 ticks in two launches."""
    env, orc = _generated(N0, 2000, seed=2)
    _compare(env, orc, 100, 0)
    _compare(env, orc, 20, 100)
    _end_state(env, orc)


def _last_kernel():
    from snac_amd import _lib

    return _lib.lib().snac_last_kernel().decode()


VARIANTS = [
    dict(layout="lnet2d"),                                           # 51 + position, frame value 2, normalised scalars on a static plan
    dict(layout="ppo"),                                              # 451 values: window, raw counters, the 400 plan cells
    dict(obs_tail=("record",)),                                      # 59
    dict(obs_tail=("position", "plan", "record"), frame_value=2, obs_scalars="raw"),   # 461
]


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("kw", VARIANTS, ids=["lnet2d", "ppo", "record", "all"])
def test_layout_variants_on_the_staged_kernel(kw, f32):
    """The observation layouts of the reference's env copies (frame value 2, raw / normalised counters, position / plan / record
    tails: snac_env_desc.frame_value / obs_scalars / obs_tail) no longer fall back to the tile kernel at N >= 65 536: rows of any
    length leave through the staging tile in groups of envs.  Against the oracle configured the same way -- a ragged last tile, time
    limit 4 (every env starts over, on a new plan row, in every launch of 5), explicit inputs -- and, for a longer launch, against the
    tile kernel's rows for a twin of the batch (an output that is not 16-byte aligned forces it), compared on the device."""
    import torch
    from snac_amd import BatchedDMPEnv

    dyn = kw.get("layout") != "lnet2d"
    n = N0 + 36
    table = helpers.plan_table(2, dyn, "dense_train" if dyn else "p0")
    dt = torch.float32 if f32 else torch.float64
    env = BatchedDMPEnv(2, dyn, n, plans=table.reshape(len(table), 26, 26), seed=4, total_step=4, obs_dtype=dt, **kw)
    orc = helpers.oracle().OracleBatch(2, dyn, n, table, seed=4)
    norm = {None: dyn, "raw": False, "norm": True}[env.obs_scalars]
    orc.configure(obs_norm=norm, frame=env.frame_value, tail=env.obs_tail)
    orc.set_total_step(4)
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    assert env.reset().cpu().numpy().tobytes() == cast(orc.reset()).tobytes()
    t0 = 0
    for T in (1, 5, 2):
        og, rg, dg = env.rollout(T)
        assert _last_kernel() == "k_rollout2d"
        oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
        assert og.cpu().numpy().tobytes() == cast(oc).tobytes(), "observations"
        assert rg.cpu().numpy().tobytes() == rc.tobytes() and np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
        t0 += T
    rng = np.random.default_rng(1)
    acts, ks = rng.integers(0, 5, size=(3, n)).astype(np.int8), rng.integers(1, 4, size=(3, n)).astype(np.int8)
    og, rg, dg = env.rollout(3, actions=torch.from_numpy(acts).to(env.device), step_size=torch.from_numpy(ks).to(env.device))
    oc, rc, dc = orc.rollout(3, t0=t0, actions=acts, step_size=ks, nthreads=16)
    assert og.cpu().numpy().tobytes() == cast(oc).tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
    del og, oc
    # a longer launch against the tile kernel, on the device
    twin = env.fork(torch.arange(n, device=env.device))
    T = 16
    o1, r1, d1 = env.rollout(T)
    assert _last_kernel() == "k_rollout2d"
    raw = torch.empty(T * n * env.obs_dim + 1, dtype=dt, device=env.device)
    o2, r2, d2 = twin.rollout(T, out=raw[1:].view(T, n, env.obs_dim))
    assert _last_kernel() == "k_rollout" and o2.data_ptr() % 16 != 0
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    assert torch.equal(env._hdr, twin._hdr) and torch.equal(env._grid, twin._grid)
    # tile-major output of the same layout
    tw2 = env.fork(torch.arange(n, device=env.device))
    ot, _, _ = env.rollout(3, obs="tiled")
    assert _last_kernel() == "k_rollout2d"
    on, _, _ = tw2.rollout(3)
    assert torch.equal(env.untile(ot), on)
