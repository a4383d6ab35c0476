"""GPU: the 16-env-tile 3D rollout kernel (k_rollout3dw, round 3: sixteen envs per wave, the plan table as bit rows in LDS, no global
load in the loop, reward and running IoU sum resolved in the tick) against the CPU oracle.  The kernel takes 3D rollouts of 16 384
envs and more that write every observation, launches of 16 ticks and more: full tiles, a ragged last tile, blocks with idle waves;
static and dataset plans; float64 and float32; launches of 16 / 17 / 40 ticks; the `>` rule bits with short time limits; explicit
actions / step sizes (those launches stay on the 8-env kernel and must carry on from the same records); the record outputs and the tile-major output (against the 8-env kernel, which a shorter launch
selects); and a plan table that is NOT {0, z}-valued, which the kernel must notice and serve by loads."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

N0 = 16384


def _pair(dyn, n, seed, tag=None, total_step=None, f32=False, brick_gt=False, time_gt=False, base=0, table=None):
    import torch
    from snac_amd import BatchedDMPEnv

    if table is None:
        table = helpers.plan_table(3, dyn, tag or ("dense_train" if dyn else "p1"))
    env = BatchedDMPEnv(3, dyn, n, plans=table.reshape(len(table), 26, 26), seed=seed, env_id_base=base, total_step=total_step,
                        obs_dtype=torch.float32 if f32 else torch.float64, brick_gt=brick_gt, time_gt=time_gt)
    orc = helpers.oracle().OracleBatch(3, dyn, n, table, seed=seed, env_id_base=base)
    if total_step:
        orc.set_total_step(total_step)
    orc.set_rules(brick_gt, time_gt)
    o = orc.reset()
    assert env.reset().cpu().numpy().tobytes() == (o.astype(np.float32) if f32 else o).tobytes()
    return env, orc


def _compare(env, orc, T, t0, f32=False, actions=None, step_size=None):
    og, rg, dg = env.rollout(T, actions=actions, step_size=step_size)
    oc, rc, dc = orc.rollout(T, t0=t0, actions=actions, step_size=step_size, nthreads=16)
    assert og.cpu().numpy().tobytes() == (oc.astype(np.float32) if f32 else oc).tobytes(), "observations"
    assert rg.cpu().numpy().tobytes() == rc.tobytes(), "rewards"
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc), "done flags"


def _end_state(env, orc):
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
    assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("n", [N0, N0 + 21, N0 + 64 + 3])
def test_tiles_dtypes_and_launch_lengths(dyn, n, f32):
    """n = 16 384: full tiles; + 21: a block of one full and one ragged tile and two idle waves; + 67: a lone 3-env tile in a block of
    its own.  Launches of 16, 17 and 40 ticks (the reward / done runs are flushed every 16 ticks and at the end); a launch of 5 ticks
    in between runs on the 8-env kernel and must carry on from the same records."""
    env, orc = _pair(dyn, n, seed=5, total_step=60, f32=f32, base=7)
    t0 = 0
    for T in (16, 17, 5, 40):
        _compare(env, orc, T, t0, f32)
        t0 += T
    _end_state(env, orc)


@pytest.mark.parametrize("rules", [(False, False), (True, False), (False, True), (True, True)], ids=str)
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_rule_bits_with_explicit_inputs(dyn, rules):
    """Build-heavy explicit actions (the reference's own mix is [0.2 x 4, 0.05 x 4]; here builds dominate so that episodes also end
    at count_brick >= (>) total_brick), explicit step sizes incl. out-of-range values; time limit 50."""
    n, T = N0 + 21, 64
    env, orc = _pair(dyn, n, seed=9, tag="sparse_train" if dyn else "p1", total_step=50, brick_gt=rules[0], time_gt=rules[1])
    rng = np.random.default_rng(4)
    acts = rng.choice(np.arange(8, dtype=np.int8), size=(T, n), p=[0.1, 0.1, 0.1, 0.1, 0.15, 0.15, 0.15, 0.15])
    ks = rng.integers(0, 6, size=(T, n)).astype(np.int8)
    og, rg, dg = env.rollout(T, actions=acts, step_size=ks)
    oc, rc, dc = orc.rollout(T, t0=0, actions=acts, step_size=np.clip(ks, 1, 3), nthreads=16)
    assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
    _compare(env, orc, 20, T, actions=acts[:20])                  # actions only: counter-RNG step sizes
    _end_state(env, orc)


@pytest.mark.parametrize("rules", [(True, False), (False, True), (True, True)], ids=str)
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_rule_bits_on_the_counter_rng(dyn, rules):
    """The `>` termination tests inside the 16-env kernel itself (counter-RNG launches): sparse plans (small total_brick) so that
    episodes end by bricks as well as by the time limit of 40."""
    env, orc = _pair(dyn, N0 + 21, seed=13, tag="sparse_train" if dyn else "p1", total_step=40, brick_gt=rules[0], time_gt=rules[1])
    _compare(env, orc, 100, 0)
    _end_state(env, orc)


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
def test_tile_major_output_and_record_equal_the_8_env_kernel(f32):
    """The same batch through both 3D rollout kernels: 40 ticks in one launch (16-env tiles) against 4 launches of 10 ticks (below
    the 16-tick threshold: k_rollout3d) -- rows, rewards, done flags, record outputs, records; and the tile-major output holds the
    same rows."""
    import torch

    n, T = N0 + 21, 40
    dt = torch.float32 if f32 else torch.float64
    a, orc = _pair(True, n, seed=4, total_step=30, f32=f32)
    b = a.fork(torch.arange(n, device=a.device))
    c = a.fork(torch.arange(n, device=a.device))
    kinds = {"actions": torch.int8, "step_size": torch.int8, "plan_idx": torch.int16, "first": torch.uint8}
    ra = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    rb = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    oa, rwa, da = a.rollout(T, record=ra)
    ob = torch.empty((T, n, 51), dtype=dt, device=a.device)
    rwb = torch.empty((T, n), dtype=torch.float32, device=a.device)
    db = torch.empty((T, n), dtype=torch.uint8, device=a.device)
    for j in range(4):
        sl = slice(10 * j, 10 * j + 10)
        b.rollout(10, out=ob[sl], reward_out=rwb[sl], done_out=db[sl], record={k: v[sl] for k, v in rb.items()})
    assert torch.equal(oa, ob) and torch.equal(rwa, rwb) and torch.equal(da.view(torch.uint8), db)
    for k in kinds:
        assert torch.equal(ra[k], rb[k]), k
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid) and torch.equal(a._stats, b._stats) and torch.equal(a._episode, b._episode)
    ot, rt, dtt = c.rollout(T, obs="tiled")
    assert torch.equal(c.untile(ot), oa) and torch.equal(rt, rwa) and torch.equal(dtt, da)
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    assert oa.cpu().numpy().tobytes() == (oc.astype(np.float32) if f32 else oc).tobytes()


def test_a_plan_table_that_is_not_binary_is_served_by_loads():
    """Hindsight relabelling turns final height maps into plans: cells of any height.  The bit rows in LDS cannot hold such a table;
    every block notices while it builds its copy (largest cell != smallest non-zero cell) and loads the plan cell of a build."""
    table = helpers.plan_table(3, True, "dense_val").copy()
    rng = np.random.default_rng(2)
    grid = table.reshape(len(table), 26, 26)
    for p in range(0, len(table), 3):                             # a third of the plans get heights 1..9 on their footprint
        mask = grid[p] > 0
        grid[p][mask] = rng.integers(1, 10, size=int(mask.sum()))
    env, orc = _pair(True, N0, seed=12, total_step=40, table=table)
    _compare(env, orc, 48, 0)
    _end_state(env, orc)
