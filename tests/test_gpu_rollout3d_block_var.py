"""GPU: 3D rollouts with a layout variant on the block kernel (k_rollout3db, MODE 1: rows of 51 .. 61 values -- raw / normalised scalar
slots, position and record tails; MODE 2: rows with the plan tail, 451 .. 461 values), round 5.  The stepper publishes the record values,
the writers assemble their 8 rows from per-lane descriptors, and -- with the plan tail -- the stepper loads the plan rows of envs that
start an episode on another row and hands them over behind a second barrier.  Against the CPU oracle: full and ragged blocks, float64 and
float32 rows, dataset and static plans, launches of 1 / 2 / 37 steps, explicit inputs, tile-major output, time limits of 1 .. 3 (an env
changes its plan row every tick: more than eight rows to hand over at once), the `>` rules; bit for bit what the tile kernel writes for
the same batch (an unaligned output selects it); the kernel either side of each default threshold.  Rows with the plan tail take the
block kernel from 10 240 envs (float64) / 16 384 (float32), so their cases run at 6180 envs in ONE child process with the thresholds lowered."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

N0 = 6144
INNER = os.environ.get("SNAC_TEST_VAR3D_INNER") == "1"
inner = pytest.mark.skipif(not INNER, reason="runs in the child process of test_rows_with_the_plan_tail_in_a_child_process")


def _kernel():
    from snac_amd import _lib

    return _lib.lib().snac_last_kernel().decode()


def _pair(dyn, n, seed, kw, tag=None, total_step=None, f32=False, brick_gt=False, time_gt=False, base=0):
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(3, dyn, tag or ("dense_train" if dyn else "p1"))
    env = BatchedDMPEnv(3, dyn, n, plans=table.reshape(len(table), 26, 26), seed=seed, env_id_base=base, total_step=total_step,
                        obs_dtype=torch.float32 if f32 else torch.float64, brick_gt=brick_gt, time_gt=time_gt, **kw)
    orc = helpers.oracle().OracleBatch(3, dyn, n, table, seed=seed, env_id_base=base)
    if total_step:
        orc.set_total_step(total_step)
    orc.set_rules(brick_gt, time_gt)
    norm = {None: dyn, "raw": False, "norm": True}[env.obs_scalars]
    orc.configure(obs_norm=norm, frame=env.frame_value, tail=env.obs_tail)
    assert orc.obs_dim == env.obs_dim
    o = orc.reset()
    assert env.reset().cpu().numpy().tobytes() == (o.astype(np.float32) if f32 else o).tobytes()
    return env, orc


def _compare(env, orc, T, t0, f32=False, actions=None, step_size=None, kernel="k_rollout3db"):
    import torch

    a = None if actions is None else torch.from_numpy(actions).to(env.device)
    k = None if step_size is None else torch.from_numpy(step_size).to(env.device)
    og, rg, dg = env.rollout(T, actions=a, step_size=k)
    assert _kernel() == kernel
    oc, rc, dc = orc.rollout(T, t0=t0, actions=actions, step_size=step_size, nthreads=16)
    want = oc.astype(np.float32) if f32 else oc
    got = og.cpu().numpy()
    if got.tobytes() != want.tobytes():
        bad = np.argwhere(got.view(np.uint32 if f32 else np.uint64) != want.view(np.uint32 if f32 else np.uint64))
        raise AssertionError("observations: %d values differ, first at (t, env, value) %s: %r != %r" % (len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))
    assert rg.cpu().numpy().tobytes() == rc.tobytes(), "rewards"
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc), "done flags"


def _end_state(env, orc):
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
    assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(env.num_envs, -1), st["grid"])


SHORT = [dict(obs_tail=("record",)), dict(obs_tail=("position", "record"), obs_scalars="raw"), dict(obs_tail=("position",)), dict(obs_scalars="raw"),
         dict(obs_scalars="norm")]


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("kw", SHORT, ids=lambda k: "+".join(sorted(map(str, k.get("obs_tail", ())))) + "_" + str(k.get("obs_scalars", "def")))
def test_rows_without_the_plan_tail(kw, dyn, f32):
    """Rows of 51 / 53 / 59 / 61 values on k_rollout3db's first variant form: a last block of 36 envs (four full writer waves, one with 4
    envs); launches of 1, 2 and 37 steps (random 3D agents box themselves in every ~22 steps: envs start over in every launch of 37)."""
    if kw == dict(obs_scalars="raw" if not dyn else "norm"):
        pytest.skip("the canonical layout of this class")
    env, orc = _pair(dyn, N0 + 36, 5, kw, f32=f32, base=11)
    t0 = 0
    for T in (1, 2, 37):
        _compare(env, orc, T, t0, f32)
        t0 += T
    _end_state(env, orc)
    _compare(env, orc, 3, t0, f32)


@pytest.mark.parametrize("n", [64, 68, 1000, N0 + 64 + 4])
def test_small_batches_and_a_last_block_of_four_envs(n):
    """From 64 envs (SNAC_3D_BLOCK_VAR_MIN) the variant rows without the plan tail leave the tile kernel: one block; one block and a
    block of 4 envs; 15 blocks and one of 40; 97 and one of 4."""
    env, orc = _pair(True, n, 3, dict(obs_tail=("position", "record")), total_step=30)
    _compare(env, orc, 41, 0)
    _compare(env, orc, 1, 41)
    _end_state(env, orc)


def test_sixty_envs_stay_on_the_tile_kernel_and_write_the_same_rows():
    """60 envs of a batch of 64 (same seeds and global ids): below the threshold, on the tile kernel; the rows of the common envs agree."""
    import torch
    from snac_amd import BatchedDMPEnv

    kw = dict(obs_tail=("record",), seed=4, total_step=25)
    big, small = BatchedDMPEnv(3, True, 64, **kw), BatchedDMPEnv(3, True, 60, **kw)
    assert torch.equal(big.reset()[:60], small.reset())
    ob, rb, db = big.rollout(60)
    assert _kernel() == "k_rollout3db"
    os_, rs, ds = small.rollout(60)
    assert _kernel() == "k_rollout"
    assert torch.equal(ob[:, :60], os_) and torch.equal(rb[:, :60], rs) and torch.equal(db[:, :60], ds)


@pytest.mark.parametrize("total_step,time_gt", [(1, False), (1, True), (2, False), (3, False)])
def test_envs_that_start_over_every_tick(total_step, time_gt):
    n = N0 + 36
    env, orc = _pair(True, n, 9, dict(obs_tail=("record",)), total_step=total_step, time_gt=time_gt)
    rng = np.random.default_rng(total_step)
    acts = rng.choice(np.arange(8, dtype=np.int8), size=(24, n), p=[0.05, 0.05, 0.05, 0.05, 0.2, 0.2, 0.2, 0.2])
    _compare(env, orc, 24, 0, actions=acts)
    _compare(env, orc, 25, 24)
    _end_state(env, orc)


@pytest.mark.parametrize("rules", [(False, False), (True, True)], ids=str)
def test_episodes_end_by_bricks_boxed_in_and_by_time(rules):
    n, T = N0 + 36, 120
    env, orc = _pair(True, n, 9, dict(layout="ppo", obs_tail=("record",)), tag="sparse_train", total_step=45, brick_gt=rules[0], time_gt=rules[1])
    rng = np.random.default_rng(3)
    acts = rng.choice(np.arange(8, dtype=np.int8), size=(T, n), p=[0.1, 0.1, 0.1, 0.1, 0.15, 0.15, 0.15, 0.15])
    _compare(env, orc, T, 0, actions=acts)
    _end_state(env, orc)
    assert env.episodic_stats()["episodes"] > 2 * n


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
def test_tile_major_output_record_outputs_and_the_tile_kernel(f32):
    """rollout(obs="tiled") holds the same rows at [env // 64, t, env % 64]; rows, record outputs and the final records equal what the
    tile kernel gives for an identical batch (an output that is not 16-byte aligned selects it)."""
    import torch

    n, T = N0 + 36, 33
    dt = torch.float32 if f32 else torch.float64
    a, orc = _pair(True, n, 4, dict(obs_tail=("position", "record"), obs_scalars="raw"), total_step=20, f32=f32)
    b = a.fork(torch.arange(n, device=a.device))
    kinds = {"actions": torch.int8, "step_size": torch.int8, "plan_idx": torch.int16, "first": torch.uint8}
    ra = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    rb = {k: torch.empty((T, n), dtype=v, device=a.device) for k, v in kinds.items()}
    ot, rt, dtt = a.rollout(T, obs="tiled", record=ra)
    assert _kernel() == "k_rollout3db"
    raw = torch.empty(T * n * 61 + 1, dtype=dt, device=a.device)
    ob, rwb, db = b.rollout(T, out=raw[1:].view(T, n, 61), record=rb)
    assert ob.data_ptr() % 16 != 0 and _kernel() == "k_rollout"
    assert torch.equal(a.untile(ot), ob) and torch.equal(rt, rwb) and torch.equal(dtt, db)
    for k in kinds:
        assert torch.equal(ra[k], rb[k]), k
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    assert ob.cpu().numpy().tobytes() == (oc.astype(np.float32) if f32 else oc).tobytes()
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid) and torch.equal(a._stats, b._stats) and torch.equal(a._episode, b._episode)


def test_plan_tail_rows_either_side_of_the_default_thresholds():
    """Rows with the plan tail: float64 from 10 240 envs on k_rollout3db (10 236: the tile kernel), float32 from 16 384 (16 380: the tile
    kernel); the common envs of each pair of batches (same seeds and global ids) agree, and the block kernel's float64 rows equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv

    big, orc = _pair(True, 10240, 6, dict(layout="ppo"), total_step=3)
    small = BatchedDMPEnv(3, True, 10236, seed=6, total_step=3, layout="ppo", plans=big.plans_full)
    small.reset()
    ob, rb, db = big.rollout(7)
    assert _kernel() == "k_rollout3db" and tuple(ob.shape) == (7, 10240, 451)
    os_, rs, ds = small.rollout(7)
    assert _kernel() == "k_rollout"
    assert torch.equal(ob[:, :10236], os_) and torch.equal(rb[:, :10236], rs) and torch.equal(db[:, :10236], ds)
    oc, rc, dc = orc.rollout(7, t0=0, nthreads=16)
    assert ob.cpu().numpy().tobytes() == oc.tobytes() and rb.cpu().numpy().tobytes() == rc.tobytes()
    del ob, os_, big, small
    b32 = BatchedDMPEnv(3, True, 16384, seed=6, total_step=4, layout="ppo", obs_dtype=torch.float32)
    s32 = BatchedDMPEnv(3, True, 16380, seed=6, total_step=4, layout="ppo", obs_dtype=torch.float32)
    assert torch.equal(b32.reset()[:16380], s32.reset())
    ob, rb, db = b32.rollout(6)
    assert _kernel() == "k_rollout3db"
    os_, rs, ds = s32.rollout(6)
    assert _kernel() == "k_rollout"
    assert torch.equal(ob[:, :16380], os_) and torch.equal(rb[:, :16380], rs) and torch.equal(db[:, :16380], ds)


# ---- rows with the plan tail at 6180 envs: in a child process with the thresholds at 4096 envs
PLAN = [dict(layout="ppo"), dict(obs_tail=("position", "plan", "record"), obs_scalars="raw"), dict(obs_tail=("plan",), obs_scalars="norm")]


@inner
@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("kw", PLAN, ids=["ppo", "all", "plan_norm"])
def test_inner_rows_with_the_plan_tail(kw, dyn, f32):
    """451 / 461-value rows: launches of 1, 2 and 37 steps on a batch with a last block of 36 envs; the dataset classes change their plan
    row with every episode (the stepper hands the new rows over), the static ones never do."""
    env, orc = _pair(dyn, N0 + 36, 5, kw, f32=f32, base=11)
    t0 = 0
    for T in (1, 2, 37):
        _compare(env, orc, T, t0, f32)
        t0 += T
    _end_state(env, orc)
    _compare(env, orc, 3, t0, f32)


@inner
@pytest.mark.parametrize("total_step,time_gt", [(1, False), (1, True), (2, False), (3, False)])
def test_inner_every_env_changes_its_plan_row_every_tick(total_step, time_gt):
    """A time limit of 1: all 64 envs of a block start every tick on a new plan row -- eight rows ride in the stepper's registers, the other
    56 are fetched in rounds between the tick's two barriers."""
    n = N0 + 36
    env, orc = _pair(True, n, 9, dict(layout="ppo"), total_step=total_step, time_gt=time_gt)
    rng = np.random.default_rng(total_step)
    acts = rng.choice(np.arange(8, dtype=np.int8), size=(12, n), p=[0.05, 0.05, 0.05, 0.05, 0.2, 0.2, 0.2, 0.2])
    _compare(env, orc, 12, 0, actions=acts)
    _compare(env, orc, 13, 12)
    _end_state(env, orc)


@inner
def test_inner_plan_tail_tile_major_explicit_inputs_and_the_tile_kernel():
    import torch

    n, T = N0 + 36, 21
    a, orc = _pair(True, n, 4, dict(layout="ppo"), total_step=20)
    b = a.fork(torch.arange(n, device=a.device))
    rng = np.random.default_rng(5)
    acts, ks = rng.integers(0, 8, size=(T, n)).astype(np.int8), rng.integers(1, 4, size=(T, n)).astype(np.int8)
    ta, tk = torch.from_numpy(acts).cuda(), torch.from_numpy(ks).cuda()
    ot, rt, dtt = a.rollout(T, obs="tiled", actions=ta, step_size=tk)
    assert _kernel() == "k_rollout3db"
    raw = torch.empty(T * n * 451 + 1, dtype=torch.float64, device=a.device)
    ob, rwb, db = b.rollout(T, out=raw[1:].view(T, n, 451), actions=ta, step_size=tk)
    assert _kernel() == "k_rollout"
    assert torch.equal(a.untile(ot), ob) and torch.equal(rt, rwb) and torch.equal(dtt, db)
    oc, rc, dc = orc.rollout(T, t0=0, actions=acts, step_size=ks, nthreads=16)
    assert ob.cpu().numpy().tobytes() == oc.tobytes()
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid) and torch.equal(a._stats, b._stats) and torch.equal(a._episode, b._episode)


@inner
def test_inner_a_pending_reset_carried_into_the_next_launch():
    """Launches that end on a done step: the header keeps the flag, the next launch picks the new plan row before its tick 0 and all nine
    waves bring the rows in."""
    n = N0
    env, orc = _pair(True, n, 6, dict(layout="ppo"), total_step=5)
    t0 = 0
    for T in (5, 1, 4, 5, 7):
        _compare(env, orc, T, t0)
        t0 += T
    _end_state(env, orc)


def test_rows_with_the_plan_tail_in_a_child_process():
    if INNER:
        pytest.skip("the child itself")
    env = dict(os.environ, SNAC_TEST_VAR3D_INNER="1", SNAC_3D_BLOCK_VAR_PLAN_F64="4096", SNAC_3D_BLOCK_VAR_PLAN_F32="4096")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "inner", "-p", "no:cacheprovider"],
                         cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "skipped" not in out.stdout.splitlines()[-1], out.stdout[-500:]
