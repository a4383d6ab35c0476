"""GPU: 1D rollouts against the CPU oracle.  Rollouts that write every row run on the time-parallel kernel (k_rollout1dt, round 3:
one wavefront per env, lane = tick, blocks of 4 (below 3584 envs) or 8 envs (round 5: 16) whose rows leave through an LDS staging tile as whole runs per tick) up
to 65 536 envs -- canonical rows on aligned outputs with N % 4 = 0 leave it for k_rollout1dl from 45 056 envs (float32: 36 864; tests/test_gpu_rollout1dl.py) --, on the tile kernel beyond: batches on either side of every switch, ragged blocks and blocks
with idle waves, rows that can and cannot leave as 16-byte pieces (odd N, unaligned outputs), canonical and tile-major layouts,
odd / tiny tick counts, episodes that end by count_brick and by the time limit (several per chunk of 64 ticks), the `>` rule
bits, float32 observations, the per-step record outputs and explicit inputs."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _tables(dyn):
    t = helpers.plan_table(1, dyn, "sin_train" if dyn else "p1")
    return t, t.reshape(len(t), 30)


def _pair(dyn, n, seed, total_step=None, obs_dtype=None, brick_gt=False, time_gt=False):
    import torch
    from snac_amd import BatchedDMPEnv

    table, full = _tables(dyn)
    env = BatchedDMPEnv(1, dyn, n, plans=full, seed=seed, env_id_base=77, total_step=total_step, obs_dtype=obs_dtype or torch.float64,
                        brick_gt=brick_gt, time_gt=time_gt)
    orc = helpers.oracle().OracleBatch(1, dyn, n, table, seed=seed, env_id_base=77)
    if total_step:
        orc.set_total_step(total_step)
    orc.set_rules(brick_gt, time_gt)
    o = orc.reset()
    assert env.reset().cpu().numpy().tobytes() == (o.astype(np.float32) if obs_dtype == torch.float32 else o).tobytes()
    return env, orc


def _compare(env, orc, T, t0, f32=False):
    og, rg, dg = env.rollout(T)
    oc, rc, dc = orc.rollout(T, t0=t0, nthreads=8)
    want = oc.astype(np.float32) if f32 else oc
    assert og.cpu().numpy().tobytes() == want.tobytes()
    assert rg.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)


def _end_state(env, orc):
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(env.num_envs, -1), st["grid"])
    for name, key in (("count_brick", "cb"), ("count_step", "cs"), ("plan_idx", "plan_idx"), ("episode", "episode"), ("episode_return", "ep_return")):
        assert np.array_equal(getattr(env, name).cpu().numpy(), st[key]), name
    assert np.array_equal(env.need_reset.cpu().numpy().astype(np.uint8), st["need_reset"])
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))


@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("n", [1, 13, 2049, 3583, 3585, 4100, 8200, 16384])
def test_batch_shapes_and_tick_counts(dyn, n):
    """n = 1 / 13 / 2049 / 3583: blocks of 4 envs, the last one ragged, odd N (rows leave element by element); 3585: blocks of 16, the
    last holds one env and fifteen idle waves; 4100 / 8200: a last block of 4 / 8 envs; 16 384: full blocks only.  Launches of 1, 2,
    37 and 80 steps."""
    env, orc = _pair(dyn, n, seed=5, total_step=60)
    t0 = 0
    for T in (1, 2, 37, 80):
        _compare(env, orc, T, t0)
        t0 += T
    _end_state(env, orc)


@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("rules", [(False, False), (True, False), (False, True), (True, True)], ids=str)
def test_episodes_end_by_bricks_and_by_time(dyn, rules):
    """total_step 3000: a third of the steps drop a brick, so episodes end at count_brick >= (>) total_brick (~600) long before the
    time limit; total_step 45: they end by time, > 60 episodes per env."""
    for total_step, T in ((3000, 2900), (45, 2900)):
        env, orc = _pair(dyn, 72, seed=9, total_step=total_step, brick_gt=rules[0], time_gt=rules[1])
        _compare(env, orc, T, 0)
        _end_state(env, orc)
        assert env.episodic_stats()["episodes"] > 0


def test_float32_observations():
    import torch

    env, orc = _pair(True, 200, seed=2, obs_dtype=torch.float32)
    _compare(env, orc, 750, 0, f32=True)
    _compare(env, orc, 33, 750, f32=True)


@pytest.mark.parametrize("n", [40, 9000, 50000])
def test_record_outputs_fed_back_as_explicit_inputs(n):
    """The record outputs (action taken, step size used, plan row, first-step flag) of a counter-RNG rollout, fed back as
    explicit inputs (the time-parallel kernel's prefetch-free EXPL variant; the tile kernel at n = 50 000), must reproduce
    observations, rewards and done flags -- which equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv

    table, full = _tables(True)
    T = 130
    a = BatchedDMPEnv(1, True, n, plans=full, seed=21, total_step=50)
    b = BatchedDMPEnv(1, True, n, plans=full, seed=21, total_step=50)
    orc = helpers.oracle().OracleBatch(1, True, n, table, seed=21, env_id_base=0)
    orc.set_total_step(50)
    orc.reset()
    a.reset()
    b.reset()
    rec = {"actions": torch.empty((T, n), dtype=torch.int8, device="cuda"), "step_size": torch.empty((T, n), dtype=torch.int8, device="cuda"),
           "plan_idx": torch.empty((T, n), dtype=torch.int16, device="cuda"), "first": torch.empty((T, n), dtype=torch.uint8, device="cuda")}
    oa, ra, da = a.rollout(T, record=rec)
    ob, rb, db = b.rollout(T, actions=rec["actions"], step_size=rec["step_size"])
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=8)
    assert oa.cpu().numpy().tobytes() == oc.tobytes() and ra.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(da.cpu().numpy().view(np.uint8), dc)
    first = rec["first"].cpu().numpy()
    done = da.cpu().numpy()
    assert first[0].all() and np.array_equal(first[1:], done[:-1].astype(np.uint8))   # auto-reset: a step opens an episode iff the last one ended one
    assert np.array_equal(rec["plan_idx"][-1].cpu().numpy(), a.plan_idx.cpu().numpy())   # an env is reset by its NEXT step


@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
@pytest.mark.parametrize("total_step,time_gt", [(1, False), (5, False), (7, True), (63, False), (64, False), (65, True)])
def test_time_parallel_kernel_segments_inside_a_chunk(dyn, total_step, time_gt):
    """k_rollout1dt: one wavefront per env, lane = tick, 64 ticks per chunk.  An
    episode that ends inside a chunk splits it into segments (the reset happens in the wave's uniform state): time limits of 1
    (every lane its own episode), 5 and 7 (a dozen segments per chunk), and 63 / 64 / 65 (ends on, just before and just behind the
    chunk border); launches of 1, 64, 65 and 200 ticks; both dtypes; explicit inputs on the last launch."""
    import torch

    for f32 in (False, True):
        env, orc = _pair(dyn, 37, seed=8, total_step=total_step, obs_dtype=torch.float32 if f32 else None, time_gt=time_gt)
        t0 = 0
        for T in (1, 64, 65, 200):
            _compare(env, orc, T, t0, f32)
            t0 += T
        rng = np.random.default_rng(total_step)
        acts = rng.integers(0, 3, size=(70, 37)).astype(np.int8)
        ks = rng.integers(1, 4, size=(70, 37)).astype(np.int8)
        og, rg, dg = env.rollout(70, actions=acts, step_size=ks)
        oc, rc, dc = orc.rollout(70, t0=t0, actions=acts, step_size=ks, nthreads=8)
        assert og.cpu().numpy().tobytes() == (oc.astype(np.float32) if f32 else oc).tobytes()
        assert rg.cpu().numpy().tobytes() == rc.tobytes() and np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
        _end_state(env, orc)


@pytest.mark.parametrize("n,f32", [(2050, False), (2051, False), (3580, False), (3586, False), (3587, False), (4100, True), (4102, True), (57344, False), (65536, False), (65552, False), (65536, True), (65552, True)])
def test_staged_rows_alignments_layouts_and_the_switch_to_the_tile_kernel(n, f32):
    """Rows of a block leave as 16-byte pieces when every run of a tick starts and ends on 16 bytes (float64: even N; float32:
    N % 4 = 0) and element by element otherwise -- also when the output itself is not 16-byte aligned; tile-major outputs hold the
    same rows at [env // 64, t, env % 64].  Blocks of 4 envs below 3584, of 8 envs from there (round 6; 16 before); 65 536 is the largest
    batch of the time-parallel kernel (round 5: 57 344 with float64 rows), 16 envs more run on the tile kernel: the same rows either way."""
    import torch

    dt = torch.float32 if f32 else torch.float64
    T = 70
    env, orc = _pair(True, n, seed=3, total_step=40, obs_dtype=dt)
    twin = env.fork(torch.arange(n, device=env.device))
    third = env.fork(torch.arange(n, device=env.device))
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    want = (oc.astype(np.float32) if f32 else oc).tobytes()
    og, rg, dg = env.rollout(T)
    assert og.cpu().numpy().tobytes() == want and rg.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
    ot, rt, dtt = twin.rollout(T, obs="tiled")
    assert twin.untile(ot).cpu().numpy().tobytes() == want and torch.equal(rt, rg) and torch.equal(dtt, dg)
    raw = torch.empty(T * n * 7 + 1, dtype=dt, device=env.device)
    ou, ru, du = third.rollout(T, out=raw[1:].view(T, n, 7))
    assert ou.data_ptr() % 16 != 0 and ou.cpu().numpy().tobytes() == want and torch.equal(ru, rg) and torch.equal(du, dg)
    _end_state(env, orc)
    assert torch.equal(env._hdr, twin._hdr) and torch.equal(env._hdr, third._hdr) and torch.equal(env._grid, third._grid)


def test_replay_rings_filled_by_the_time_parallel_kernel():
    """ReplayRing.collect on a 1D batch whose rollouts run on k_rollout1dt in blocks of 16 envs (a ragged last block): the tick ring
    and the tile-major ring (launches that write at an offset of the ring and wrap) hold the same rows, records and samples; the last
    launch's rows equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    n = 4100
    table, full = _tables(True)
    envs = [BatchedDMPEnv(1, True, n, plans=full, seed=12, total_step=40) for _ in range(2)]
    orc = helpers.oracle().OracleBatch(1, True, n, table, seed=12)
    orc.set_total_step(40)
    orc.reset()
    orc.rollout(13, t0=0, obs=None, nthreads=8)
    rings = []
    for e, layout in zip(envs, ("ticks", "tiled")):
        e.reset()
        e.rollout(13, obs=None)                                       # attach in mid-episode
        rings.append(ReplayRing(e, 100, layout=layout))
    t0 = 13
    for T in (70, 90, 100, 7):                                        # 64-tick chunks that straddle the ring's end
        for r in rings:
            r.collect(T)
        oc, rc, dc = orc.rollout(T, t0=t0, nthreads=8)
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


VARIANTS_1D = [
    (False, dict(layout="lnet1d")),                                                     # 8 values: the position appended
    (True, dict(layout="ppo")),                                                         # 37: window, counters, the 30 plan heights
    (True, dict(obs_tail=("record",), obs_scalars="raw")),                              # 15
    (False, dict(obs_tail=("position", "plan", "record"), frame_value=2, obs_scalars="norm")),   # 46
]


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn,kw", VARIANTS_1D, ids=["lnet1d", "ppo", "record", "all"])
def test_layout_variants_on_the_time_parallel_kernel(dyn, kw, f32):
    """The observation layouts of the reference's 1D env copies (snac_env_desc.frame_value / obs_scalars / obs_tail) on k_rollout1dt:
    a lane files its whole row, tails included.  Against the oracle configured the same way -- time limit 6 (every env starts over,
    on a new plan row, ten times per chunk) and the kind's own; launches of 1, 5, 70 and 130 ticks, explicit inputs --, against the
    tile kernel (an unaligned output; a batch that is not whole groups of four envs), and the tile-major output."""
    import torch
    from snac_amd import BatchedDMPEnv, _lib

    def kernel():
        return _lib.lib().snac_last_kernel().decode()

    n = 1000
    table, full = _tables(dyn)
    dt = torch.float32 if f32 else torch.float64
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    for total_step in (6, None):
        env = BatchedDMPEnv(1, dyn, n, plans=full, seed=4, total_step=total_step, obs_dtype=dt, **kw)
        orc = helpers.oracle().OracleBatch(1, dyn, n, table, seed=4)
        norm = {None: dyn, "raw": False, "norm": True}[env.obs_scalars]
        orc.configure(obs_norm=norm, frame=env.frame_value, tail=env.obs_tail)
        if total_step:
            orc.set_total_step(total_step)
        assert env.reset().cpu().numpy().tobytes() == cast(orc.reset()).tobytes()
        t0 = 0
        for T in (1, 5, 70, 130):
            og, rg, dg = env.rollout(T)
            assert kernel() == "k_rollout1dt"
            oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
            assert og.cpu().numpy().tobytes() == cast(oc).tobytes(), ("observations", total_step, T)
            assert rg.cpu().numpy().tobytes() == rc.tobytes() and np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
            t0 += T
        rng = np.random.default_rng(1)
        acts, ks = rng.integers(0, 3, size=(70, n)).astype(np.int8), rng.integers(1, 4, size=(70, n)).astype(np.int8)
        og, rg, dg = env.rollout(70, actions=torch.from_numpy(acts).to(env.device), step_size=torch.from_numpy(ks).to(env.device))
        assert kernel() == "k_rollout1dt"
        oc, rc, dc = orc.rollout(70, t0=t0, actions=acts, step_size=ks, nthreads=16)
        assert og.cpu().numpy().tobytes() == cast(oc).tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
        del og, oc
        _end_state(env, orc)
    twin = env.fork(torch.arange(n, device=env.device))
    T = 750
    o1, r1, d1 = env.rollout(T)
    assert kernel() == "k_rollout1dt"
    raw = torch.empty(T * n * env.obs_dim + 4, dtype=dt, device=env.device)
    o2, r2, d2 = twin.rollout(T, out=raw[1:1 + T * n * env.obs_dim].view(T, n, env.obs_dim))
    assert kernel() == "k_rollout" and o2.data_ptr() % 16 != 0
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    assert torch.equal(env._hdr, twin._hdr) and torch.equal(env._grid, twin._grid)
    tw2 = env.fork(torch.arange(n, device=env.device))
    ot, _, _ = env.rollout(70, obs="tiled")
    assert kernel() == "k_rollout1dt"
    on, _, _ = tw2.rollout(70)
    assert torch.equal(env.untile(ot), on)
    # not whole groups of four envs: the tile kernel, the same oracle
    big = BatchedDMPEnv(1, dyn, 4102, plans=full, seed=9, total_step=6, obs_dtype=dt, **kw)
    orc = helpers.oracle().OracleBatch(1, dyn, 4102, table, seed=9)
    orc.configure(obs_norm={None: dyn, "raw": False, "norm": True}[big.obs_scalars], frame=big.frame_value, tail=big.obs_tail)
    orc.set_total_step(6)
    assert big.reset().cpu().numpy().tobytes() == cast(orc.reset()).tobytes()
    og, rg, dg = big.rollout(20)
    assert kernel() == "k_rollout"
    oc, rc, dc = orc.rollout(20, t0=0, nthreads=16)
    assert og.cpu().numpy().tobytes() == cast(oc).tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
