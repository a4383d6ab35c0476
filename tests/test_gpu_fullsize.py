"""GPU: BASELINE.json configurations at full size, bit-compared with the CPU oracle, plus edge cases (ragged tile
tails, single env, masks, output modes)."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _threads():
    try:
        return max(1, min(32, len(os.sched_getaffinity(0))))
    except AttributeError:
        return 4


def _pair(dim, dyn, n, tag, seed=1, base=0, **kw):
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(dim, dyn, tag)
    full = table.reshape(len(table), 30) if dim == 1 else table.reshape(len(table), 26, 26)
    env = BatchedDMPEnv(dim, dyn, n, plans=full, seed=seed, env_id_base=base, **kw)
    orc = helpers.oracle().OracleBatch(dim, dyn, n, table, seed=seed, env_id_base=base)
    return env, orc


def _compare_chunks(env, orc, T, chunk, memory="malloc"):
    import torch

    nt = _threads()
    assert helpers.same_bytes(env.reset().cpu().numpy(), orc.reset())
    if memory == "vmm":                                           # trajectory memory (snac_traj_alloc): chunks of two slices taking turns
        from snac_amd import trajmem

        buf = trajmem.traj_empty((chunk, env.num_envs, env.obs_dim), torch.float64, env.device)
    else:
        buf = torch.empty((chunk, env.num_envs, env.obs_dim), dtype=torch.float64, device=env.device)
    t = 0
    while t < T:
        c = min(chunk, T - t)
        og, rg, dg = env.rollout(c, out=buf[:c])
        oc, rc, dc = orc.rollout(c, t0=t, nthreads=nt)
        assert helpers.same_bytes(og.cpu().numpy(), oc), ("obs", t)
        assert helpers.same_bytes(rg.cpu().numpy(), rc), ("reward", t)
        assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc), ("done", t)
        t += c
    s = orc.stats()
    e = env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
    assert helpers.same_bytes(env.iou().cpu().numpy(), orc.iou())
    return e


def test_config2_1d_static_4096_envs_full_episode():
    """BASELINE configs[1]: 1D static plan 0 (sin), N = 4096, all 750 steps."""
    env, orc = _pair(1, False, 4096, "p0")
    e = _compare_chunks(env, orc, 750, 250)
    assert e["episodes"] == 4096                                  # every env hits the time limit exactly once


def test_config3_2d_dynamic_dense_65536_envs_full_pass():
    """BASELINE configs[2] (headline): 2D dynamic dense, N = 65536, T = 600 -- every observation, reward and done
    flag of all 39 321 600 env-steps equals the oracle's.  The observations are written into trajectory memory (a 1.07 GB block of
    snac_traj_alloc, the measured two-slice layout: what bench.py writes into) and read back from there."""
    env, orc = _pair(2, True, 65536, "dense_train")
    e = _compare_chunks(env, orc, 600, 40, memory="vmm")
    assert e["episodes"] > 65536


def test_config5_3d_dynamic_dense_16384_envs_full_pass():
    """BASELINE configs[4]: 3D dynamic dense, N = 16384, T = 1000."""
    env, orc = _pair(3, True, 16384, "dense_train")
    e = _compare_chunks(env, orc, 1000, 100)
    assert e["episodes"] > 16384 * 20                             # random agents box themselves in every ~22 steps


@pytest.mark.parametrize("rank", [0, 5, 7])
def test_config4_shard_of_524288_envs(rank):
    """BASELINE configs[3]: ranks 0, 5 and 7 of 8 x 65536 envs, every one of the pass's 600 ticks -- the shard's global ids key
    the counter RNG, so what a rank computes does not depend on the ranks beside it."""
    env, orc = _pair(2, True, 65536, "dense_train", base=rank * 65536)
    e = _compare_chunks(env, orc, 600, 40)
    assert e["episodes"] > 65536


@pytest.mark.parametrize("kind", [(1, False), (1, True), (2, False), (2, True), (3, False), (3, True)], ids=str)
@pytest.mark.parametrize("n", [1, 17, 65, 130])
def test_ragged_tile_tails(kind, n):
    dim, dyn = kind
    tag = ("sin_val" if dim == 1 else "sparse_val") if dyn else "p1"
    env, orc = _pair(dim, dyn, n, tag, seed=4)
    _compare_chunks(env, orc, 90, 45)


def test_masked_reset_and_explicit_plan_indices():
    import torch

    n = 100
    env, orc = _pair(2, True, n, "dense_train", seed=2)
    env.reset()
    orc.reset()
    env.rollout(50, obs=None)
    orc.rollout(50, obs=None)
    rng = np.random.default_rng(0)
    mask = (rng.random(n) < 0.4).astype(np.uint8)
    pidx = rng.integers(0, 400, size=n).astype(np.int32)
    og = env.reset(torch.from_numpy(mask), torch.from_numpy(pidx))
    oc = orc.reset(mask, pidx)
    assert helpers.same_bytes(og.cpu().numpy(), oc)             # untouched envs report their current observation
    assert np.array_equal(env.plan_idx.cpu().numpy()[mask == 1], pidx[mask == 1])
    assert np.array_equal(env.count_step.cpu().numpy()[mask == 0], np.full(int((mask == 0).sum()), 50))
    og, rg, dg = env.rollout(30, obs="last")
    oc, rc, dc = orc.rollout(30, t0=50, obs="last")
    assert helpers.same_bytes(og.cpu().numpy(), oc) and helpers.same_bytes(rg.cpu().numpy(), rc)
    with pytest.raises(ValueError):
        env.reset(plan_idx=np.full(n, 400))


def test_step_without_auto_reset_keeps_mutating_like_the_reference():
    """no auto-reset; stepping after done keeps counting."""
    import torch

    env, orc = _pair(2, False, 8, "p1", seed=1)
    env.reset()
    orc.reset()
    a = torch.full((8,), 4, dtype=torch.int8)
    k = torch.ones(8, dtype=torch.int8)
    for t in range(70):                                           # total_brick = 60: done at step 60, then keep dropping
        og, rg, dg = env.step(a, k)
        oc, rc, dc = orc.step(t, a.numpy(), k.numpy())
        assert helpers.same_bytes(og.cpu().numpy(), oc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
    assert int(env.count_brick[0]) == 70 and bool(env.need_reset[0])


def test_api_argument_checks():
    from snac_amd import BatchedDMPEnv, SnacError

    env = BatchedDMPEnv(2, True, 4)
    with pytest.raises(SnacError):
        env.step()
    env.reset()
    with pytest.raises(ValueError):
        env.step(actions=np.zeros(5, np.int8))
    with pytest.raises(ValueError):
        env.rollout(3, actions=np.zeros((2, 4), np.int8))
    import torch

    with pytest.raises(ValueError):
        env.rollout(3, out=torch.empty((3, 4, 51), dtype=torch.float32, device=env.device))
    assert env.input_plan().shape == (4, 20, 20) and env.plan().shape == (4, 26, 26)
    assert env.environment_memory().shape == (4, 26, 26)


@pytest.mark.parametrize("kind", [2, 3])
def test_four_million_envs_in_one_launch_equal_their_shards(kind):
    """Index arithmetic far beyond the benchmark sizes: 2^22 envs (a 3.4 GB observation tensor per two ticks) stepped in
    one launch must equal the same global env ids stepped as 8 shards (shard results are oracle-checked at small N)."""
    import torch
    from snac_amd import BatchedDMPEnv

    N, S, T = 1 << 22, 8, 2
    big = BatchedDMPEnv(kind, True, N, seed=77)
    big.reset()
    big.rollout(3, obs=None)
    o, r, d = big.rollout(T)
    assert o.shape == (T, N, 51)
    for sh in (0, 3, S - 1):
        n = N // S
        part = BatchedDMPEnv(kind, True, n, seed=77, env_id_base=sh * n)
        part.reset()
        part.rollout(3, obs=None)
        po, pr, pd = part.rollout(T)
        sl = slice(sh * n, (sh + 1) * n)
        assert torch.equal(po, o[:, sl]) and torch.equal(pr, r[:, sl]) and torch.equal(pd, d[:, sl])
        assert torch.equal(part._grid, big._grid[sl]) and torch.equal(part._hdr, big._hdr[sl])
    # the single-step kernel and tree-search edges at the same scale: the last rows of the pool
    src = torch.arange(N - 4096, N, dtype=torch.int32, device=big.device)
    acts = torch.zeros(4096, dtype=torch.int8, device=big.device)
    before = big._hdr[N - 4096:].clone()
    o1, _, _ = big.transition(acts, None, src, src)
    assert o1.shape == (4096, 51) and not torch.equal(big._hdr[N - 4096:], before)


def _one_launch(dim, dyn, n, T, slab, tag):
    """ONE rollout launch of T ticks into a trajectory block, compared with the oracle on the device in slabs of `slab` ticks."""
    import torch
    from snac_amd import trajmem

    env, orc = _pair(dim, dyn, n, tag)
    nt = _threads()
    assert helpers.same_bytes(env.reset().cpu().numpy(), orc.reset())
    buf = trajmem.traj_empty((T, n, env.obs_dim), torch.float64, env.device)
    og, rg, dg = env.rollout(T, out=buf)                          # the launch bench.py times
    assert og.data_ptr() == buf.data_ptr()
    t = 0
    while t < T:
        c = min(slab, T - t)
        oc, rc, dc = orc.rollout(c, t0=t, nthreads=nt)
        assert torch.equal(og[t:t + c], torch.from_numpy(oc).to(env.device)), ("obs", t)
        assert torch.equal(rg[t:t + c], torch.from_numpy(rc).to(env.device)), ("reward", t)
        assert torch.equal(dg[t:t + c].view(torch.uint8), torch.from_numpy(dc).to(env.device)), ("done", t)
        t += c
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
    assert helpers.same_bytes(env.iou().cpu().numpy(), orc.iou())
    assert helpers.same_bytes(env.observe().cpu().numpy(), og[T - 1].cpu().numpy())


def test_config3_headline_pass_as_one_launch():
    """The exact launch of the headline measurement: rollout(600) of 65 536 2D dynamic dense envs, one launch, 16 GB of rows in a
    trajectory block -- all 39 321 600 rows, rewards and done flags equal the oracle's."""
    _one_launch(2, True, 65536, 600, 40, "dense_train")


def test_config5_3d_pass_as_one_launch():
    """BASELINE configs[4] as one launch of the 3D block kernel (k_rollout3db): rollout(1000) of 16 384 envs."""
    _one_launch(3, True, 16384, 1000, 125, "dense_train")


BIG = [
    # kind, N, T, float32 rows, kernel of the whole batch: every output below holds more than 2^31 values
    (2, 131072, 330, True, "k_rollout2d"),                           # 2.2e9 values, 8.8 GB
    (2, 8192, 5400, True, "k_rollout2dt"),                           # 2.26e9 values: 85 chunks of 64 ticks, rows [T][N] beyond 2^31 values
    (2, 14336, 3100, True, "k_rollout2dt"),
    (2, 24576, 1800, True, "k_rollout2db"),                          # blocks of 128 envs (round 5)
    (2, 24576, 1800, False, "k_rollout2db"),                         # 18 GB
    (2, 24578, 1800, False, "k_rollout"),                            # N % 4 != 0: the tile kernel, 18 GB
    (3, 65536, 660, True, "k_rollout3db"),                           # 2.21e9 values
    (3, 4088, 10900, True, "k_rollout3d"),                           # 2.27e9 values
]


@pytest.mark.parametrize("kind,n,T,f32,kernel", BIG, ids=lambda v: str(v))
def test_outputs_of_more_than_2_31_values(kind, n, T, f32, kernel):
    """Element and row indices beyond 2^31 (the headline's output is 2.0e9 values: just below).  The whole batch in one launch against
    the same envs stepped as two half batches (env_id_base shifts the counter-RNG ids; each half's output is below 2^31 values, sizes
    the other tests cover against the oracle), compared on the device in slabs of ticks; the last slab and the end state also
    against the oracle for a window of envs at the far end of the batch."""
    import torch
    from snac_amd import BatchedDMPEnv, _lib

    dt = torch.float32 if f32 else torch.float64
    table = helpers.plan_table(kind, True, "dense_train")
    full = table.reshape(len(table), 26, 26)
    h = n // 2
    big = BatchedDMPEnv(kind, True, n, plans=full, seed=13, obs_dtype=dt)
    lo = BatchedDMPEnv(kind, True, h, plans=full, seed=13, obs_dtype=dt)
    hi = BatchedDMPEnv(kind, True, h, plans=full, seed=13, obs_dtype=dt, env_id_base=h)
    for e in (big, lo, hi):
        e.reset()
    ob, rb, db = big.rollout(T)
    assert _lib.lib().snac_last_kernel().decode() == kernel
    assert ob.numel() > 2 ** 31
    ol, rl, dl = lo.rollout(T)
    oh, rh, dh = hi.rollout(T)
    step = max(1, (1 << 28) // (n * big.obs_dim))
    for t0 in range(0, T, step):
        t1 = min(T, t0 + step)
        assert torch.equal(ob[t0:t1, :h], ol[t0:t1]) and torch.equal(ob[t0:t1, h:], oh[t0:t1]), (t0, t1)
    assert torch.equal(rb[:, :h], rl) and torch.equal(rb[:, h:], rh) and torch.equal(db[:, :h], dl) and torch.equal(db[:, h:], dh)
    assert torch.equal(big._hdr[:h], lo._hdr) and torch.equal(big._hdr[h:], hi._hdr)
    assert torch.equal(big._grid[:h], lo._grid) and torch.equal(big._grid[h:], hi._grid)
    # the far corner against the oracle: the last 64 envs over the last ticks (the oracle steps them from the start, keeps the tail)
    w = 64
    orc = helpers.oracle().OracleBatch(kind, True, w, table, seed=13, env_id_base=n - w)
    orc.reset()
    keep = min(T, 40)
    if T > keep:
        orc.rollout(T - keep, t0=0, obs=None, nthreads=16)
    oc, rc, dc = orc.rollout(keep, t0=T - keep, nthreads=16)
    want = oc.astype(np.float32) if f32 else oc
    assert ob[T - keep:, n - w:].cpu().numpy().tobytes() == want.tobytes()
    assert rb[T - keep:, n - w:].cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(db[T - keep:, n - w:].cpu().numpy().view(np.uint8), dc)
