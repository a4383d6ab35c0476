"""GPU: k_step1d (round 6) -- the canonical 1D snac_step on identity rows: 64 envs per wave, the records by wide loads into K1D's bordered
rows in LDS, rules1d() per lane, the rows of a wave as one run of 64 x 56 bytes (rows1d.h).  It takes N % 4 == 0 and an aligned (or no)
observation output from 256 envs on; against the CPU oracle: ragged last waves, float64 / float32, static / dynamic plans, counter RNG and
explicit inputs biased to moves and to drops, scalar inputs, steps without observations, auto-reset with plan changes (short episodes), the
`>` rule bits, manual reset(mask) between steps; what it does not take (odd N, an unaligned output) stays on the tile kernel, same rows.  Also k_edges1d, the same shape for 1D tree edges with gathered rows."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _kernel():
    from snac_amd import _lib

    return _lib.lib().snac_last_kernel().decode()


def _pair(dyn, n, seed, f32=False, total_step=40, brick_gt=False, time_gt=False):
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(1, dyn, "sin_train" if dyn else "p1")
    env = BatchedDMPEnv(1, dyn, n, plans=table.reshape(len(table), 30), seed=seed, total_step=total_step, obs_dtype=torch.float32 if f32 else torch.float64,
                        brick_gt=brick_gt, time_gt=time_gt)
    orc = helpers.oracle().OracleBatch(1, dyn, n, table, seed=seed)
    orc.set_total_step(total_step)
    orc.set_rules(brick_gt, time_gt)
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    assert helpers.same_bytes(env.reset().cpu().numpy(), cast(orc.reset()))
    return env, orc, cast


def _walk(env, orc, cast, ticks, rng, kernel="k_step1d", explicit_from=15, probs=(0.3, 0.3, 0.4), t0=0):
    import torch

    n = env.num_envs
    out = (torch.empty((n, 7), dtype=env.obs_dtype, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda"))
    for t in range(t0, t0 + ticks):
        a = k = None
        if t >= explicit_from:
            a = rng.choice(3, size=n, p=list(probs)).astype(np.int8)
            k = rng.integers(1, 4, size=n).astype(np.int8)
        og, rg, dg = env.step(None if a is None else torch.from_numpy(a).cuda(), None if k is None else torch.from_numpy(k).cuda(), auto_reset=True, out=out)
        assert _kernel() == kernel
        oc, rc, dc = orc.step(t, a, k, auto_reset=True, nthreads=8)
        assert helpers.same_bytes(og.cpu().numpy(), cast(oc)), t
        assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
    _end(env, orc)


def _end(env, orc):
    n = env.num_envs
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(n, -1), st["grid"])
    for name, key in (("count_brick", "cb"), ("count_step", "cs"), ("plan_idx", "plan_idx"), ("episode", "episode"), ("episode_return", "ep_return")):
        assert np.array_equal(getattr(env, name).cpu().numpy(), st[key]), name
    assert np.array_equal(env.need_reset.cpu().numpy().astype(np.uint8), st["need_reset"])
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_large_batches_step_like_the_oracle(dyn, f32):
    """N = 65 536 + 36 (a last wave of 36 envs): 60 ticks with auto-reset (episodes of at most 40 steps: every env starts over, the dynamic
    ones on another plan row) -- counter RNG, then explicit actions."""
    env, orc, cast = _pair(dyn, 65536 + 36, 31, f32)
    _walk(env, orc, cast, 60, np.random.default_rng(8))


@pytest.mark.parametrize("n", [256, 260, 1000, 4096 + 12])
def test_small_batches_ragged_waves_and_the_rule_bits(n):
    """Four waves and a bit, ragged last waves; drops only (episodes end by count_brick on the dynamic plans), moves only (the agent walks
    into both ends of the row: the frame cells of the window), the `>` forms of both end tests."""
    rng = np.random.default_rng(n)
    env, orc, cast = _pair(True, n, 7, False, total_step=25)
    _walk(env, orc, cast, 40, rng, explicit_from=10, probs=(0.05, 0.05, 0.9))
    env2, orc2, cast2 = _pair(False, n, 9, True, total_step=70)
    _walk(env2, orc2, cast2, 80, rng, explicit_from=0, probs=(0.55, 0.4, 0.05))
    env3, orc3, cast3 = _pair(True, n, 11, False, total_step=12, brick_gt=True, time_gt=True)
    _walk(env3, orc3, cast3, 30, rng, explicit_from=5)


def test_steps_without_observations_scalar_inputs_and_manual_resets():
    """want_obs=False steps (the kernel returns before the rows), snac_step_scalar's by-value action / step size (no reward / done outputs),
    reset(mask) between steps, and steps that write rows again: the state carried through equals the oracle's."""
    import torch

    n = 2048 + 4
    env, orc, cast = _pair(True, n, 3, False, total_step=30)
    rng = np.random.default_rng(2)
    for t in range(60):
        if t % 10 == 9:
            mask = rng.random(n) < 0.2
            og = env.reset(mask=torch.from_numpy(mask).cuda())
            oc = orc.reset(mask=mask.astype(np.uint8))
            assert helpers.same_bytes(og.cpu().numpy(), oc), t
        if t % 3 == 0:
            a = rng.choice(3, size=n, p=[0.2, 0.2, 0.6]).astype(np.int8)
            k = rng.integers(1, 4, size=n).astype(np.int8)
            og, rg, dg = env.step(torch.from_numpy(a).cuda(), torch.from_numpy(k).cuda(), auto_reset=True, want_obs=False)
            assert og is None and _kernel() == "k_step1d"
            oc, rc, dc = orc.step(t, a, k, auto_reset=True, want_obs=False)
            assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
        elif t % 3 == 1:
            act, kk = (2 if t % 2 else 1), 1 + t % 3
            og = env.step_scalar(act, kk, auto_reset=True)
            assert _kernel() == "k_step1d"
            oc, rc, dc = orc.step(t, np.full(n, act, np.int8), np.full(n, kk, np.int8), auto_reset=True)
            assert helpers.same_bytes(og.cpu().numpy(), oc), t
        else:
            og, rg, dg = env.step(auto_reset=True)
            assert _kernel() == "k_step1d"
            oc, rc, dc = orc.step(t, None, None, auto_reset=True)
            assert helpers.same_bytes(og.cpu().numpy(), oc), t
            assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
    _end(env, orc)


def test_what_the_kernel_does_not_take_steps_the_same_on_the_tile_kernel():
    """An odd batch and an unaligned observation output stay on k_transition; no auto-reset: envs stepped past `done` keep counting
    (saturating counters are the tile kernel's tests; here 20 steps past the end against the oracle)."""
    import torch

    env, orc, cast = _pair(True, 1001, 5, False, total_step=20)
    _walk(env, orc, cast, 25, np.random.default_rng(1), kernel="k_transition", explicit_from=5)
    n = 512
    env, orc, cast = _pair(False, n, 6, False, total_step=8)
    raw = torch.empty(n * 7 + 1, dtype=torch.float64, device="cuda")
    out = (raw[1:].view(n, 7), torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda"))
    for t in range(12):
        og, rg, dg = env.step(auto_reset=False, out=out)
        assert _kernel() == "k_transition"
        oc, rc, dc = orc.step(t, None, None, auto_reset=False)
        assert helpers.same_bytes(og.cpu().numpy(), oc) and helpers.same_bytes(rg.cpu().numpy(), rc), t
    twin = _pair(False, n, 6, False, total_step=8)
    for t in range(28):                                              # ... and on k_step1d: 20 steps past `done` without a reset
        og, rg, dg = twin[0].step(auto_reset=False)
        assert _kernel() == "k_step1d"
        oc, rc, dc = twin[1].step(t, None, None, auto_reset=False)
        assert helpers.same_bytes(og.cpu().numpy(), oc) and helpers.same_bytes(rg.cpu().numpy(), rc), t
        assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
    _end(twin[0], twin[1])


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_large_waves_of_1d_tree_edges(dyn, f32):
    """k_edges1d: a wave of 65 536 + 36 edges on a pool of 2^18 rows -- every source record fetched once by four neighbouring lanes, through
    K1D's bordered rows in LDS, out to its destination row the same way, rows as one run per wave.  Shared random parents from the lower
    half, distinct children in the upper half, some edges in place; against the oracle, and a second wave on the children; m % 4 != 0 takes
    the same kernel's value-by-value rows; 40 edges (below SNAC_EDGES1D_MIN) the tile kernel: the same rows for the same edges."""
    import torch
    from snac_amd import BatchedDMPEnv

    pool, m = 1 << 18, 65536 + 36
    table = helpers.plan_table(1, dyn, "sin_train" if dyn else "p1")
    env = BatchedDMPEnv(1, dyn, pool, plans=table.reshape(len(table), 30), seed=21, obs_dtype=torch.float32 if f32 else torch.float64)
    orc = helpers.oracle().OracleBatch(1, dyn, pool, table, seed=21)
    env.reset(); orc.reset()
    env.rollout(25, obs=None); orc.rollout(25, obs=None, nthreads=16)
    rng = np.random.default_rng(5)
    for wave in range(2):
        dst = (pool // 2 + rng.choice(pool // 2, m, replace=False)).astype(np.int32)
        src = rng.integers(0, pool // 2, m).astype(np.int32)
        inplace = rng.random(m) < 0.1
        src = np.where(inplace, dst, src).astype(np.int32)
        acts = rng.integers(0, 3, m).astype(np.int8)
        ks = rng.integers(1, 4, m).astype(np.int8) if wave == 0 else None
        o, r, d = env.transition(acts, ks, src, dst, t=wave)
        assert _kernel() == "k_edges1d"
        oo, ro, do = orc.transition(acts, ks, src, dst, t=wave)
        assert helpers.same_bytes(o.cpu().numpy(), oo.astype(np.float32) if f32 else oo), wave
        assert helpers.same_bytes(r.cpu().numpy(), ro) and np.array_equal(d.cpu().numpy().astype(np.uint8), do), wave
    assert helpers.same_bytes(env.iou().cpu().numpy(), orc.iou())
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(pool, -1), st["grid"])
    for name, key in (("count_brick", "cb"), ("count_step", "cs"), ("plan_idx", "plan_idx")):
        assert np.array_equal(getattr(env, name).cpu().numpy(), st[key]), name
    # m % 4 != 0 (rows value by value) and a small wave (the tile kernel): the same rows and records for the same edges, compared on the device
    twin = env.fork(torch.arange(pool, device=env.device))
    third = env.fork(torch.arange(pool, device=env.device))
    dst = (pool // 2 + rng.choice(pool // 2, m, replace=False)).astype(np.int32)
    src = rng.integers(0, pool // 2, m).astype(np.int32)
    acts = rng.integers(0, 3, m).astype(np.int8)
    o1, r1, d1 = env.transition(acts, None, src, dst, t=7)
    o2, r2, d2 = twin.transition(acts[: m - 2], None, src[: m - 2], dst[: m - 2], t=7)
    assert _kernel() == "k_edges1d"
    assert torch.equal(o1[: m - 2], o2) and torch.equal(r1[: m - 2], r2) and torch.equal(d1[: m - 2], d2)
    keep = torch.from_numpy(dst[: m - 2].astype(np.int64)).to(env.device)
    assert torch.equal(env._grid[keep], twin._grid[keep]) and torch.equal(env._hdr[keep], twin._hdr[keep]) and torch.equal(env._episode[keep], twin._episode[keep])
    o3, r3, d3 = third.transition(acts[:40], None, src[:40], dst[:40], t=7)
    assert _kernel() == "k_transition"
    assert torch.equal(o1[:40], o3) and torch.equal(r1[:40], r3) and torch.equal(d1[:40], d3)
    k40 = torch.from_numpy(dst[:40].astype(np.int64)).to(env.device)
    assert torch.equal(env._grid[k40], third._grid[k40]) and torch.equal(env._hdr[k40], third._hdr[k40])


VARIANTS_1D = [
    (False, dict(layout="lnet1d")),                                                     # 8 values: the position appended
    (True, dict(layout="ppo")),                                                         # 37: window, counters, the 30 plan heights
    (True, dict(obs_tail=("record",), obs_scalars="raw")),                              # 15
    (False, dict(obs_tail=("position", "plan", "record"), frame_value=2, obs_scalars="norm")),   # 46
]


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn,kw", VARIANTS_1D, ids=["lnet1d", "ppo", "record", "all"])
def test_layout_variants_step_on_the_same_kernel(dyn, kw, f32):
    """The observation layouts of the reference's 1D env copies (frame value, raw / normalised counters, the position, the plan's heights and
    the record appended) through k_step1d's VAR form (from 256 envs, N % 4 == 0): ragged last waves, short episodes (time limit 9: plan changes every few
    steps), counter RNG and explicit inputs, against the oracle configured the same way; N % 4 != 0 stays on the tile kernel, same rows."""
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(1, dyn, "sin_train" if dyn else "p1")
    dt = torch.float32 if f32 else torch.float64
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    for n, kernel in ((4096 + 36, "k_step1d"), (260, "k_step1d"), (1002, "k_transition")):
        env = BatchedDMPEnv(1, dyn, n, plans=table.reshape(len(table), 30), seed=4, total_step=9, obs_dtype=dt, **kw)
        orc = helpers.oracle().OracleBatch(1, dyn, n, table, seed=4)
        norm = {None: dyn, "raw": False, "norm": True}[env.obs_scalars]
        orc.configure(obs_norm=norm, frame=env.frame_value, tail=env.obs_tail)
        orc.set_total_step(9)
        assert helpers.same_bytes(env.reset().cpu().numpy(), cast(orc.reset()))
        rng = np.random.default_rng(n)
        out = (torch.empty((n, env.obs_dim), dtype=dt, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda"))
        for t in range(30):
            a = k = None
            if t >= 12:
                a = rng.choice(3, size=n, p=[0.25, 0.25, 0.5]).astype(np.int8)
                k = rng.integers(1, 4, size=n).astype(np.int8)
            og, rg, dg = env.step(None if a is None else torch.from_numpy(a).cuda(), None if k is None else torch.from_numpy(k).cuda(), auto_reset=True, out=out)
            assert _kernel() == kernel
            oc, rc, dc = orc.step(t, a, k, auto_reset=True, nthreads=8)
            assert helpers.same_bytes(og.cpu().numpy(), cast(oc)), (n, t)
            assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), (n, t)
        _end(env, orc)
