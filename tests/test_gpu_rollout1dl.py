"""GPU: k_rollout1dl (round 6) -- 1D rollouts of large batches with a lane per env (the headline kernel's shape: K1D::step per lane on the
wave's LDS image, the wave's rows of a tick as one run of 64 x 56 bytes through a staging tile).  It takes canonical rows, every row
written, N % 4 == 0 and an aligned output from 45 056 envs on (float32 rows: 36 864; SNAC_1D_LANE_MIN_*); here against the CPU oracle on both sides of the
threshold: ragged last tiles, float64 / float32, static / dynamic plans, the `>` rule bits, short episodes (many resets and plan changes per
launch), launches that continue each other, tile-major outputs, the record outputs and explicit inputs; what it does not take (odd N,
an unaligned output) stays on the kernels behind it with the same rows; the layout variants of large batches take its VARLD forms."""
import numpy as np
import pytest

import helpers
import test_gpu_rollout1d as base

pytestmark = pytest.mark.gpu


def _kernel():
    from snac_amd import _lib

    return _lib.lib().snac_last_kernel().decode()


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_large_batches_against_the_oracle(dyn, f32):
    """49 152 + 36 envs (a last tile of 36), episodes of at most 23 steps: 150 ticks in three launches that continue each other."""
    import torch

    n = 49152 + 36
    env, orc = base._pair(dyn, n, seed=11, total_step=23, obs_dtype=torch.float32 if f32 else torch.float64)
    t0 = 0
    for T in (64, 1, 85):
        base._compare(env, orc, T, t0, f32=f32)
        assert _kernel() == "k_rollout1dl"
        t0 += T
    base._end_state(env, orc)


@pytest.mark.parametrize("rules", [(True, False), (False, True), (True, True)], ids=str)
def test_rule_bits_and_default_episode_length(rules):
    """The `>` forms of both end tests; the kind's own time limit (750 ticks: the dynamic classes end by bricks first)."""
    env, orc = base._pair(True, 49152, seed=5, brick_gt=rules[0], time_gt=rules[1])
    base._compare(env, orc, 120, 0)
    assert _kernel() == "k_rollout1dl"
    base._end_state(env, orc)


@pytest.mark.parametrize("n", [45052, 45056, 65536 + 4, 131072])
def test_both_sides_of_the_threshold_and_tile_major_rows(n):
    """One batch size below the threshold (the time-parallel kernel) and three above: the oracle's rows either way, in the canonical and in
    the tile-major layout; an unaligned output and a layout variant leave the kernel."""
    import torch

    T = 70
    env, orc = base._pair(True, n, seed=3, total_step=40)
    twin = env.fork(torch.arange(n, device=env.device))
    third = env.fork(torch.arange(n, device=env.device))
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    og, rg, dg = env.rollout(T)
    assert _kernel() == ("k_rollout1dl" if n >= 45056 else "k_rollout1dt")
    assert helpers.same_bytes(og.cpu().numpy(), oc) and helpers.same_bytes(rg.cpu().numpy(), rc)
    assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
    ot, rt, dtt = twin.rollout(T, obs="tiled")
    assert _kernel() == ("k_rollout1dl" if n >= 45056 else "k_rollout1dt")
    assert helpers.same_bytes(twin.untile(ot).cpu().numpy(), oc) and torch.equal(rt, rg) and torch.equal(dtt, dg)
    raw = torch.empty(T * n * 7 + 1, dtype=torch.float64, device=env.device)
    ou, ru, du = third.rollout(T, out=raw[1:].view(T, n, 7))
    assert _kernel() != "k_rollout1dl" and ou.data_ptr() % 16 != 0
    assert helpers.same_bytes(ou.cpu().numpy(), oc) and torch.equal(ru, rg) and torch.equal(du, dg)
    base._end_state(env, orc)
    assert torch.equal(env._hdr, twin._hdr) and torch.equal(env._hdr, third._hdr) and torch.equal(env._grid, third._grid)


def test_record_outputs_fed_back_as_explicit_inputs():
    """The record outputs of a counter-RNG rollout (action, step size, plan row, first-step flag) fed back as explicit inputs -- the kernel's
    EXPL form, which asks for a tick's bytes one tick ahead -- reproduce rows, rewards and done flags, which equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv

    n, T = 49152 + 8, 130
    table, full = base._tables(True)
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
    assert _kernel() == "k_rollout1dl"
    ob, rb, db = b.rollout(T, actions=rec["actions"], step_size=rec["step_size"])
    assert _kernel() == "k_rollout1dl"
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    oc, rc, dc = orc.rollout(T, t0=0, nthreads=16)
    assert helpers.same_bytes(oa.cpu().numpy(), oc) and helpers.same_bytes(ra.cpu().numpy(), rc)
    assert np.array_equal(da.cpu().numpy().view(np.uint8), dc)
    first, done = rec["first"].cpu().numpy(), da.cpu().numpy()
    assert first[0].all() and np.array_equal(first[1:], done[:-1].astype(np.uint8))
    assert np.array_equal(rec["plan_idx"][-1].cpu().numpy(), a.plan_idx.cpu().numpy())
    # only the actions explicit (step sizes from the counter RNG), then only the step sizes
    c = BatchedDMPEnv(1, True, n, plans=full, seed=21, total_step=50)
    c.reset()
    oc2, _, _ = c.rollout(T, actions=rec["actions"])
    assert _kernel() == "k_rollout1dl" and torch.equal(oc2, oa)
    d = BatchedDMPEnv(1, True, n, plans=full, seed=21, total_step=50)
    d.reset()
    od, _, _ = d.rollout(T, step_size=rec["step_size"])
    assert torch.equal(od, oa)


def test_a_layout_variant_equals_the_canonical_rows_in_its_first_seven_values():
    """A layout variant of a large batch (the position appended) takes the kernel's VARLD form and equals the canonical rows in its first 7
    values."""
    import torch
    from snac_amd import BatchedDMPEnv

    n, T = 49152, 40
    table, full = base._tables(True)
    a = BatchedDMPEnv(1, True, n, plans=full, seed=9, total_step=30)
    b = BatchedDMPEnv(1, True, n, plans=full, seed=9, total_step=30, obs_tail=("position",))
    a.reset()
    b.reset()
    oa, ra, da = a.rollout(T)
    assert _kernel() == "k_rollout1dl"
    ob, rb, db = b.rollout(T)
    assert _kernel() == "k_rollout1dl"
    assert torch.equal(oa, ob[..., :7]) and torch.equal(ra, rb) and torch.equal(da, db)


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn,kw", base.VARIANTS_1D, ids=["lnet1d", "ppo", "record", "all"])
def test_layout_variants_of_large_batches(dyn, kw, f32):
    """The observation layouts of the reference's 1D env copies on k_rollout1dl's VARLD forms (staging tiles for rows of <= 16 / 38 / 46
    values): 49 152 + 36 envs, time limit 7 (plan changes every few ticks), launches of 1, 70 and 33 ticks, explicit inputs, the tile-major
    output -- against the oracle configured the same way; an unaligned output leaves the kernel with the same rows."""
    import torch
    from snac_amd import BatchedDMPEnv

    n = 49152 + 36
    table, full = base._tables(dyn)
    dt = torch.float32 if f32 else torch.float64
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    env = BatchedDMPEnv(1, dyn, n, plans=full, seed=4, total_step=7, obs_dtype=dt, **kw)
    orc = helpers.oracle().OracleBatch(1, dyn, n, table, seed=4)
    norm = {None: dyn, "raw": False, "norm": True}[env.obs_scalars]
    orc.configure(obs_norm=norm, frame=env.frame_value, tail=env.obs_tail)
    orc.set_total_step(7)
    assert helpers.same_bytes(env.reset().cpu().numpy(), cast(orc.reset()))
    t0 = 0
    for T in (1, 70, 33):
        og, rg, dg = env.rollout(T)
        assert _kernel() == "k_rollout1dl"
        oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
        assert helpers.same_bytes(og.cpu().numpy(), cast(oc)), ("observations", T)
        assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
        t0 += T
        del og, oc
    twin = env.fork(torch.arange(n, device=env.device))
    third = env.fork(torch.arange(n, device=env.device))
    rng = np.random.default_rng(1)
    T = 40
    acts, ks = rng.integers(0, 3, size=(T, n)).astype(np.int8), rng.integers(1, 4, size=(T, n)).astype(np.int8)
    ta, tk = torch.from_numpy(acts).to(env.device), torch.from_numpy(ks).to(env.device)
    og, rg, dg = env.rollout(T, actions=ta, step_size=tk)
    assert _kernel() == "k_rollout1dl"
    oc, rc, dc = orc.rollout(T, t0=t0, actions=acts, step_size=ks, nthreads=16)
    assert helpers.same_bytes(og.cpu().numpy(), cast(oc)) and helpers.same_bytes(rg.cpu().numpy(), rc)
    ot, rt, dtt = twin.rollout(T, actions=ta, step_size=tk, obs="tiled")
    assert _kernel() == "k_rollout1dl"
    assert torch.equal(twin.untile(ot), og) and torch.equal(rt, rg) and torch.equal(dtt, dg)
    raw = torch.empty(T * n * env.obs_dim + 1, dtype=dt, device=env.device)
    ou, ru, du = third.rollout(T, actions=ta, step_size=tk, out=raw[1:].view(T, n, env.obs_dim))
    assert _kernel() != "k_rollout1dl" and torch.equal(ou, og) and torch.equal(ru, rg) and torch.equal(du, dg)
    base._end_state(env, orc)
    assert torch.equal(env._hdr, twin._hdr) and torch.equal(env._hdr, third._hdr) and torch.equal(env._grid, third._grid)


def test_replay_rings_filled_by_the_lane_kernel():
    """ReplayRing.collect on a 1D batch of 45 056 + 36 envs: the tick ring and the tile-major ring (launches that write at an offset of the
    ring and wrap: snac_rollout_tiled's tiled_T / tiled_t0, with the record outputs) hold the same rows, records and samples; the last launch's
    rows equal the oracle's."""
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    n = 45056 + 36
    table, full = base._tables(True)
    envs = [BatchedDMPEnv(1, True, n, plans=full, seed=12, total_step=40) for _ in range(2)]
    orc = helpers.oracle().OracleBatch(1, True, n, table, seed=12)
    orc.set_total_step(40)
    orc.reset()
    orc.rollout(13, t0=0, obs=None, nthreads=16)
    rings = []
    for e, layout in zip(envs, ("ticks", "tiled")):
        e.reset()
        e.rollout(13, obs=None)                                       # attach in mid-episode
        rings.append(ReplayRing(e, 24, layout=layout))
    t0 = 13
    for T in (17, 20, 24, 7):                                         # launches that straddle the ring's end
        for r in rings:
            r.collect(T)
            assert _kernel() == "k_rollout1dl"
        oc, rc, dc = orc.rollout(T, t0=t0, nthreads=16)
        t0 += T
    a, b = rings
    for slot in range(24):
        assert torch.equal(a.obs_at(slot), b.obs_at(slot)), slot
    for name in ("reward", "done", "action", "step_size", "plan_idx", "first"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert a.head == b.head == (17 + 20 + 24 + 7) % 24
    for i in range(7):
        assert helpers.same_bytes(a.obs_at((a.head - 7 + i) % 24).cpu().numpy(), oc[i]), i
    ga, gb = torch.Generator(device="cuda"), torch.Generator(device="cuda")
    ga.manual_seed(3); gb.manual_seed(3)
    sa, sb = a.sample(300, generator=ga), b.sample(300, generator=gb)
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
