"""GPU: the tile form of snac_step (k_step2d / k_step3d, round 3: wide loads of the records / window rows, lane-per-env window
extraction, rows transposed through LDS, 16-byte-per-lane stores, episodic sums by atomics) against the CPU oracle, tick by tick.
The kernels take 2D / 3D steps on the identity rows when N % 4 == 0: a lone ragged tile, whole tiles, blocks with idle waves and a
ragged last tile; float64 (two staged halves) and float32; counter-RNG and explicit inputs; auto-reset on and off; no observation;
short time limits so that every env resets many times (3D random agents also box themselves in every ~22 steps); agents driven to
every edge of the plan area (the 3D window loads clamp their first column into the record)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

KINDS = [(2, False), (2, True), (3, False), (3, True)]


def _ids(v):
    return "%dd_%s" % (v[0], "dyn" if v[1] else "sta") if isinstance(v, tuple) else str(v)


def _pair(dim, dyn, n, seed, total_step=None, f32=False, base=0):
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(dim, dyn, "dense_train" if dyn else "p0")
    env = BatchedDMPEnv(dim, dyn, n, plans=table.reshape(len(table), 26, 26), seed=seed, env_id_base=base, total_step=total_step,
                        obs_dtype=torch.float32 if f32 else torch.float64)
    orc = helpers.oracle().OracleBatch(dim, dyn, n, table, seed=seed, env_id_base=base)
    if total_step:
        orc.set_total_step(total_step)
    o = orc.reset()
    assert helpers.same_bytes(env.reset().cpu().numpy(), (o.astype(np.float32) if f32 else o))
    return env, orc


def _end_state(env, orc):
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(env.num_envs, -1), st["grid"])
    for name, key in (("count_brick", "cb"), ("count_step", "cs"), ("plan_idx", "plan_idx"), ("episode", "episode"), ("episode_return", "ep_return"),
                      ("total_brick", "tb")):
        assert np.array_equal(getattr(env, name).cpu().numpy(), st[key]), name
    assert np.array_equal(env.position.cpu().numpy(), st["pos"])
    assert np.array_equal(env.need_reset.cpu().numpy().astype(np.uint8), st["need_reset"])
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
    assert helpers.same_bytes(env.iou().cpu().numpy(), orc.iou())


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", KINDS, ids=_ids)
@pytest.mark.parametrize("n", [4, 36, 64, 256 + 64 + 12, 4096 + 40])
def test_counter_rng_steps_with_auto_reset(kind, n, f32):
    dim, dyn = kind
    env, orc = _pair(dim, dyn, n, seed=6, total_step=25, f32=f32, base=123456789012)
    for t in range(70):
        og, rg, dg = env.step(auto_reset=True)
        oc, rc, dc = orc.step(t, auto_reset=True, nthreads=8)
        assert helpers.same_bytes(og.cpu().numpy(), (oc.astype(np.float32) if f32 else oc)), (t, "obs")
        assert helpers.same_bytes(rg.cpu().numpy(), rc), (t, "reward")
        assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc), (t, "done")
    _end_state(env, orc)
    assert env.episodic_stats()["episodes"] >= 2 * n


@pytest.mark.parametrize("kind", KINDS, ids=_ids)
def test_explicit_inputs_walk_to_every_edge(kind):
    """Each env repeats one direction for a while (so that agents reach and lean on all four borders and corners of the plan area),
    then drops / builds for a while; no auto-reset: envs past `done` keep mutating like the reference."""
    import torch

    dim, dyn = kind
    n, A = 512, helpers.DIMS[dim]["A"]
    env, orc = _pair(dim, dyn, n, seed=8)
    rng = np.random.default_rng(dim)
    phase = rng.integers(0, A, size=(12, n))
    for t in range(12 * 9):
        a = phase[t // 9].astype(np.int8)
        if t % 9 >= 6:                                            # the last third of a phase: random actions
            a = rng.integers(0, A, size=n).astype(np.int8)
        k = rng.integers(1, 4, size=n).astype(np.int8)
        og, rg, dg = env.step(torch.from_numpy(a), torch.from_numpy(k))
        oc, rc, dc = orc.step(t, a, k)
        assert helpers.same_bytes(og.cpu().numpy(), oc), (t, "obs")
        assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
    _end_state(env, orc)
    pos = env.position.cpu().numpy()
    assert pos.min() <= 4 and pos.max() >= 21


@pytest.mark.parametrize("kind", KINDS, ids=_ids)
def test_steps_without_observation_and_reused_outputs(kind):
    """want_obs=False (no staging at all) interleaved with steps that write into preallocated outputs."""
    import torch

    dim, dyn = kind
    n = 1024 + 8
    env, orc = _pair(dim, dyn, n, seed=9, total_step=20)
    out = (torch.empty((n, 51), dtype=torch.float64, device=env.device), torch.empty(n, dtype=torch.float32, device=env.device),
           torch.empty(n, dtype=torch.uint8, device=env.device))
    for t in range(60):
        blind = t % 3 == 1
        og, rg, dg = env.step(auto_reset=True, want_obs=not blind, out=out)
        oc, rc, dc = orc.step(t, auto_reset=True, nthreads=8)
        assert (og is None) if blind else helpers.same_bytes(og.cpu().numpy(), oc)
        assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
    _end_state(env, orc)


@pytest.mark.parametrize("kind", [(2, True), (3, True)], ids=_ids)
def test_large_batch_equals_the_rollout(kind):
    """N = 65 536 + 36: 80 per-tick steps (the tile step kernels) write the rows one fused rollout of a forked batch writes."""
    import torch

    dim, dyn = kind
    n, T = 65536 + 36, 80
    env, _ = _pair(dim, dyn, n, seed=10, total_step=30)
    twin = env.fork(torch.arange(n, device=env.device))
    ot, rt, dt = twin.rollout(T)
    for t in range(T):
        og, rg, dg = env.step(auto_reset=True)
        assert torch.equal(og, ot[t]) and torch.equal(rg, rt[t]) and torch.equal(dg, dt[t]), t
    assert torch.equal(env._hdr, twin._hdr) and torch.equal(env._grid, twin._grid) and torch.equal(env._stats, twin._stats)
    assert torch.equal(env._episode, twin._episode)


def test_fast_path_revalidates_tensors_that_were_changed_in_place():
    """step() skips its argument checks when it is handed the same tensor objects as the call before -- but only while their shapes,
    strides and dtypes are what was validated: a tensor resized or re-strided in place keeps its data_ptr(), and the kernel would
    read or write past it."""
    import torch
    from snac_amd import BatchedDMPEnv

    n = 256
    env = BatchedDMPEnv(2, True, n, seed=3)
    env.reset()
    acts = torch.zeros(n, dtype=torch.int8, device=env.device)
    ks = torch.ones(n, dtype=torch.int8, device=env.device)
    base = torch.empty(2 * n * 51, dtype=torch.float64, device=env.device)
    out = (base[: n * 51].view(n, 51), torch.empty(n, dtype=torch.float32, device=env.device),
           torch.empty(n, dtype=torch.uint8, device=env.device))
    env.step(acts, ks, auto_reset=True, out=out)
    assert env._fast is not None
    env.step(acts, ks, auto_reset=True, out=out)                     # the fast path
    t_before = env.t
    acts.resize_(n // 2)                                             # same object, same data_ptr, half the elements
    with pytest.raises(ValueError):
        env.step(acts, ks, auto_reset=True, out=out)
    assert env.t == t_before
    acts.resize_(n)
    env.step(acts, ks, auto_reset=True, out=out)
    out[0].as_strided_((n, 51), (102, 1))                            # same pointer, rows twice as far apart
    with pytest.raises(ValueError):
        env.step(acts, ks, auto_reset=True, out=out)
    out[0].as_strided_((n, 51), (51, 1))
    env.step(acts, ks, auto_reset=True, out=out)
    assert env.t == t_before + 2


STEP_VARIANTS = [
    (2, False, dict(layout="lnet2d")),                               # 53 values, frame value 2, normalised scalars on a static plan
    (2, True, dict(layout="ppo")),                                   # 451 values: window, raw counters, the 400 plan cells
    (2, True, dict(obs_tail=("position", "plan", "record"), frame_value=2, obs_scalars="raw")),   # 461
    (3, True, dict(layout="ppo")),                                   # 3D: 451 values, the plan tail is the 400 plan heights
    (3, False, dict(obs_tail=("position", "plan", "record"), obs_scalars="norm")),   # 461 on the static plan
    (3, True, dict(obs_tail=("position", "record"))),                # 61: no plan tail
]


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dim,dyn,kw", STEP_VARIANTS, ids=["lnet2d", "ppo", "all", "ppo3d", "all3d", "short3d"])
def test_layout_variants_of_large_batches_step_on_the_tile_form(dim, dyn, kw, f32):
    """snac_step with a layout variant (rows of 53 .. 461 values) takes k_step2d<.., VAR> from 45 056 envs (rows with the plan tail;
    float32 rows: 32 768) / 24 576 (short rows): 40 ticks with auto-reset at a time limit of 9 against the oracle configured the same
    way -- counter-RNG ticks, then explicit inputs --, a ragged last tile (45 056 + 36 envs); a batch below the limits on k_transition
    (whole plans staged in LDS for the plan tail), the same oracle."""
    import torch
    from snac_amd import BatchedDMPEnv, _lib

    n = (45056 if dim == 2 else 24576) + 36                          # 3D: k_step3d<.., VAR> from 24 576 envs
    table = helpers.plan_table(dim, dyn, "dense_train" if dyn else "p0")
    dt = torch.float32 if f32 else torch.float64
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    A = 5 if dim == 2 else 8
    sizes = [(n, "k_step%dd" % dim), (4100, "k_transition")]
    if dim == 2 and (kw.get("layout") == "ppo" or "plan" in (kw.get("obs_tail") or ())):
        sizes.append((32768 - 28, "k_step2d"))                       # half-filled tiles (rows with the plan tail, 24 577 .. 32 768 envs), a ragged last one
    for nn, kern in sizes:
        env = BatchedDMPEnv(dim, dyn, nn, plans=table.reshape(len(table), 26, 26), seed=3, total_step=9, obs_dtype=dt, **kw)
        orc = helpers.oracle().OracleBatch(dim, dyn, nn, table, seed=3)
        orc.configure(obs_norm={None: dyn, "raw": False, "norm": True}[env.obs_scalars], frame=env.frame_value, tail=env.obs_tail)
        orc.set_total_step(9)
        assert helpers.same_bytes(env.reset().cpu().numpy(), cast(orc.reset()))
        rng = np.random.default_rng(2)
        for t in range(40 if nn == n else (20 if nn > 20000 else 12)):
            if t % 3 == 2:
                acts, ks = rng.integers(0, A, nn).astype(np.int8), rng.integers(1, 4, nn).astype(np.int8)
                og, rg, dg = env.step(torch.from_numpy(acts).to(env.device), torch.from_numpy(ks).to(env.device), auto_reset=True)
                oc, rc, dc = orc.step(t, acts, ks, auto_reset=True, nthreads=16)
            else:
                og, rg, dg = env.step(auto_reset=True)
                oc, rc, dc = orc.step(t, auto_reset=True, nthreads=16)
            assert _lib.lib().snac_last_kernel().decode() == kern
            assert helpers.same_bytes(og.cpu().numpy(), cast(oc)), (t, "obs")
            assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
        s, e = orc.stats(), env.episodic_stats()
        assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))
        assert helpers.same_bytes(env.iou().cpu().numpy(), orc.iou())


@pytest.mark.parametrize("n", [20476, 20480, 278528, 278532, 475136, 475140])
def test_2d_records_by_plain_and_by_non_temporal_loads(n):
    """Round 6: k_step2d picks a form by batch size -- non-temporal record loads + plain rows below 20 480 envs; the "resident" form (plain
    loads keep the state in the Infinity Cache, non-temporal rows stay out of it) from 20 480 to 278 528 envs and again above 475 136;
    plain loads + plain rows in between -- the same rows on either side of every threshold, against the oracle."""
    from snac_amd import _lib

    env, orc = _pair(2, True, n, seed=8, total_step=12)
    for t in range(6):
        og, rg, dg = env.step(auto_reset=True)
        assert _lib.lib().snac_last_kernel() == b"k_step2d"
        oc, rc, dc = orc.step(t, auto_reset=True, nthreads=16)
        assert helpers.same_bytes(og.cpu().numpy(), oc), t
        assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
    _end_state(env, orc)
