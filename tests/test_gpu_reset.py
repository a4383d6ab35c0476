"""GPU: k_reset (round 6) -- snac_reset / snac_reset_scalar of EVERY env of a batch (no mask, canonical layout, N % 4 == 0, aligned or no
observation output, from 256 envs): header from K::reset, records zeroed as one run per wave, the constant reset window as rows.  Against
the CPU oracle for the six classes: dirty state (envs mid-episode, some pending a reset), ragged last waves, float64 / float32, the plan row
from the counter RNG, from explicit indices and from the scalar form, resets without an observation, and the state AFTER the reset by
stepping on; masked resets and observe() on the step kernels' AUX forms; an odd batch and an unaligned output
stay on the tile kernel with the same results."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

CONFIGS = [(1, False), (1, True), (2, False), (2, True), (3, False), (3, True)]


def _kernel():
    from snac_amd import _lib

    return _lib.lib().snac_last_kernel().decode()


def _pair(dim, dyn, n, seed, f32=False, total_step=30):
    import torch
    from snac_amd import BatchedDMPEnv

    tag = {1: "sin_train", 2: "dense_train", 3: "dense_train"}[dim] if dyn else {1: "p1", 2: "p0", 3: "p1"}[dim]
    table = helpers.plan_table(dim, dyn, tag)
    full = table.reshape(len(table), 30) if dim == 1 else table.reshape(len(table), 26, 26)
    env = BatchedDMPEnv(dim, dyn, n, plans=full, seed=seed, total_step=total_step, obs_dtype=torch.float32 if f32 else torch.float64)
    orc = helpers.oracle().OracleBatch(dim, dyn, n, table, seed=seed)
    orc.set_total_step(total_step)
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    return env, orc, cast, len(table)


def _state_equal(env, orc):
    n = env.num_envs
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(n, -1), st["grid"])
    for name, key in (("count_brick", "cb"), ("count_step", "cs"), ("plan_idx", "plan_idx"), ("episode", "episode"), ("episode_return", "ep_return")):
        assert np.array_equal(getattr(env, name).cpu().numpy(), st[key]), name
    assert np.array_equal(env.need_reset.cpu().numpy().astype(np.uint8), st["need_reset"])


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dim,dyn", CONFIGS)
def test_whole_batch_resets_match_the_oracle(dim, dyn, f32):
    """16 384 + 36 envs (3D: 4096 + 36): reset, 20 ticks (no auto-reset: some envs end up pending), reset again (counter-RNG plan rows of
    the second episode), 5 ticks, a reset with explicit plan rows, 5 ticks, a reset without an observation, 5 ticks."""
    import torch

    n = (4096 if dim == 3 else 16384) + 36
    env, orc, cast, P = _pair(dim, dyn, n, seed=13, f32=f32, total_step=15)
    rng = np.random.default_rng(dim * 7 + dyn)
    t = 0
    for phase in range(4):
        if phase == 2:
            pidx = rng.integers(0, P, n).astype(np.int32)
            og = env.reset(plan_idx=torch.from_numpy(pidx).cuda())
            oc = orc.reset(plan_idx=pidx)
        elif phase == 3:
            og = env.reset(want_obs=False)
            oc = orc.reset()
        else:
            og = env.reset()
            oc = orc.reset()
        assert _kernel() == "k_reset", phase
        if og is not None:
            assert helpers.same_bytes(og.cpu().numpy(), cast(oc)), phase
        _state_equal(env, orc)
        for _ in range(20 if phase == 0 else 5):
            o, r, d = env.step(auto_reset=False)
            oo, ro, do = orc.step(t, None, None, auto_reset=False, nthreads=8)
            assert helpers.same_bytes(o.cpu().numpy(), cast(oo)), (phase, t)
            assert helpers.same_bytes(r.cpu().numpy(), ro) and np.array_equal(d.cpu().numpy().view(np.uint8), do), (phase, t)
            t += 1
    _state_equal(env, orc)


@pytest.mark.parametrize("dim,dyn", [(1, True), (1, False), (2, True), (2, False), (3, False), (3, True)])
def test_masked_resets_observe_and_what_stays_on_the_tile_kernel(dim, dyn):
    """A masked reset (the other envs report their current observation) and observe() on the step kernels' AUX forms (no action, no rules; a
    header written only for an env that was reset); an odd batch and an unaligned observation output: the tile kernel k_aux, with the same rows
    as k_reset gives a twin where both apply."""
    import torch

    n = 2048
    env, orc, cast, P = _pair(dim, dyn, n, seed=3)
    env.reset(); orc.reset()
    assert _kernel() == "k_reset"
    for t in range(12):
        env.step(auto_reset=False); orc.step(t, None, None, auto_reset=False)
    last = None
    for t in range(12, 15):
        last = env.step(auto_reset=False)[0]; orc.step(t, None, None, auto_reset=False)
    aux = {1: "k_step1d", 2: "k_step2d", 3: "k_step3dq"}[dim]      # the step kernels' AUX forms: a masked reset, an observe
    assert torch.equal(env.observe(), last) and _kernel() == aux
    mask = np.random.default_rng(1).random(n) < 0.3
    og = env.reset(mask=torch.from_numpy(mask).cuda())
    assert _kernel() == aux
    oc = orc.reset(mask=mask.astype(np.uint8))
    assert helpers.same_bytes(og.cpu().numpy(), oc)
    _state_equal(env, orc)
    assert torch.equal(env.observe(), og)
    pidx = np.random.default_rng(2).integers(0, P, n).astype(np.int32)
    mask2 = np.random.default_rng(3).random(n) < 0.5
    og = env.reset(mask=torch.from_numpy(mask2).cuda(), plan_idx=torch.from_numpy(pidx).cuda())
    assert _kernel() == aux
    oc = orc.reset(mask=mask2.astype(np.uint8), plan_idx=pidx)
    assert helpers.same_bytes(og.cpu().numpy(), oc)
    _state_equal(env, orc)
    for t in range(15, 20):                                          # ... and the state after it steps like the oracle's
        o, r, d = env.step(auto_reset=True); oo, ro, do = orc.step(t, None, None, auto_reset=True)
        assert helpers.same_bytes(o.cpu().numpy(), oo) and helpers.same_bytes(r.cpu().numpy(), ro), t
    twin = env.fork(torch.arange(n, device=env.device))
    raw = torch.empty(n * env.obs_dim + 1, dtype=torch.float64, device="cuda")
    with torch.cuda.device(env.device):
        import ctypes as C
        from snac_amd import _lib
        rows = raw[1:].view(n, env.obs_dim)
        _lib.check(env._lib.snac_reset(C.byref(env._desc), C.byref(env._state), None, None, C.c_void_p(rows.data_ptr()), env._stream()))
    assert _kernel() != "k_reset" and rows.data_ptr() % 16 != 0
    o2 = twin.reset()
    assert _kernel() == "k_reset"
    assert torch.equal(rows, o2) and torch.equal(env._hdr, twin._hdr) and torch.equal(env._grid, twin._grid) and torch.equal(env._episode, twin._episode)
    odd, orc2, cast2, _ = _pair(dim, dyn, 1001, seed=4)
    assert helpers.same_bytes(odd.reset().cpu().numpy(), orc2.reset())
    assert _kernel() != "k_reset"


def test_the_scalar_form_and_static_plan_tables():
    """snac_reset_scalar (one plan row for every env, by value) on the same kernel; a static class keeps row 0."""
    import torch

    n = 4096
    env, orc, cast, P = _pair(2, True, n, seed=5)
    env.reset(); orc.reset()
    with torch.cuda.device(env.device):
        import ctypes as C
        from snac_amd import _lib
        rows = torch.empty((n, env.obs_dim), dtype=torch.float64, device="cuda")
        _lib.check(env._lib.snac_reset_scalar(C.byref(env._desc), C.byref(env._state), 17, C.c_void_p(rows.data_ptr()), env._stream()))
    assert _kernel() == "k_reset"
    oc = orc.reset(plan_idx=np.full(n, 17, np.int32))
    assert helpers.same_bytes(rows.cpu().numpy(), oc)
    _state_equal(env, orc)
    sta, orc_s, _, _ = _pair(3, False, n, seed=6)
    assert helpers.same_bytes(sta.reset().cpu().numpy(), orc_s.reset()) and _kernel() == "k_reset"
    assert int(sta.plan_idx.max()) == 0
    _state_equal(sta, orc_s)
