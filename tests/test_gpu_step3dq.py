"""GPU: k_step3dq (round 5) -- the canonical 3D snac_step on identity rows, 16 envs per wave and four lanes per env: the env's four lanes
step it redundantly, the wave's lanes fetch the 16 spans of ten rows together (pieces the tick cannot touch skipped), the four lanes of an
env extract its 49 window cells between them and the 16 rows leave as one run.  It takes every batch size (N % 4 = 0, aligned obs):
against the CPU oracle from 4 envs to 98 340, ragged last waves (4 / 8 / 12 envs), float64 and float32 rows, counter RNG and explicit
inputs biased to every edge, steps without observations, auto-reset; tests/test_gpu_step_tile.py and tests/test_gpu_property.py run on it
too (their 3D cases).  The kernels it replaced are tested in a child process (tests/test_gpu_step3ds.py)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _pair(dyn, n, seed, f32, total_step=40):
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(3, dyn, "dense_train" if dyn else "p1")
    env = BatchedDMPEnv(3, dyn, n, plans=table.reshape(len(table), 26, 26), seed=seed, total_step=total_step, obs_dtype=torch.float32 if f32 else torch.float64)
    orc = helpers.oracle().OracleBatch(3, dyn, n, table, seed=seed)
    orc.set_total_step(total_step)
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    assert helpers.same_bytes(env.reset().cpu().numpy(), cast(orc.reset()))
    return env, orc, cast


def _walk(env, orc, cast, ticks, rng, kernel=b"k_step3dq", explicit_from=15, probs=None):
    import torch
    from snac_amd import _lib

    n = env.num_envs
    out = (torch.empty((n, 51), dtype=env.obs_dtype, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda"))
    for t in range(ticks):
        a = k = None
        if t >= explicit_from:
            a = rng.choice(8, size=n, p=probs or [0.08, 0.08, 0.27, 0.27, 0.075, 0.075, 0.075, 0.075]).astype(np.int8)
            k = rng.integers(1, 4, size=n).astype(np.int8)
        og, rg, dg = env.step(None if a is None else torch.from_numpy(a).cuda(), None if k is None else torch.from_numpy(k).cuda(), auto_reset=True, out=out)
        assert _lib.lib().snac_last_kernel() == kernel
        oc, rc, dc = orc.step(t, a, k, auto_reset=True, nthreads=16)
        assert helpers.same_bytes(og.cpu().numpy(), cast(oc)), t
        assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(n, -1), st["grid"])
    assert np.array_equal(env.count_brick.cpu().numpy(), st["cb"]) and np.array_equal(env.episode.cpu().numpy(), st["episode"])
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))


@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_large_batches_step_like_the_oracle(dyn, f32):
    """N = 98 304 + 36 (a last wave of 4 envs behind two full ones): 45 ticks with auto-reset -- counter RNG, then explicit actions biased
    towards row moves (the side the extra rows of a span lie on) and builds."""
    env, orc, cast = _pair(dyn, 98304 + 36, 31, f32)
    _walk(env, orc, cast, 45, np.random.default_rng(8))


@pytest.mark.parametrize("n", [4, 8, 12, 16, 20, 100, 1000, 4096 + 12])
def test_small_batches_and_ragged_waves(n):
    """One wave with 4 / 8 / 12 / 16 envs, two waves, a ragged last wave: 60 ticks, explicit actions biased to each side in turn (column
    moves left and right: the columns a span piece may skip)."""
    env, orc, cast = _pair(True, n, 7, False, total_step=25)
    rng = np.random.default_rng(n)
    _walk(env, orc, cast, 30, rng, explicit_from=10, probs=[0.3, 0.3, 0.05, 0.05, 0.075, 0.075, 0.075, 0.075])
    env2, orc2, cast2 = _pair(False, n, 9, True, total_step=25)
    _walk(env2, orc2, cast2, 30, rng, explicit_from=5, probs=[0.05, 0.05, 0.3, 0.3, 0.075, 0.075, 0.075, 0.075])


def test_steps_without_observations_and_scalar_inputs():
    """obs=None steps (the kernel returns before the window), the scalar forms of action / step size, and a step that writes rows again:
    the state carried through equals the oracle's."""
    import torch
    from snac_amd import _lib

    n = 2048 + 4
    env, orc, cast = _pair(True, n, 3, False, total_step=30)
    rng = np.random.default_rng(2)
    for t in range(40):
        a = rng.integers(0, 8, size=n).astype(np.int8)
        k = rng.integers(1, 4, size=n).astype(np.int8)
        if t % 3 == 0:
            rg, dg = env.step(torch.from_numpy(a).cuda(), torch.from_numpy(k).cuda(), auto_reset=True, want_obs=False)[1:]
            assert _lib.lib().snac_last_kernel() == b"k_step3dq"
            oc, rc, dc = orc.step(t, a, k, auto_reset=True, nthreads=16)
        else:
            og, rg, dg = env.step(torch.from_numpy(a).cuda(), torch.from_numpy(k).cuda(), auto_reset=True)
            oc, rc, dc = orc.step(t, a, k, auto_reset=True, nthreads=16)
            assert helpers.same_bytes(og.cpu().numpy(), oc), t
        assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(n, -1), st["grid"])


def test_unaligned_rows_and_odd_batches_stay_on_the_other_kernels():
    import torch
    from snac_amd import BatchedDMPEnv, _lib

    e = BatchedDMPEnv(3, True, 1002, seed=1)
    e.reset()
    e.step(auto_reset=True)
    assert _lib.lib().snac_last_kernel() != b"k_step3dq"             # N % 4 != 0
    e = BatchedDMPEnv(3, True, 1000, seed=1)
    e.reset()
    raw = torch.empty(1000 * 51 + 1, dtype=torch.float64, device="cuda")
    out = (raw[1:].view(1000, 51), torch.empty(1000, dtype=torch.float32, device="cuda"), torch.empty(1000, dtype=torch.uint8, device="cuda"))
    e.step(auto_reset=True, out=out)
    assert _lib.lib().snac_last_kernel() != b"k_step3dq"             # rows that do not start on 16 bytes
    e.step(auto_reset=True)
    assert _lib.lib().snac_last_kernel() == b"k_step3dq"


@pytest.mark.parametrize("n", [376828, 376832, 557052, 557056])
def test_the_forms_of_loads_and_stores_at_their_thresholds(n):
    """Round 6: k_step3dq picks a form by batch size -- below 376 832 envs (SNAC_STEP3D_NTLOAD_MIN) the "resident" form: plain span loads
    (the state stays in the Infinity Cache) + non-temporal row stores; up to 557 056 (SNAC_STEP3D_HUGE_MIN) non-temporal loads + plain rows;
    above, both non-temporal -- the same rows on either side of both thresholds (and the largest batches this file steps against the oracle)."""
    env, orc, cast = _pair(True, n, seed=21, f32=False, total_step=25)
    _walk(env, orc, cast, 6, np.random.default_rng(3), explicit_from=3)
