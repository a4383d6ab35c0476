"""GPU: k_step3ds and k_step3d behind k_step3dq.  Since the end of round 5 the canonical 3D snac_step on identity rows takes k_step3dq
(tests/test_gpu_step3dq.py) at every batch size; the two kernels it replaced stay as what SNAC_STEP3D_QUARTER=0 falls back to, and their
tests run in ONE child process with that switch (and the span kernel's threshold lowered to 4 envs).  Below: what this file said when
k_step3ds was the default for large batches.

k_step3ds (round 5) -- the canonical 3D snac_step of large batches with cooperative span loads (26 neighbouring lanes read the ten
rows a tick can need; pieces outside the rows / columns the tick can touch are not fetched; half a wave at a time through the staging tile).
By default it takes batches of 81 920 envs and more (SNAC_STEP3D_SPAN_MIN); here: at its own batch sizes against the CPU oracle and against
k_step3d's rows, and -- in ONE child process with the threshold lowered to 4 envs -- under every step test of the suite (ragged tiles,
explicit inputs that walk to every edge, steps without observations, properties)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

INNER = os.environ.get("SNAC_TEST_STEP3DS_INNER") == "1"
inner = pytest.mark.skipif(not INNER, reason="runs in the child process of test_every_step_test_of_the_suite_on_span_loads (SNAC_STEP3D_QUARTER=0)")


@inner
@pytest.mark.parametrize("f32", [False, True], ids=["f64", "f32"])
@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_inner_large_batches_step_on_span_loads_like_the_oracle(dyn, f32):
    """N = 98 304 + 36 (a ragged last tile of 36 envs): 45 ticks with auto-reset -- counter RNG, then explicit actions biased towards
    row moves (the side the extra rows of a span lie on) and builds -- rows, rewards, done flags and the end state against the oracle."""
    import torch
    from snac_amd import BatchedDMPEnv, _lib

    n = 98304 + 36
    table = helpers.plan_table(3, dyn, "dense_train" if dyn else "p1")
    env = BatchedDMPEnv(3, dyn, n, plans=table.reshape(len(table), 26, 26), seed=31, total_step=40, obs_dtype=torch.float32 if f32 else torch.float64)
    orc = helpers.oracle().OracleBatch(3, dyn, n, table, seed=31)
    orc.set_total_step(40)
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    assert helpers.same_bytes(env.reset().cpu().numpy(), cast(orc.reset()))
    out = (torch.empty((n, 51), dtype=env.obs_dtype, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda"))
    rng = np.random.default_rng(8)
    for t in range(45):
        a = k = None
        if t >= 15:
            a = rng.choice(8, size=n, p=[0.08, 0.08, 0.27, 0.27, 0.075, 0.075, 0.075, 0.075]).astype(np.int8)
            k = rng.integers(1, 4, size=n).astype(np.int8)
        og, rg, dg = env.step(None if a is None else torch.from_numpy(a).cuda(), None if k is None else torch.from_numpy(k).cuda(), auto_reset=True, out=out)
        assert _lib.lib().snac_last_kernel() == b"k_step3ds"
        oc, rc, dc = orc.step(t, a, k, auto_reset=True, nthreads=16)
        assert helpers.same_bytes(og.cpu().numpy(), cast(oc)), t
        assert helpers.same_bytes(rg.cpu().numpy(), rc) and np.array_equal(dg.cpu().numpy().view(np.uint8), dc), t
    st = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(n, -1), st["grid"])
    assert np.array_equal(env.count_brick.cpu().numpy(), st["cb"]) and np.array_equal(env.episode.cpu().numpy(), st["episode"])
    s, e = orc.stats(), env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))


def test_every_step_test_of_the_suite_on_span_loads():
    """One child process, SNAC_STEP3D_QUARTER=0 and SNAC_STEP3D_SPAN_MIN=4: the step tests of tests/test_gpu_step_tile.py and
    tests/test_gpu_property.py (their 3D cases on canonical rows: the layout variants and the 2D tests never reach these kernels) and the
    large batches above, with every 3D snac_step on identity rows taking k_step3ds."""
    if INNER:
        pytest.skip("the child itself")
    env = dict(os.environ, SNAC_STEP3D_SPAN_MIN="4", SNAC_STEP3D_QUARTER="0", SNAC_TEST_STEP3DS_INNER="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_step_tile.py", "tests/test_gpu_property.py", os.path.abspath(__file__), "-x", "-q", "-m", "gpu",
                          "-k", "(3d or property or dim or inner) and not layout_variants and not 2d_", "-p", "no:cacheprovider"],
                         cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert " passed" in out.stdout
