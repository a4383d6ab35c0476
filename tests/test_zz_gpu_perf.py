"""GPU, LAST in the suite (the file name sorts behind every parity file): every host- or event-timed bound of the suite lives here.
Each bound REPORTS -- a figure outside it is a warning in the pytest summary and a line in gpurun_out/perf_report.json, not a failure:
a noisy box must not be able to turn the parity suite red (`pytest -x` stops at the first failure, and round 5's timing lines sat in
front of the oracle comparisons).  SNAC_PERF_STRICT=1 makes the bounds binding (the builder's own runs)."""
import json
import os
import time
import warnings

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

STRICT = os.environ.get("SNAC_PERF_STRICT", "0") == "1"
REPORT = []


def bound(name, ok, detail):
    REPORT.append({"name": name, "ok": bool(ok), "detail": detail})
    if ok:
        return
    if STRICT:
        pytest.fail("perf bound %s: %r" % (name, detail))
    warnings.warn("perf bound missed (reported, not failed): %s: %r" % (name, detail))


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    out = os.path.join(helpers.ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "perf_report.json"), "w") as fh:
            json.dump({"strict": STRICT, "bounds": REPORT, "notes": helpers.perf_notes()}, fh, indent=1, default=str)
    except OSError:
        pass


def test_drop_in_class_rate_mailbox_against_launch_path():
    """The class on its resident wavefront against SNAC_MAILBOX=0 (one launch + one wait per step): same trajectory (asserted in
    tests/test_gpu_mailbox.py), here the rates.  The reference class's 110 k steps/s is a figure from the build container's CPU, not
    from this box (README): reported beside the rates, no bound hangs on it."""
    from snac_amd.envs import deep_mobile_printing_2d1r_dynamic

    def run(steps):
        e = deep_mobile_printing_2d1r_dynamic("data_2d_dynamic_dense_envplan_500_train.pkl")
        np.random.seed(1)
        e.reset()
        acts = np.random.RandomState(0).randint(0, 5, steps)
        t0 = time.perf_counter()
        for i in range(steps):
            if e.step(int(acts[i]))[2]:
                e.reset()
        dt = time.perf_counter() - t0
        e.close()
        return steps / dt

    rate1 = run(6000)
    os.environ["SNAC_MAILBOX"] = "0"
    try:
        rate0 = run(6000)
    finally:
        del os.environ["SNAC_MAILBOX"]
    bound("facade_mailbox_vs_launch", rate1 > 1.5 * rate0, {"mailbox_steps_per_s": rate1, "launch_steps_per_s": rate0,
                                                           "reference_class_other_cpu": 110300.0})


def test_headline_pass_on_measured_blocks():
    """Four headline-sized blocks in turn: each takes the headline pass at the two-slice level -- within 4 % of the fastest of the four,
    at most 2.45 ms -- and says so in its own description (no slow window, the block within 5 % of the box's fast level)."""
    import torch
    from snac_amd import BatchedDMPEnv, trajmem

    n, T = 65536, 600
    env = BatchedDMPEnv(2, True, n, seed=1)
    env.reset()
    times, infos = [], []
    for _ in range(4):
        buf = trajmem.traj_empty((T, n, env.obs_dim), torch.float64, "cuda")
        d = trajmem.describe(buf)
        for _ in range(12):
            env.rollout(T, obs="all", out=buf, want_reward=False, want_done=False)
        ev = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            env.rollout(T, obs="all", out=buf, want_reward=False, want_done=False)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        times.append(sorted(a.elapsed_time(b) for a, b in ev)[3])
        infos.append(d)
        del buf
    for i, d in enumerate(infos):
        bound("traj_block_%d_described_fast" % i,
              d["layout"] == "measured: two slices in turn" and d["windows_slow"] == 0
              and d["us_per_gib"]["block"] <= 1.05 * d["us_per_gib"]["fast"]
              and max(d["us_per_gib"]["windows"]) <= 1.08 * d["us_per_gib"]["fast"] + 0.5, d)
    bound("traj_blocks_within_4_percent", max(times) <= 1.04 * min(times), times)
    bound("traj_blocks_headline_ms", max(times) <= 2.45, times)


def test_figures_noted_by_the_parity_files():
    """Build times etc. measured in passing by earlier files of this run (helpers.perf_note)."""
    notes = helpers.perf_notes()
    b = notes.get("trajmem_32_blocks_build_s")
    if b is not None:
        bound("trajmem_block_build_median_s", b["median"] < 1.5, b)
