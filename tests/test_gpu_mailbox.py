"""GPU: the resident single-env stepper (snac_mailbox_*, snac_amd/csrc/k_mailbox.hip) -- what the drop-in classes step through.
Every facade test of the suite runs on it already (it is their default path); here: the raw protocol against the launch path and the
oracle, the wave's exits (idle timeout, close, another entry point touching the state).  Rates: tests/test_zz_gpu_perf.py."""
import os
import time

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _env(dim, dyn, **kw):
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(dim, dyn, {1: "sin_train", 2: "dense_train", 3: "dense_train"}[dim] if dyn else "p0")
    full = table.reshape(len(table), 30) if dim == 1 else table.reshape(len(table), 26, 26)
    return BatchedDMPEnv(dim, dyn, 1, plans=full, obs_tail=("record",), seed=3, **kw), table


@pytest.mark.parametrize("dim,dyn", [(1, False), (1, True), (2, False), (2, True), (3, False), (3, True)])
def test_mailbox_steps_equal_the_launch_path_and_the_oracle(dim, dyn):
    """2000 steps with resets on done: row by row the mailbox wave, snac_step_scalar on a twin and the CPU oracle agree -- observation,
    reward, done, position and counters (the record tail) -- and so do the records in HBM afterwards (the wave writes them through)."""
    import torch

    a, table = _env(dim, dyn)
    b, _ = _env(dim, dyn)
    orc = helpers.oracle().OracleEnv(dim, dyn)
    row = a.mailbox_open(idle_us=300)
    host = b.new_host_obs()
    rng = np.random.default_rng(5)
    A = a.num_actions
    mix = np.full(A, 1.0 / A) if dim != 3 else np.array([0.2] * 4 + [0.05] * 4)
    nobs = a.obs_dim - 8
    pidx = 0
    for ep in range(4):
        pidx = int(rng.integers(0, len(table))) if dyn else 0
        a.reset_scalar(pidx, out=row)
        b.reset_scalar(pidx, out=host)
        a.sync(), b.sync()
        o0 = orc.reset(table[pidx])
        assert row.numpy().tobytes() == host.numpy().tobytes()
        assert row.numpy()[0, :nobs].tobytes() == np.asarray(o0, np.float64).reshape(-1)[:nobs].tobytes()
        for t in range(500):
            act, k = int(rng.choice(A, p=mix)), int(rng.integers(1, 4))
            a.mailbox_step(act, k)
            b.step_scalar_wait(act, k, host)
            oo, r, d = orc.step(act, k)
            ra, rb = row.numpy().copy(), host.numpy()
            assert ra.tobytes() == rb.tobytes(), (ep, t)
            assert ra[0, :nobs].tobytes() == np.asarray(oo, np.float64).reshape(-1)[:nobs].tobytes(), (ep, t)
            assert ra[0, nobs] == r and bool(ra[0, nobs + 1]) == bool(d), (ep, t)
            if d:
                break
        a.sync()                                                     # (the write-through trails the acknowledgement: any entry point settles it)
        assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid) and torch.equal(a._stats, b._stats)
        assert a.iou().cpu().numpy().tobytes() == b.iou().cpu().numpy().tobytes()
    st = a.mailbox_stats()
    assert st["steps_served"] >= 4 and st["launches"] >= 1
    a.mailbox_close()
    assert a.mailbox_stats() is None


def test_wave_leaves_when_idle_and_comes_back():
    """The exit condition every wave reaches: idle_us without a command.  A step after that arms a new wave (one more launch) and
    continues the same trajectory; a device-wide synchronisation while a wave is resident returns within the idle time."""
    import torch

    a, table = _env(2, True)
    b, _ = _env(2, True)
    row = a.mailbox_open(idle_us=2000)
    host = b.new_host_obs()
    a.reset_scalar(7, out=row), b.reset_scalar(7, out=host)
    a.sync(), b.sync()
    for t in range(50):
        a.mailbox_step(t % 5, 1 + t % 3), b.step_scalar_wait(t % 5, 1 + t % 3, host)
    assert a.mailbox_stats()["alive"] and a.mailbox_stats()["launches"] == 1
    t0 = time.perf_counter()
    torch.cuda.synchronize()                                   # waits for the resident wave: at most its idle time
    waited = time.perf_counter() - t0
    assert waited < 0.5, waited
    time.sleep(0.05)
    assert not a.mailbox_stats()["alive"]                        # gone by itself
    for t in range(50):
        a.mailbox_step(t % 5, 1 + t % 3), b.step_scalar_wait(t % 5, 1 + t % 3, host)
        assert row.numpy().tobytes() == host.numpy().tobytes()
    assert a.mailbox_stats()["launches"] == 2 and a.mailbox_stats()["steps_served"] == 100
    # another entry point changes the state under the resident wave: it reloads the records before its next step
    a.reset_scalar(11, out=row), b.reset_scalar(11, out=host)
    a.sync(), b.sync()
    for t in range(30):
        a.mailbox_step(4 if t % 2 else 1, 1), b.step_scalar_wait(4 if t % 2 else 1, 1, host)
        assert row.numpy().tobytes() == host.numpy().tobytes()
    a.set_plan_row(11, table[3].reshape(26, 26)), b.set_plan_row(11, table[3].reshape(26, 26))
    for t in range(30):
        a.mailbox_step(4 if t % 2 else 2, 1), b.step_scalar_wait(4 if t % 2 else 2, 1, host)
        assert row.numpy().tobytes() == host.numpy().tobytes()
    a.sync()
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid)
    a.mailbox_close()


def test_the_drop_in_class_steps_through_the_mailbox():
    """Env/2D/DMP_Env_2D_dynamic_usedata_plan.py driven like script/DQN/2d/DQN_2d_dynamic.py:214 drives it: the class is on the
    mailbox by default, SNAC_MAILBOX=0 gives the launch path, both produce the same trajectory from the same np.random seed.  (How fast
    either path is belongs to tests/test_zz_gpu_perf.py, which runs last and reports: nothing in this file can fail on a clock.)"""
    from snac_amd.envs import deep_mobile_printing_2d1r_dynamic

    def run(steps):
        e = deep_mobile_printing_2d1r_dynamic("data_2d_dynamic_dense_envplan_500_train.pkl")
        np.random.seed(1)
        e.reset()
        acts = np.random.RandomState(0).randint(0, 5, steps)
        out = []
        t0 = time.perf_counter()
        for i in range(steps):
            o, r, d = e.step(int(acts[i]))
            out.append((o[0].tobytes(), r, d))
            if d:
                e.reset()
        dt = time.perf_counter() - t0
        return e, out, steps / dt

    e1, t1, rate1 = run(6000)
    assert e1._mbox and e1._env.mailbox_stats()["steps_served"] >= 5990
    os.environ["SNAC_MAILBOX"] = "0"
    try:
        e0, t0, rate0 = run(6000)
    finally:
        del os.environ["SNAC_MAILBOX"]
    assert not e0._mbox and t0 == t1
    e1.close()
    assert not e1._mbox
    o, r, d = e1.step(1)                                         # usable after close(): the launch path
    assert o[0].shape == (1, 51)


@pytest.mark.parametrize("dim,dyn,n", [(1, True, 3), (2, True, 64), (2, False, 9), (3, True, 24), (3, False, 64),
                                       (2, True, 65), (1, False, 128), (3, True, 130), (2, True, 256), (3, False, 256)])
def test_batches_of_up_to_256_envs_step_through_resident_waves(dim, dyn, n):
    """snac_mailbox_step_n: env 64 w + e on lane e of resident wave w (up to four waves, each a launch of its own polling the same
    doorbell; round 5: one wave, 64 envs).  300 vector steps with random actions / step sizes and a masked reset
    every 50 steps (the launch path, under the resident wave): rows, rewards and done flags equal those of snac_step on a twin batch
    and of the oracle; the records in HBM and the episodic sums agree afterwards."""
    import torch
    from snac_amd import BatchedDMPEnv

    table = helpers.plan_table(dim, dyn, {1: "sin_train", 2: "dense_train", 3: "dense_train"}[dim] if dyn else "p0")
    full = table.reshape(len(table), 30) if dim == 1 else table.reshape(len(table), 26, 26)
    a = BatchedDMPEnv(dim, dyn, n, plans=full, seed=11)
    b = BatchedDMPEnv(dim, dyn, n, plans=full, seed=11)
    orc = helpers.oracle().OracleBatch(dim, dyn, n, table, seed=11)
    rows = a.mailbox_open(idle_us=500).numpy()
    rew, don = a.mailbox_outputs()
    rng = np.random.default_rng(2)
    pidx = rng.integers(0, len(table), size=n).astype(np.int32)
    o0 = orc.reset(plan_idx=pidx)
    assert a.reset(plan_idx=pidx).cpu().numpy().tobytes() == o0.tobytes() and b.reset(plan_idx=pidx).cpu().numpy().tobytes() == o0.tobytes()
    A = a.num_actions
    for t in range(300):
        if t and t % 50 == 0:                                    # some envs start over, through another entry point
            mask = (rng.random(n) < 0.5).astype(np.uint8)
            pidx = rng.integers(0, len(table), size=n).astype(np.int32)
            oa, ob = a.reset(mask, pidx), b.reset(mask, pidx)
            assert torch.equal(oa, ob) and oa.cpu().numpy().tobytes() == orc.reset(mask, pidx).tobytes()
        act = rng.integers(0, A, size=n).astype(np.int8)
        k = rng.integers(1, 4, size=n).astype(np.int8)
        a.mailbox_step_n(act, k)
        ob, rb, db = b.step(torch.from_numpy(act), torch.from_numpy(k))
        oc, rc, dc = orc.step(t, act, k)
        assert rows.tobytes() == ob.cpu().numpy().tobytes() == oc.tobytes(), t
        assert rew.tobytes() == rb.cpu().numpy().tobytes() == rc.tobytes() and np.array_equal(don, dc), t
    a.sync()
    assert torch.equal(a._hdr, b._hdr) and torch.equal(a._grid, b._grid) and torch.equal(a._stats, b._stats) and torch.equal(a._episode, b._episode)
    assert a.mailbox_stats()["steps_served"] == 300
    a.mailbox_close()


def test_a_script_that_never_closes_its_env_exits_cleanly():
    """A reference-style script steps its env and simply ends (no close()): the interpreter's exit tells the resident wave to leave
    (weakref.finalize -> snac_mailbox_destroy) and the process ends at once, with status 0; killed outright (SIGKILL, nothing runs at
    exit) the wave is gone with the process's queues -- a second process can use the GPU straight away."""
    import signal
    import subprocess
    import sys

    code = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np
from snac_amd.envs import deep_mobile_printing_2d1r_dynamic
e = deep_mobile_printing_2d1r_dynamic("data_2d_dynamic_dense_envplan_500_train.pkl")
np.random.seed(1); e.reset()
for i in range(500):
    if e.step(i %% 5)[2]: e.reset()
assert e._mbox and e._env.mailbox_stats()["alive"]
print("STEPPED", flush=True)
if len(sys.argv) > 1:
    time.sleep(60)
''' % helpers.ROOT
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "STEPPED" in out.stdout, out.stderr[-1500:]
    assert time.perf_counter() - t0 < 60
    p = subprocess.Popen([sys.executable, "-c", code, "hang"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert "STEPPED" in p.stdout.readline()
    p.send_signal(signal.SIGKILL)                                # the exact child started here
    p.wait(timeout=30)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "STEPPED" in out.stdout, out.stderr[-1500:]
