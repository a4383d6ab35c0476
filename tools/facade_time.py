"""Single-env drop-in rate on the GPU box: the reference-shaped classes at N = 1 (what a DQN / DRQN script calls once per
env-step), 2D dynamic dense, 5000 steps with random actions, reset on done.

  round-1 path   two H2D tensors (action, step size) + launch + header .cpu() + obs .cpu() + reward.item() + done.item()
                 -- reconstructed here on BatchedDMPEnv.step() exactly as snac_amd/envs.py::_do_step did it
  record path    snac_step_scalar (action / step size by value) writing [obs | reward | done | position | counters] into one row,
                 one .cpu()  -- what the classes do now
  the class      deep_mobile_printing_2d1r(data_path).step(a) itself, with its python bookkeeping

The reference class does 110 k steps/s/core in this container.  Prints one line per variant."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from snac_amd import BatchedDMPEnv  # noqa: E402

STEPS = int(os.environ.get("FACADE_STEPS", "5000"))


def legacy():
    env = BatchedDMPEnv(2, True, 1, seed=1)
    env.reset(plan_idx=np.asarray([3], np.int16))
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 5, STEPS)
    t0 = time.perf_counter()
    for i in range(STEPS):
        k = int(np.random.randint(1, 4))
        obs, reward, done = env.step(torch.tensor([int(acts[i])], dtype=torch.int8), torch.tensor([k], dtype=torch.int8))
        h8 = env._hdr.cpu().numpy().view(np.int8).reshape(-1)
        h16 = h8.view(np.int16)
        _ = int(h8[0]), int(h8[1]), int(h16[2]), int(h16[3]), int(h16[4])
        o, r, d = obs.cpu().numpy(), float(reward.item()), bool(done.item())
        if d:
            env.reset(plan_idx=np.asarray([int(np.random.randint(0, 400))], np.int16)).cpu()
    return STEPS / (time.perf_counter() - t0)


def record(pinned=False):
    env = BatchedDMPEnv(2, True, 1, seed=1, obs_tail=("record",))
    row = env._new_obs()
    host = torch.empty(row.shape, dtype=row.dtype, pin_memory=True) if pinned else None
    env.reset_scalar(3, out=row)
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 5, STEPS)
    t0 = time.perf_counter()
    for i in range(STEPS):
        k = int(np.random.randint(1, 4))
        env.step_scalar(int(acts[i]), k, out=row)
        if pinned:
            host.copy_(row, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            r = host.numpy()
        else:
            r = row.cpu().numpy()
        if r[0, 52]:
            env.reset_scalar(int(np.random.randint(0, 400)), out=row)
    return STEPS / (time.perf_counter() - t0)


def zerocopy(spin=False):
    """The kernel writes its row straight into page-locked host memory (the pointer is valid on the device: unified addressing);
    the host only waits for the stream -- no copy command at all."""
    import ctypes as C

    from snac_amd import _lib

    env = BatchedDMPEnv(2, True, 1, seed=1, obs_tail=("record",))
    host = torch.empty((1, env.obs_dim), dtype=torch.float64, pin_memory=True)
    view = host.numpy()
    L = env._lib
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    hp = C.c_void_p(host.data_ptr())
    _lib.check(L.snac_reset_scalar(C.byref(env._desc), C.byref(env._state), 3, hp, sp))
    stream.synchronize()
    env._was_reset = True
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 5, STEPS)
    ev = torch.cuda.Event()
    t0 = time.perf_counter()
    for i in range(STEPS):
        k = int(np.random.randint(1, 4))
        _lib.check(L.snac_step_scalar(C.byref(env._desc), C.byref(env._state), env.t & 0xFFFFFFFF, int(acts[i]), k, 0, hp, None, None, sp))
        env.t += 1
        if spin:
            ev.record(stream)
            while not ev.query():
                pass
        else:
            stream.synchronize()
        if view[0, 52]:
            _lib.check(L.snac_reset_scalar(C.byref(env._desc), C.byref(env._state), int(np.random.randint(0, 400)), hp, sp))
            stream.synchronize()
    return STEPS / (time.perf_counter() - t0)


def the_class():
    from snac_amd.envs import deep_mobile_printing_2d1r_dynamic

    env = deep_mobile_printing_2d1r_dynamic("data_2d_dynamic_dense_envplan_500_train.pkl")
    env.reset()
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 5, STEPS)
    t0 = time.perf_counter()
    for i in range(STEPS):
        s, r, d = env.step(int(acts[i]))
        if d:
            env.reset()
    return STEPS / (time.perf_counter() - t0)


def main():
    for name, fn in (("round-1 path (2 H2D + 4 D2H syncs per step)", legacy), ("record row + .cpu() (1 D2H per step)", record),
                     ("record row + pinned async copy + stream sync", lambda: record(True)),
                     ("row written into pinned host memory + stream sync", zerocopy),
                     ("row written into pinned host memory + event spin", lambda: zerocopy(True)),
                     ("deep_mobile_printing_2d1r.step() (the class)", the_class)):
        fn()                                                     # warm-up pass
        rate = fn()
        print("N=1 2D dynamic dense  %-48s %9.0f steps/s  %7.1f us/step" % (name, rate, 1e6 / rate))


if __name__ == "__main__":
    main()
