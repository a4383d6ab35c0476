"""GPU: the C ABI is re-entrant and stream-ordered (caller-owned buffers, the caller's stream, no hidden
syncs, no global mutable state but a thread-local error string).  Four host threads, each with its own torch stream and its own
batch (different env kinds and sizes), step and roll out concurrently; every thread's outputs must equal what the same calls
give when they run alone, and an error raised in one thread must not leak into another thread's snac_last_error()."""
import threading

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

JOBS = [(1, True, 3000, 11), (2, True, 5000, 12), (3, True, 700, 13), (2, False, 64, 14)]


def _work(kind, dyn, n, seed, stream=None):
    import torch
    from snac_amd import BatchedDMPEnv

    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        env = BatchedDMPEnv(kind, dyn, n, seed=seed)
        outs = [env.reset().clone()]
        for rep in range(6):
            o, r, d = env.rollout(40)
            outs += [o.clone(), r.clone(), d.clone()]
            for _ in range(10):
                o, r, d = env.step(auto_reset=True)
                outs += [o.clone(), r.clone(), d.clone()]
        outs.append(env.iou().clone())
        torch.cuda.current_stream().synchronize()
    return [t.cpu().numpy() for t in outs]


def test_concurrent_threads_on_their_own_streams():
    import torch

    alone = [_work(*job) for job in JOBS]
    got = [None] * len(JOBS)
    errs = []

    def run(i):
        try:
            got[i] = _work(*JOBS[i], stream=torch.cuda.Stream())
        except Exception as e:                                   # noqa: BLE001
            errs.append((i, repr(e)))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(JOBS))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for a, g in zip(alone, got):
        assert len(a) == len(g) and all(x.tobytes() == y.tobytes() for x, y in zip(a, g))


def test_error_strings_are_per_thread():
    import ctypes as C

    from snac_amd import BatchedDMPEnv, _lib

    L = _lib.lib()
    env = BatchedDMPEnv(2, True, 8, seed=1)
    env.reset()
    seen = {}

    def bad():
        desc = _lib.EnvDesc.from_buffer_copy(env._desc)
        desc.num_envs = -5
        rc = L.snac_reset(C.byref(desc), C.byref(env._state), None, None, None, env._stream())
        seen["bad"] = (rc, L.snac_last_error())

    t = threading.Thread(target=bad)
    t.start()
    t.join()
    assert seen["bad"][0] < 0 and b"num_envs must be positive" in seen["bad"][1]
    assert b"num_envs must be positive" not in (L.snac_last_error() or b"")   # this thread never failed with that message
    env.step(auto_reset=True)                                    # and the env of this thread is still usable
