"""snac_replay_gather alone (raw C ABI, device time over back-to-back launches): float32 (s, s', plan) minibatches of `batch` random
transitions out of a float64 ring of 64 ticks x 65 536 envs.  SNAC_GATHER16=0 keeps every sample on the row-per-store kernel.

    gpurun -- python tools/gather_time.py [batch] [reps]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv, ReplayRing, _lib  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    for kind in (2, 3):
        n, cap = 65536, 64
        env = BatchedDMPEnv(kind, True, n, seed=1)
        env.reset()
        ring = ReplayRing(env, cap)
        ring.collect(cap)
        slot = torch.randint(1, cap, (batch,), device=env.device).to(torch.int32)
        ei = torch.randint(0, n, (batch,), device=env.device).to(torch.int32)
        s = torch.empty((batch, 51), dtype=torch.float32, device=env.device)
        sn = torch.empty_like(s)
        plan = torch.empty((batch, 400), dtype=torch.float32, device=env.device)
        vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

        def call():
            _lib.check(env._lib.snac_replay_gather(C.byref(env._desc), C.byref(env._state), cap, vp(ring.obs), vp(ring.first), vp(ring.plan_idx),
                                                   vp(slot), vp(ei), batch, vp(s), vp(sn), vp(plan), env._stream()))

        for _ in range(10):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            call()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        per = 2 * 408 + 2 * 204 + 1600 + 8 + 3
        print("%dD gather16=%s: %d samples %.4f ms  %.3e samples/s  %.0f GB/s of %d B per sample = %.2f of 8 TB/s" % (
            kind, os.environ.get("SNAC_GATHER16", "1"), batch, ms, batch / ms * 1e3, per * batch / ms / 1e6, per, per * batch / ms / 1e6 / 8000))


if __name__ == "__main__":
    main()
