"""What bounds snac_replay_gather: the same launch with random samples, with samples in ring order (sequential reads), without the plan
output, and with the s / s' outputs only -- device time, back to back.

    gpurun -- python tools/gather_probe.py [batch] [reps]
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
    n, cap = 65536, 64
    env = BatchedDMPEnv(2, True, n, seed=1)
    env.reset()
    ring = ReplayRing(env, cap)
    ring.collect(cap)
    dev = env.device
    s = torch.empty((batch, 51), dtype=torch.float32, device=dev)
    sn = torch.empty_like(s)
    plan = torch.empty((batch, 400), dtype=torch.float32, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    arms = {
        "random samples": (torch.randint(1, cap, (batch,), device=dev).to(torch.int32), torch.randint(0, n, (batch,), device=dev).to(torch.int32), plan),
        "ring order (tick 5, env i)": (torch.full((batch,), 5, device=dev, dtype=torch.int32), (torch.arange(batch, device=dev) % n).to(torch.int32), plan),
        "random ticks, env i": (torch.randint(1, cap, (batch,), device=dev).to(torch.int32), (torch.arange(batch, device=dev) % n).to(torch.int32), plan),
        "random samples, no plan output": (torch.randint(1, cap, (batch,), device=dev).to(torch.int32), torch.randint(0, n, (batch,), device=dev).to(torch.int32), None),
        "ring order, no plan output": (torch.full((batch,), 5, device=dev, dtype=torch.int32), (torch.arange(batch, device=dev) % n).to(torch.int32), None),
    }
    # the arms above repeat ONE set of samples: its 70 MB of rows stay in the 256 MB Infinity Cache from launch to launch.  A trainer draws new
    # samples every time out of a ring of 1.7 GB:
    sets = [(torch.randint(1, cap, (batch,), device=dev).to(torch.int32), torch.randint(0, n, (batch,), device=dev).to(torch.int32)) for _ in range(reps + 10)]
    it = iter(sets)

    def fresh():
        sl, e = next(it)
        _lib.check(env._lib.snac_replay_gather(C.byref(env._desc), C.byref(env._state), cap, vp(ring.obs), vp(ring.first), vp(ring.plan_idx),
                                               vp(sl), vp(e), batch, vp(s), vp(sn), vp(plan), env._stream()))
    for _ in range(10):
        fresh()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fresh()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    print("%-34s %d samples %.1f us   %d B per sample: %.2f TB/s" % ("NEW random samples every launch", batch, us, 2835, 2835 * batch / us / 1e6), flush=True)
    for name, (slot, ei, pl) in arms.items():
        def call():
            _lib.check(env._lib.snac_replay_gather(C.byref(env._desc), C.byref(env._state), cap, vp(ring.obs), vp(ring.first), vp(ring.plan_idx),
                                                   vp(slot), vp(ei), batch, vp(s), vp(sn), vp(pl), env._stream()))
        for _ in range(10):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            call()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        per = 2 * 408 + 2 * 204 + (1600 if pl is not None else 0) + 8 + 3
        print("%-34s %d samples %.1f us   %d B per sample: %.2f TB/s" % (name, batch, us, per, per * batch / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
