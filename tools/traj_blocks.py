"""snac_traj_alloc under the headline workload: K trajectory blocks allocated, rolled into and freed in turn (or held, `hold`),
each with the allocator's own description (snac_traj_describe) beside the real rollout's time.  SNAC_TRAJ_DEBUG=1 adds the probe
trace on stderr.

    gpurun -- python tools/traj_blocks.py [blocks] [reps] [hold]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv, trajmem  # noqa: E402


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    hold = len(sys.argv) > 3 and sys.argv[3] == "hold"
    n, T = 65536, 600
    env = BatchedDMPEnv(2, True, n, seed=1)
    env.reset()
    rew = torch.empty((T, n), dtype=torch.float32, device="cuda")
    done = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    held = []
    for b in range(blocks):
        t0 = time.perf_counter()
        buf = trajmem.traj_empty((T, n, env.obs_dim), torch.float64, "cuda")
        dt = time.perf_counter() - t0
        for _ in range(14):
            env.rollout(T, obs="all", out=buf, reward_out=rew, done_out=done)
        ev = []
        for _ in range(reps):
            a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            env.rollout(T, obs="all", out=buf, reward_out=rew, done_out=done)
            c.record()
            ev.append((a, c))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(c) for a, c in ev)
        d = trajmem.describe(buf)
        print("BLOCK %d: built in %.2f s, rollout min %.3f median %.3f ms  %s" % (b, dt, ms[0], ms[len(ms) // 2], json.dumps(d)), flush=True)
        if hold:
            held.append(buf)
        del buf
    print("reserved address space: %.1f GiB" % (trajmem.reserved_bytes() / 2 ** 30))


if __name__ == "__main__":
    main()
