"""snac_traj_alloc under allocator churn: blocks of 1 .. 16 GiB allocated, held (up to three at a time) and freed in random order with
hipMalloc tensors of 4 .. 48 GiB coming and going in between -- the states of the driver's free lists a long-lived process sees.
Every block: its description (snac_traj_describe), and the whole of it written by the store pattern of the headline rollout when it
is headline-sized.  Prints one line per block and a summary; SNAC_TRAJ_DEBUG=1 adds the probe traces.

    gpurun -- python tools/traj_stress.py [blocks] [seed]
"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv, trajmem  # noqa: E402


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    n, T = 65536, 600
    env = BatchedDMPEnv(2, True, n, seed=1)
    env.reset()
    held, ballast = [], []
    stats = {"measured": 0, "small_one_run": 0, "fallback": 0, "slow_windows": 0, "rebuilds": 0, "worst_block_us": 0.0, "worst_build_s": 0.0, "rollout_ms": []}
    for b in range(blocks):
        # churn: ballast tensors come and go
        while ballast and rnd.random() < 0.5:
            ballast.pop(rnd.randrange(len(ballast)))
        torch.cuda.empty_cache()
        if rnd.random() < 0.7:
            free = torch.cuda.mem_get_info()[0]
            gib = rnd.choice([4, 8, 16, 24, 32, 48])
            if free > (gib + 64) << 30:
                ballast.append(torch.empty(gib << 30, dtype=torch.uint8, device="cuda"))
        while held and (len(held) >= 3 or rnd.random() < 0.4):
            held.pop(rnd.randrange(len(held)))
        headline = rnd.random() < 0.6
        shape = (T, n, 51) if headline else (rnd.choice([40, 80, 150, 300]), n, 51)
        t0 = time.perf_counter()
        buf = trajmem.traj_empty(shape, torch.float64, "cuda")
        dt = time.perf_counter() - t0
        d = trajmem.describe(buf)
        ms = None
        if headline:
            for _ in range(10):
                env.rollout(T, obs="all", out=buf, want_reward=False, want_done=False)
            ev = []
            for _ in range(7):
                a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                env.rollout(T, obs="all", out=buf, want_reward=False, want_done=False)
                c.record()
                ev.append((a, c))
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(c) for a, c in ev)[3]
            stats["rollout_ms"].append(round(ms, 3))
        meas = d["layout"].startswith("measured")
        stats["measured" if meas else ("small_one_run" if d["layout"] == "one run" else "fallback")] += 1   # below 1 GiB: one run by design
        stats["slow_windows"] += d["windows_slow"]
        stats["rebuilds"] += d["rebuilds"]
        stats["worst_block_us"] = max(stats["worst_block_us"], d["us_per_gib"]["block"])
        stats["worst_build_s"] = max(stats["worst_build_s"], dt)
        print("BLOCK %d: %.1f GiB, %d held, %d ballast (%.0f GiB free before), built in %.2f s, %s, rebuilds %d, pool %d groups, block %.0f us/GiB "
              "(fast %.0f slow %.0f, windows max %.0f, slow windows %d)%s" % (
                  b, d["bytes"] / 2 ** 30, len(held), len(ballast), torch.cuda.mem_get_info()[0] / 2 ** 30 + d["bytes"] / 2 ** 30, dt, d["layout"],
                  d["rebuilds"], d["pool_groups"], d["us_per_gib"]["block"], d["us_per_gib"]["fast"], d["us_per_gib"]["slow"],
                  d["us_per_gib"]["window_max"], d["windows_slow"], "" if ms is None else ", rollout median %.3f ms" % ms), flush=True)
        held.append(buf)
        del buf
    print("SUMMARY " + json.dumps(stats))
    print("reserved address space: %.1f GiB" % (trajmem.reserved_bytes() / 2 ** 30))


if __name__ == "__main__":
    main()
