"""Where does a 10x slower step() extra come from (the driver's BENCH_r04 had
step_2d_ppo_layout_n65536 at 396 us per tick, the builder's own run 38)?  The extra as bench.py ran it -- output rows at the start of
a trajectory block, 25 untimed + 100 timed launches -- but with an event PER launch, in a few conditions:
  mem     trajectory block (snac_traj_alloc) / torch.empty
  fresh   a block nothing has written before / one that a headline rollout went through first
Prints min / median / max and the slowest launches with their index.  usage: step_outlier.py [kind N layout reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snac_amd import BatchedDMPEnv, _lib, trajmem  # noqa: E402


def run(kind, nn, layout, reps, buf, label):
    dev = torch.device("cuda:0")
    e = BatchedDMPEnv(kind, True, nn, device=dev, seed=1, **({"layout": layout} if layout else {}))
    e.reset()
    nb = nn * e.obs_dim * 8
    out = (buf[:nb].view(torch.float64).view(nn, e.obs_dim), torch.empty(nn, dtype=torch.float32, device=dev), torch.empty(nn, dtype=torch.uint8, device=dev))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    ev[0].record()
    host = []
    for i in range(reps):
        t0 = time.perf_counter()
        e.step(auto_reset=True, out=out)
        host.append((time.perf_counter() - t0) * 1e6)
        ev[i + 1].record()
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps)]
    s = sorted(us)
    worst = sorted(range(reps), key=lambda i: -us[i])[:5]
    print("%-44s %s  min %.1f  med %.1f  mean %.1f  max %.1f us; total %.2f ms (host loop %.2f ms); slowest: %s; host call max %.0f us at %d" % (
        label, _lib.lib().snac_last_kernel().decode(), s[0], s[reps // 2], sum(us) / reps, s[-1], ev[0].elapsed_time(ev[reps]), (h1 - h0) * 1e3,
        ", ".join("#%d %.0f" % (i, us[i]) for i in worst), max(host), host.index(max(host))), flush=True)


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    nn = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    layout = (sys.argv[3] if len(sys.argv) > 3 else "ppo") or None
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 125
    dev = torch.device("cuda:0")
    big = 600 * 65536 * 51 * 8
    for trial in range(3):
        blk = trajmem.traj_empty((big,), torch.uint8, dev)
        run(kind, nn, layout, reps, blk, "trial %d: fresh trajectory block" % trial)
        run(kind, nn, layout, reps, blk, "trial %d: the same block again" % trial)
        h = BatchedDMPEnv(2, True, 65536, device=dev, seed=1)
        h.reset()
        for _ in range(3):
            h.rollout(600, obs="all", out=blk.view(torch.float64).view(600, 65536, 51))
        torch.cuda.synchronize()
        run(kind, nn, layout, reps, blk, "trial %d: after headline rollouts into it" % trial)
        del h
        pl = torch.empty(nn * 460 * 8, dtype=torch.uint8, device=dev)
        run(kind, nn, layout, reps, pl, "trial %d: torch.empty rows" % trial)
        del pl, blk
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
