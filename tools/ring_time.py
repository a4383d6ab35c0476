"""A 600-step collection into a replay ring at N = 65 536 (snac_rollout_rec with the record outputs), ring in hipMalloc memory and in
trajectory memory, both layouts.    gpurun -- python tools/ring_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv, ReplayRing  # noqa: E402


def main():
    n, cap = 65536, 600
    for layout in ("ticks", "tiled"):
        for memory in ("malloc", "vmm"):
            env = BatchedDMPEnv(2, True, n, seed=1)
            env.reset()
            ring = ReplayRing(env, cap, layout=layout, memory=memory)
            for _ in range(8):
                ring.collect(cap)
            ev = []
            for _ in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ring.collect(cap)
                b.record()
                ev.append((a, b))
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in ev)
            print("ring %-5s %-6s: collect(%d) at N=%d  min %.3f  median %.3f ms" % (layout, memory, cap, n, t[0], t[len(t) // 2]), flush=True)
            del ring, env
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
