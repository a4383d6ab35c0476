"""snac_reset timed on the device: all envs reset back to back, a masked reset of a quarter of them, and snac_iou / snac_observe of a batch mid-episode.

    gpurun -- python tools/reset_time.py [kind] [N] [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv, _lib  # noqa: E402


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    env = BatchedDMPEnv(kind, True, n, seed=1)
    env.reset()
    mask = (torch.arange(n, device="cuda") % 4 == 0).to(torch.uint8)
    for name, kw in (("all envs", {}), ("a quarter (mask)", {"mask": mask}), ("all envs, no rows", {"want_obs": False})):
        for _ in range(10):
            env.reset(**kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            env.reset(**kw)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        rowb = env.obs_dim * 8
        print("%dD reset N=%d %-18s %.1f us  (%s; rows %d B per env: %.2f TB/s written)" % (kind, n, name, us, _lib.lib().snac_last_kernel().decode(), rowb, n * rowb / us / 1e6), flush=True)
    env.rollout(40, obs=None)                                        # envs mid-episode: boards with bricks
    for _ in range(10):
        env.iou()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        env.iou()
    b.record()
    torch.cuda.synchronize()
    print("%dD iou   N=%d %.1f us  (%s)" % (kind, n, a.elapsed_time(b) / reps * 1e3, _lib.lib().snac_last_kernel().decode()), flush=True)
    for _ in range(10):
        env.observe()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        env.observe()
    b.record()
    torch.cuda.synchronize()
    print("%dD observe N=%d %.1f us  (%s)" % (kind, n, a.elapsed_time(b) / reps * 1e3, _lib.lib().snac_last_kernel().decode()), flush=True)


if __name__ == "__main__":
    main()
