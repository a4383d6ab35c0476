"""snac_rollout WITHOUT observation rows (obs=None: what a default-policy evaluation of the tree-search scripts, or a policy evaluation that only
wants the episodic sums, runs): device time per launch of T ticks.

    gpurun -- python tools/noobs_time.py [kind] [N] [T] [reps] [last]          last: obs="last" (the final observation only)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv, _lib  # noqa: E402


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    mode = "last" if (len(sys.argv) > 5 and sys.argv[5] == "last") else None
    env = BatchedDMPEnv(kind, True, n, seed=1)
    env.reset()
    T = T or env.total_step
    for _ in range(3):
        env.rollout(T, obs=mode)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        env.rollout(T, obs=mode)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print("%dD rollout obs=%s N=%d T=%d: %.3f ms  %.3e env-steps/s  (%s)" % (kind, mode, n, T, ms, n * T / ms * 1e3, _lib.lib().snac_last_kernel().decode()), flush=True)


if __name__ == "__main__":
    main()
