"""BatchedDMPEnv.evaluate (default-policy evaluation of tree leaves: script/MCTS/utils/mcts.py:100-110) timed on the host, per call: the leaves
forked, rolled out without observation rows, their discounted sums by snac_discounted_return -- and, for scale, the sums as round 5 did them
(a python loop of H steps of torch operations on the same reward / done arrays).

    gpurun -- python tools/eval_time.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv  # noqa: E402


def old_sums(env, rows, H, gamma):
    terminal = env.need_reset[rows]
    leaves = env.fork(rows)
    leaves.t = 0
    _, reward, done = leaves.rollout(H, obs=None)
    est = torch.zeros(len(rows), dtype=torch.float64, device="cuda")
    alive = ~terminal
    steps = torch.zeros(len(rows), dtype=torch.int64, device="cuda")
    for t in range(H):
        est = torch.where(alive, est + reward[t].to(torch.float64) * (float(gamma) ** t), est)
        steps += alive
        alive = alive & ~done[t]
    return est, steps


def main():
    for kind, H in ((2, 600), (3, 200), (1, 300)):
        env = BatchedDMPEnv(kind, True, 1 << 16, seed=1)
        env.reset()
        env.rollout(5, obs=None)
        rows = torch.randint(0, 1 << 16, (4096,), device="cuda")
        for name, fn in (("evaluate()", lambda: env.evaluate(rows, H, 0.99)), ("round 5's loop", lambda: old_sums(env, rows, H, 0.99))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                est, steps = fn()
            torch.cuda.synchronize()
            print("%dD 4096 leaves H=%d %-16s %.2f ms per call  (mean steps %.1f)" % (kind, H, name, (time.perf_counter() - t0) / 5 * 1e3, steps.double().mean().item()), flush=True)


if __name__ == "__main__":
    main()
