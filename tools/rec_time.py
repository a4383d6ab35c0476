"""Cost of the per-step record outputs of snac_rollout_rec (action, step size, plan row, first-step flag): one fused rollout
with and without them, every observation written.  python tools/rec_time.py [kind N T]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv  # noqa: E402


def main():
    cfgs = [(3, 16384, 1000), (2, 65536, 600), (1, 65536, 750), (1, 4096, 750)]
    if len(sys.argv) > 3:
        cfgs = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))]
    for kind, n, T in cfgs:
        env = BatchedDMPEnv(kind, True, n, seed=1)
        env.reset()
        out = torch.empty((T, n, env.obs_dim), dtype=torch.float64, device="cuda")
        rew = torch.empty((T, n), dtype=torch.float32, device="cuda")
        done = torch.empty((T, n), dtype=torch.uint8, device="cuda")
        rec = {"actions": torch.empty((T, n), dtype=torch.int8, device="cuda"), "step_size": torch.empty((T, n), dtype=torch.int8, device="cuda"),
               "plan_idx": torch.empty((T, n), dtype=torch.int16, device="cuda"), "first": torch.empty((T, n), dtype=torch.uint8, device="cuda")}
        for name, r in (("without record", None), ("with record", rec)):
            best = 1e9
            for i in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                env.rollout(T, out=out, reward_out=rew, done_out=done, record=r)
                b.record()
                torch.cuda.synchronize()
                if i >= 2:
                    best = min(best, a.elapsed_time(b))
            print("%dD dynamic N=%d T=%d %-15s %.3f ms" % (kind, n, T, name, best), flush=True)
        del env, out, rew, done, rec


if __name__ == "__main__":
    main()
