"""Back-to-back against isolated launches: the same fused rollout 12 times with HIP events round every launch, once with nothing
between the launches (what bench.py does) and once with a device synchronisation + 20 ms pause before each.
python tools/b2b_time.py [kind N T]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv  # noqa: E402


def run(kind, n, T, pause):
    env = BatchedDMPEnv(kind, True, n, seed=1)
    env.reset()
    out = torch.empty((T, n, env.obs_dim), dtype=torch.float64, device="cuda")
    rew = torch.empty((T, n), dtype=torch.float32, device="cuda")
    done = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        env.rollout(T, out=out, reward_out=rew, done_out=done)
    torch.cuda.synchronize()
    ev = []
    for i in range(12):
        if pause:
            torch.cuda.synchronize()
            time.sleep(0.02)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        env.rollout(T, out=out, reward_out=rew, done_out=done)
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    print("%dD N=%d T=%d %-12s %s" % (kind, n, T, "isolated" if pause else "back-to-back", " ".join("%.3f" % m for m in ms)), flush=True)


def main():
    cfgs = [(3, 16384, 1000), (2, 65536, 600), (1, 65536, 750)]
    if len(sys.argv) > 3:
        cfgs = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))]
    for kind, n, T in cfgs:
        run(kind, n, T, True)
        run(kind, n, T, False)
        run(kind, n, T, True)


if __name__ == "__main__":
    main()
