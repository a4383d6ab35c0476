"""The same fused rollout into the two kinds of memory, in one process: torch.empty (hipMalloc: one contiguous physical run) against
snac_traj_alloc (HIP virtual-memory API: two physical runs 32 GiB apart, 32 MB chunks taking turns).  Launches back to back after a
warm-up; median of `reps` per tensor, two tensors of each kind, two rounds.

    gpurun -- python tools/mem_ab.py [kind] [N] [T] [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv, trajmem  # noqa: E402


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    env = BatchedDMPEnv(kind, True, n, seed=1)
    env.reset()
    T = T or env.total_step
    shape = (T, n, env.obs_dim)
    bufs = [("hipMalloc", torch.empty(shape, dtype=torch.float64, device="cuda")), ("virtual memory", trajmem.traj_empty(shape, torch.float64, "cuda")),
            ("hipMalloc", torch.empty(shape, dtype=torch.float64, device="cuda")), ("virtual memory", trajmem.traj_empty(shape, torch.float64, "cuda"))]
    rew = torch.empty((T, n), dtype=torch.float32, device="cuda")
    done = torch.empty((T, n), dtype=torch.uint8, device="cuda")

    def run(buf, k):
        ev = []
        for _ in range(k):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            env.rollout(T, obs="all", out=buf, reward_out=rew, done_out=done)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in ev)

    run(bufs[0][1], 25)                                          # clocks up
    wb = (61 if kind == 1 else 413) * n * T                  # observation row + reward + done per env-step
    for rnd in range(2):
        for i, (name, buf) in enumerate(bufs):
            run(buf, 3)
            t = run(buf, reps)
            print("%dD N=%d T=%d round %d tensor %d %-15s min %.3f  median %.3f ms   %.2f TB/s written (median)" % (
                kind, n, T, rnd, i, name, t[0], t[len(t) // 2], wb / t[len(t) // 2] / 1e9), flush=True)


if __name__ == "__main__":
    main()
