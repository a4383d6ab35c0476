"""Workload for profiling the single-step kernels (k_transition2d / 3d): snac_step at N = 524 288 (2D and 3D, 50 ticks, counter RNG) and
snac_transition on a 2^20-row node pool (2D and 3D: random parents x all actions, 20 launches each).  Run under rocprofv3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv  # noqa: E402


def main():
    env = BatchedDMPEnv(2, True, 524288, seed=1)
    env.reset()
    out = (torch.empty((env.num_envs, 51), dtype=torch.float64, device="cuda"), torch.empty(env.num_envs, dtype=torch.float32, device="cuda"),
           torch.empty(env.num_envs, dtype=torch.uint8, device="cuda"))
    for _ in range(50):
        env.step(auto_reset=True, out=out)
    torch.cuda.synchronize()
    del env, out
    env = BatchedDMPEnv(3, True, 524288, seed=1)                     # 3D snac_step: k_transition3d on identity rows
    env.reset()
    out = (torch.empty((env.num_envs, 51), dtype=torch.float64, device="cuda"), torch.empty(env.num_envs, dtype=torch.float32, device="cuda"),
           torch.empty(env.num_envs, dtype=torch.uint8, device="cuda"))
    for _ in range(50):
        env.step(auto_reset=True, out=out)
    torch.cuda.synchronize()
    del env, out
    for kind in (2, 3):
        pool = BatchedDMPEnv(kind, True, 1 << 20, seed=1)
        pool.reset()
        pool.rollout(40, obs=None)                               # used states in the first half of the pool
        A = pool.num_actions
        parents = 1 << 16
        g = torch.Generator(device="cuda").manual_seed(kind)
        src = torch.randint(0, 1 << 19, (parents,), generator=g, device="cuda", dtype=torch.int32).repeat_interleave(A)
        dst = (1 << 19) + torch.arange(parents * A, device="cuda", dtype=torch.int32)
        acts = torch.arange(A, device="cuda", dtype=torch.int8).repeat(parents)
        ks = torch.randint(1, 4, (parents * A,), generator=g, device="cuda").to(torch.int8)
        for _ in range(20):
            pool.transition(acts, ks, src=src, dst=dst)
        torch.cuda.synchronize()
        del pool


if __name__ == "__main__":
    main()
