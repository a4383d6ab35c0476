"""3D rollouts with a layout variant: k_rollout3db<VAR> (round 5) against the tile kernel (SNAC_3D_BLOCK_VAR=0), T ticks into trajectory memory.

    gpurun -- 'SNAC_3D_BLOCK_VAR_MIN=4 SNAC_3D_BLOCK_VAR_PLAN_F64=4 SNAC_3D_BLOCK_VAR_PLAN_F32=4 python tools/var3d_time.py && SNAC_3D_BLOCK_VAR=0 python tools/var3d_time.py'
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snac_amd import BatchedDMPEnv, _lib  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    per = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        per.append(a.elapsed_time(b) / reps)
    return sorted(per)[2]


CASES = [(True, 64, 1000, torch.float64, dict(obs_tail=("record",))), (True, 256, 1000, torch.float64, dict(obs_tail=("record",))),
         (True, 1024, 1000, torch.float64, dict(obs_tail=("record",))), (True, 4096, 1000, torch.float64, dict(obs_tail=("record",))),
         (True, 16384, 1000, torch.float64, dict(obs_tail=("record",))), (True, 65536, 300, torch.float64, dict(obs_tail=("record",))),
         (True, 65536, 300, torch.float32, dict(obs_tail=("position", "record"), obs_scalars="raw")), (False, 16384, 1000, torch.float64, dict(obs_scalars="norm")),
         (True, 8192, 200, torch.float64, dict(layout="ppo")), (True, 12288, 200, torch.float64, dict(layout="ppo")), (True, 14336, 200, torch.float64, dict(layout="ppo")),
         (True, 16384, 200, torch.float64, dict(layout="ppo")), (True, 32768, 100, torch.float64, dict(layout="ppo")), (True, 65536, 50, torch.float64, dict(layout="ppo")),
         (False, 16384, 200, torch.float64, dict(obs_tail=("plan",))), (True, 16384, 200, torch.float32, dict(layout="ppo")), (True, 65536, 50, torch.float32, dict(layout="ppo"))]


def main():
    print("SNAC_3D_BLOCK_VAR =", os.environ.get("SNAC_3D_BLOCK_VAR", "(default)"))
    for dyn, n, T, dt, kw in CASES:
        e = BatchedDMPEnv(3, dyn, n, seed=1, obs_dtype=dt, **kw)
        e.reset()
        obs = e._traj_out((T, n, e.obs_dim))
        rw = torch.empty((T, n), dtype=torch.float32, device="cuda")
        dn = torch.empty((T, n), dtype=torch.uint8, device="cuda")
        nbytes = T * n * (e.obs_dim * obs.element_size() + 5)
        ms = timed(lambda: e.rollout(T, obs="all", out=obs, reward_out=rw, done_out=dn), max(2, int(30e9 / nbytes)))
        print("%s N %6d T %4d %s dim %3d  %-14s %8.4f ms  %6.0f GB/s  %.3f of 8 TB/s" % ("dyn" if dyn else "sta", n, T, "f64" if dt == torch.float64 else "f32", e.obs_dim,
              _lib.lib().snac_last_kernel().decode(), ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000), flush=True)
        del obs, e


if __name__ == "__main__":
    main()
