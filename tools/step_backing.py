"""Does the BACKING of a per-tick step's row buffer matter the way it does for a rollout's trajectory?

For each case: k_step* writing its (N, obs_dim) rows into (a) a torch.empty tensor (one physical run of hipMalloc memory) and (b) the head
of a measured two-slice trajectory block (snac_amd/trajmem.py).  Prints the median launch time of 5 groups each.

    gpurun -- python tools/step_backing.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snac_amd import BatchedDMPEnv, _lib, trajmem  # noqa: E402


def timed(fn, reps=200):
    for _ in range(reps // 3):
        fn()
    per = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        per.append(a.elapsed_time(b) / reps * 1e3)
    return sorted(per)[2]


def main():
    cases = [(2, 65536, torch.float64, None), (2, 131072, torch.float64, None), (2, 262144, torch.float64, None), (2, 524288, torch.float64, None),
             (2, 524288, torch.float32, None), (3, 131072, torch.float64, None), (3, 524288, torch.float64, None), (3, 524288, torch.float32, None),
             (2, 65536, torch.float64, "ppo"), (2, 131072, torch.float64, "ppo"), (1, 524288, torch.float64, None)]
    for kind, n, dt, layout in cases:
        e = BatchedDMPEnv(kind, True, n, seed=1, obs_dtype=dt, **({"layout": layout} if layout else {}))
        e.reset()
        rw = torch.empty(n, dtype=torch.float32, device="cuda")
        dn = torch.empty(n, dtype=torch.uint8, device="cuda")
        res = []
        for backing in ("torch", "traj"):
            if backing == "torch":
                obs = torch.empty((n, e.obs_dim), dtype=dt, device="cuda")
            else:
                obs = trajmem.cached_empty((n, e.obs_dim), dt, "cuda")
            us = timed(lambda: e.step(auto_reset=True, out=(obs, rw, dn)))
            res.append((backing, us, _lib.lib().snac_last_kernel().decode()))
            del obs
        nbytes = n * e.obs_dim * (8 if dt == torch.float64 else 4)
        print("kind %d N %7d %s %-4s rows %6.1f MB   %s" % (kind, n, "f64" if dt == torch.float64 else "f32", layout or "-", nbytes / 1e6,
                                                           "   ".join("%s %7.2f us (%s)" % r for r in res)), flush=True)
        del e
        trajmem.cache_trim()


if __name__ == "__main__":
    main()
