"""One fused rollout timed back to back: kind, batch, dtype and the memory the trajectory lies in are arguments, so that two builds
or two kernels (SNAC_2D_STAGE=0 keeps 2D rollouts on the tile kernel) can be compared on one box, one process per arm.

    gpurun -- python tools/roll_time.py [kind] [N] [T] [reps] [f64|f32] [vmm|malloc] [tiled]
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
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    dt = torch.float32 if (len(sys.argv) > 5 and sys.argv[5] == "f32") else torch.float64
    mem = sys.argv[6] if len(sys.argv) > 6 else "vmm"
    tiled = len(sys.argv) > 7 and sys.argv[7] == "tiled"
    P = int(os.environ.get("SNAC_ROLL_PLANS", "0"))              # > 0: a table of P generated plans instead of the 400 stored ones
    layout = os.environ.get("SNAC_ROLL_LAYOUT") or None          # ppo | lnet2d | lnet1d: the layout variants of the descriptor
    kw = {}
    if P:
        import numpy as np
        kw["plans"] = np.zeros((P, 30) if kind == 1 else (P, 26, 26)) + (20 if kind == 1 else 0)
    if layout:
        kw["layout"] = layout
    env = BatchedDMPEnv(kind, True, n, seed=1, obs_dtype=dt, **kw)
    if P:
        env.generate_plans(0, P, seed=5)
    env.reset()
    T = T or env.total_step
    shape = ((n + 63) // 64, T, 64, env.obs_dim) if tiled else (T, n, env.obs_dim)
    buf = trajmem.traj_empty(shape, dt, "cuda") if mem == "vmm" else torch.empty(shape, dtype=dt, device="cuda")
    rew = torch.empty((T, n), dtype=torch.float32, device="cuda")
    done = torch.empty((T, n), dtype=torch.uint8, device="cuda")

    def run(k):
        ev = []
        for _ in range(k):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            env.rollout(T, obs="tiled" if tiled else "all", out=buf, reward_out=rew, done_out=done)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in ev)

    import time
    t_end = time.perf_counter() + 0.06                           # ~60 ms of the workload itself: clocks up (small batches: many launches)
    while time.perf_counter() < t_end:
        run(8)
    t = run(reps)
    from snac_amd import _lib
    kern = _lib.lib().snac_last_kernel().decode()
    esz = 4 if dt == torch.float32 else 8
    wb = (env.obs_dim * esz + 5) * n * T
    med = t[len(t) // 2]
    print("%dD %s N=%d T=%d %s %s%s stage=%s table=%s plans=%d layout=%s (%d values): min %.3f  median %.3f ms   %.2f TB/s written   %.3e env-steps/s" % (
        kind, kern, n, T, "f32" if esz == 4 else "f64", mem, " tiled" if tiled else "", os.environ.get("SNAC_2D_STAGE", "1"),
        os.environ.get("SNAC_2D_TABLE", "-"), env.num_plans, layout, env.obs_dim, t[0], med,
        wb / med / 1e9, n * T / med * 1e3), flush=True)


if __name__ == "__main__":
    main()
