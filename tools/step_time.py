"""snac_step timed on the device: `reps` per-tick steps back to back into preallocated outputs, events round the whole run (at
N = 524 288 a tick is tens of microseconds, the launches queue up behind each other).  One process per arm:
SNAC_STEP_STAGE=0 keeps the steps on k_transition2d / k_transition3d.

    gpurun -- python tools/step_time.py [kind] [N] [reps] [f64|f32] [explicit]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv  # noqa: E402

STATE = {1: 16 + 4 + 64, 2: 16 + 4 + 80, 3: 16 + 4 + 800}


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    dt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.float64
    explicit = len(sys.argv) > 5 and sys.argv[5] == "explicit"
    vmm = len(sys.argv) > 6 and sys.argv[6] == "vmm"             # the observation rows in a two-slice block of snac_traj_alloc
    env = BatchedDMPEnv(kind, True, n, seed=1, obs_dtype=dt)
    env.reset()
    if vmm:
        from snac_amd import trajmem

        rows = (1 << 30) // (n * env.obs_dim * (4 if dt == torch.float32 else 8)) + 1      # blocks under 1 GiB are one plain run
        obs = trajmem.traj_empty((rows, n, env.obs_dim), dt, env.device)[0]
    else:
        obs = torch.empty((n, env.obs_dim), dtype=dt, device=env.device)
    out = (obs, torch.empty(n, dtype=torch.float32, device=env.device), torch.empty(n, dtype=torch.uint8, device=env.device))
    acts = torch.randint(0, env.num_actions, (n,), dtype=torch.int8, device=env.device) if explicit else None
    ks = torch.randint(1, 4, (n,), dtype=torch.int8, device=env.device) if explicit else None

    def run(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k):
            env.step(acts, ks, auto_reset=True, out=out)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / k

    run(300)
    t = sorted(run(reps) for _ in range(5))
    esz = 4 if dt == torch.float32 else 8
    contract = {1: 88, 2: 481, 3: 574}[kind] - (0 if esz == 8 else (28 if kind == 1 else 204))   # SURVEY 8d's figures (bench.py CONTRACT_BYTES)
    print("%dD step N=%d %s %s%s stage=%s: min %.2f  median %.2f us/tick   %.3e env-steps/s   contract figure %.2f TB/s" % (
        kind, n, "f32" if esz == 4 else "f64", "explicit a,k" if explicit else "counter RNG", " obs in vmm" if vmm else "", os.environ.get("SNAC_STEP_STAGE", "1"),
        t[0] * 1e3, t[2] * 1e3, n / t[2] * 1e3, contract * n / t[2] / 1e9), flush=True)


if __name__ == "__main__":
    main()
