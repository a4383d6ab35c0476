"""A/B of the 2D tile size in ONE process on ONE trajectory tensor: libsnac_hip.so loaded twice (a copy under another name has its
own statics), the copy's first launch made with SNAC_TILE set.    gpurun -- python tools/tile_ab.py [tile] [N] [T]"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import snac_amd._lib as L  # noqa: E402
from snac_amd import BatchedDMPEnv, trajmem  # noqa: E402


def main():
    tile = sys.argv[1] if len(sys.argv) > 1 else "32"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 600
    os.environ.pop("SNAC_TILE", None)
    ea = BatchedDMPEnv(2, True, n, seed=1)
    ea.reset()
    ea.rollout(2, obs=None)                                      # lib A has read its (unset) SNAC_TILE
    copy = os.path.join(ROOT, "gpurun_out", "libsnac_tile_b.so")
    os.makedirs(os.path.dirname(copy), exist_ok=True)
    shutil.copyfile(L.LIB_PATH, copy)
    L._lib, L.LIB_PATH = None, copy
    os.environ["SNAC_TILE"] = tile
    eb = BatchedDMPEnv(2, True, n, seed=1)
    eb.reset()
    eb.rollout(2, obs=None)                                      # lib B: tile forced
    os.environ.pop("SNAC_TILE", None)
    bufs = [trajmem.traj_empty((T, n, 51), torch.float64, "cuda") for _ in range(2)]
    rew = torch.empty((T, n), dtype=torch.float32, device="cuda")
    done = torch.empty((T, n), dtype=torch.uint8, device="cuda")

    def run(env, buf, k, want_done):
        ev = []
        for _ in range(k):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            env.rollout(T, obs="all", out=buf, reward_out=rew, done_out=done if want_done else None, want_done=want_done)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in ev)

    run(ea, bufs[0], 25, True)
    for rnd in range(2):
        for i, buf in enumerate(bufs):
            for name, env in (("default tile", ea), ("tile " + tile, eb)):
                for wd in (True, False):
                    run(env, buf, 2, wd)
                    t = run(env, buf, 10, wd)
                    print("round %d tensor %d %-13s done %-5s min %.3f median %.3f ms" % (rnd, i, name, wd, t[0], t[len(t) // 2]), flush=True)


if __name__ == "__main__":
    main()
