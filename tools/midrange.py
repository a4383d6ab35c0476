"""2D float64 rollouts of 19 457 .. 32 768 envs (whole episodes into the memory rollout() itself uses): every kernel / tile size that can
take them, one subprocess per arm.   gpurun -- python tools/midrange.py [N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import retune  # noqa: E402

ARMS = [("default", {}),
        ("k_rollout2dt", {"SNAC_2D_TP_MAX": "10000000"}),
        ("tile kernel, 8 envs/wave", {"SNAC_2D_TP": "0", "SNAC_2D_STAGE": "0", "SNAC_TILE": "8"}),
        ("tile kernel, 16", {"SNAC_2D_TP": "0", "SNAC_2D_STAGE": "0", "SNAC_TILE": "16"}),
        ("tile kernel, 32", {"SNAC_2D_TP": "0", "SNAC_2D_STAGE": "0", "SNAC_TILE": "32"}),
        ("tile kernel, 64", {"SNAC_2D_TP": "0", "SNAC_2D_STAGE": "0", "SNAC_TILE": "64"}),
        ("k_rollout2d (64 envs/wave)", {"SNAC_2D_TP": "0", "SNAC_2D_STAGE_MIN": "1"})]


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [20480, 24576, 28672, 32768]
    work = dict(kind=2, T=0, f32=0, layout=None, mode="rollout")
    for n in sizes:
        print("N = %d (600 ticks, %.2f GB of rows)" % (n, n * 600 * 408 / 1e9), flush=True)
        for name, env in ARMS:
            r = retune.run(work, n, env)
            print("   %-28s %-14s %7.4f ms   %5.2f TB/s" % (name, r["kernel"], r["ms"], n * 600 * 413.4 / r["ms"] / 1e9), flush=True)


if __name__ == "__main__":
    main()
