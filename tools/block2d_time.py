"""2D rollouts of the middle batches: k_rollout2db with one / two / four stepper waves per block against what the table would otherwise pick
(k_rollout2dt, the tile kernel, k_rollout2d), whole episodes (600 ticks) into trajectory memory; one subprocess per arm (tools/retune.py's worker).

    gpurun -- python tools/block2d_time.py [f32] [N ...]
"""
import sys

import retune

ARMS = [("64-env blocks", {"SNAC_2D_BLOCK_MIN_F64": "4", "SNAC_2D_BLOCK_MAX_F64": "100000000", "SNAC_2D_BLOCK_TWO_F64": "100000000",
                           "SNAC_2D_BLOCK_MIN_F32": "4", "SNAC_2D_BLOCK_MAX_F32": "100000000", "SNAC_2D_BLOCK_TWO_F32": "100000000", "SNAC_2D_BLOCK_FOUR_MAX_F64": "0", "SNAC_2D_BLOCK_FOUR_MAX_F32": "0"}),
        ("128-env blocks", {"SNAC_2D_BLOCK_MIN_F64": "4", "SNAC_2D_BLOCK_MAX_F64": "100000000", "SNAC_2D_BLOCK_TWO_F64": "4",
                            "SNAC_2D_BLOCK_MIN_F32": "4", "SNAC_2D_BLOCK_MAX_F32": "100000000", "SNAC_2D_BLOCK_TWO_F32": "4", "SNAC_2D_BLOCK_FOUR_MAX_F64": "0", "SNAC_2D_BLOCK_FOUR_MAX_F32": "0"}),
        ("256-env blocks", {"SNAC_2D_BLOCK_MIN_F64": "4", "SNAC_2D_BLOCK_MAX_F64": "3", "SNAC_2D_BLOCK_FOUR_MAX_F64": "100000000",
                            "SNAC_2D_BLOCK_MIN_F32": "4", "SNAC_2D_BLOCK_MAX_F32": "3", "SNAC_2D_BLOCK_FOUR_MAX_F32": "100000000"}),
        ("without k_rollout2db", {"SNAC_2D_BLOCK": "0"})]


def main():
    f32 = "f32" in sys.argv[1:]
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8192, 12288, 16384, 20480, 24576, 28672, 32768, 40960, 49152, 65536]
    esz = 4 if f32 else 8
    for n in sizes:
        line = "N = %6d %s " % (n, "f32" if f32 else "f64")
        for name, env in ARMS:
            r = retune.run(dict(kind=2, T=0, f32=int(f32), layout=None, mode="rollout"), n, env)
            line += "  %s: %-12s %7.4f ms %5.2f TB/s" % (name, r["kernel"], r["ms"], n * 600 * (51 * esz + 5) / r["ms"] / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
