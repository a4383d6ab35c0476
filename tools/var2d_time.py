"""2D rollouts with a layout variant without the plan tail (L-Net rows, rows with position / record tails): k_rollout2db's variant form
against what the table would otherwise pick (k_rollout2dt up to 6144 envs, the tile kernel, k_rollout2d from 65 536 envs); 600 ticks into
trajectory memory, one subprocess per arm (tools/retune.py's worker).

    gpurun -- python tools/var2d_time.py
"""
import retune

ARMS = [("64-env blocks", {"SNAC_2D_BLOCK_VAR_MIN": "4", "SNAC_2D_BLOCK_VAR_MAX": "100000000", "SNAC_2D_BLOCK_VAR_TWO": "100000000"}),
        ("128-env blocks", {"SNAC_2D_BLOCK_VAR_MIN": "4", "SNAC_2D_BLOCK_VAR_MAX": "100000000", "SNAC_2D_BLOCK_VAR_TWO": "4"}), ("without it", {"SNAC_2D_BLOCK_VAR_MIN": "100000000"})]
CASES = [("lnet2d", 0, n) for n in (4096, 8192, 16384, 24576, 65536)] + [("record", 0, n) for n in (2048, 4096, 6144, 8192, 16384, 32768, 49152, 65536)] + [("record", 1, 16384), ("record", 1, 65536)]


def main():
    for name, f32, n in CASES:
        layout = "lnet2d" if name == "lnet2d" else None
        line = "%-7s %s N = %6d " % (name, "f32" if f32 else "f64", n)
        dim = 51 if name == "lnet2d" else 59
        for arm, env in ARMS:
            work = dict(kind=2, T=0, f32=f32, layout=layout, mode="rollout")
            e = dict(env)
            if name == "record":
                e["SNAC_RETUNE_TAIL"] = "record"
            r = retune.run(work, n, e)
            line += "  %s: %-12s %7.4f ms %5.2f TB/s" % (arm, r["kernel"], r["ms"], n * 600 * (dim * (4 if f32 else 8) + 5) / r["ms"] / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
