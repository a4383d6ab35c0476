"""The canonical 3D snac_step per tick: k_step3dq (16 envs per wave, four lanes per env) against k_step3d / k_step3ds, one subprocess per arm.

    gpurun -- python tools/step3dq_time.py [f32]
"""
import sys

import retune

ARMS = [("k_step3dq", {"SNAC_STEP3D_QUARTER_MIN": "4", "SNAC_STEP3D_QUARTER_MAX": "100000000"}), ("without it", {"SNAC_STEP3D_QUARTER": "0"})]


def main():
    f32 = int("f32" in sys.argv[1:])
    for n in (1024, 4096, 16384, 32768, 65536, 98304, 131072, 262144, 524288):
        line = "N = %6d %s" % (n, "f32" if f32 else "f64")
        for name, env in ARMS:
            r = retune.run(dict(kind=3, T=1, f32=f32, layout=None, mode="step"), n, env)
            line += "   %s: %-10s %7.2f us" % (name, r["kernel"], r["ms"] * 1e3)
        print(line, flush=True)


if __name__ == "__main__":
    main()
