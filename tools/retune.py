"""Re-measure the crossovers behind libsnac_hip.so's dispatch table on THIS box and print the overrides that would move them.

The table (snac_amd/csrc/snac_hip.hip KNOBS, `python -c "from snac_amd import _lib; print(_lib.tuning())"`) holds the batch sizes at
which a call switches from one kernel to another; its defaults were measured on the MI355X boxes of the build pool.  For every
crossover below this tool times BOTH kernels at a few batch sizes round the default (one subprocess per arm: the table is read once
per process; the arm is forced with the switch of the same table), reports the two times and where the faster kernel changes, and
prints `export NAME=value` lines for crossovers that moved by more than one grid step.

    gpurun -- python tools/retune.py [quick] [only=SUBSTRING ...]     (quick: fewer batch sizes, ~2 minutes; only=: the entries whose name contains SUBSTRING); rollouts are whole episodes (T = total_step) into the memory rollout() itself would use
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch
from snac_amd import BatchedDMPEnv, _lib
kind, n, T, f32, layout, mode = %(kind)d, %(n)d, %(T)d, %(f32)d, %(layout)r, %(mode)r
dt = torch.float32 if f32 else torch.float64
kw = {"layout": layout} if layout else {}
if os.environ.get("SNAC_RETUNE_TAIL"): kw["obs_tail"] = tuple(os.environ["SNAC_RETUNE_TAIL"].split(","))   # (tools/var2d_time.py: rows with a tail)
e = BatchedDMPEnv(kind, layout != "lnet2d", n, seed=1, obs_dtype=dt, **kw)
e.reset()
def timed(fn, reps):
    for _ in range(max(3, reps // 3)): fn()
    per = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        per.append(a.elapsed_time(b) / reps)
    return sorted(per)[2]
if mode == "rollout":
    T = T or e.total_step
    obs = e._traj_out((T, n, e.obs_dim))       # where rollout() itself puts its rows: a measured trajectory block from 1 GiB on
    rw = torch.empty((T, n), dtype=torch.float32, device="cuda"); dn = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    ms = timed(lambda: e.rollout(T, obs="all", out=obs, reward_out=rw, done_out=dn), max(3, int(40 / max(1, n * T * e.obs_dim * 8 / 6e9))))
else:
    out = (torch.empty((n, e.obs_dim), dtype=dt, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda"))
    ms = timed(lambda: e.step(auto_reset=True, out=out), 200)
print(json.dumps({"ms": ms, "kernel": _lib.lib().snac_last_kernel().decode()}))
'''

# (name of the table entry that holds the crossover, what is compared, workload, env of arm A, env of arm B, direction, batch sizes)
#   direction "max": kernel A is used UP TO the entry's value; "min": kernel A is used FROM the entry's value
CROSSOVERS = [
    # (the three entries below order the kernels BEHIND k_rollout2db: both arms run with SNAC_2D_BLOCK=0)
    ("SNAC_2D_TP_MAX_F64", "k_rollout2dt | tile kernel / k_rollout2d, 2D float64 rows", dict(kind=2, T=0, f32=0, layout=None, mode="rollout"),
     {"SNAC_2D_TP_MAX": "10000000", "SNAC_2D_BLOCK": "0"}, {"SNAC_2D_TP": "0", "SNAC_2D_BLOCK": "0"}, "max", [12288, 14336, 16384, 18432, 20480, 24576, 28672]),
    ("SNAC_2D_TP_MAX_F32", "k_rollout2dt | tile kernel / k_rollout2d, 2D float32 rows", dict(kind=2, T=0, f32=1, layout=None, mode="rollout"),
     {"SNAC_2D_TP_MAX": "10000000", "SNAC_2D_BLOCK": "0"}, {"SNAC_2D_TP": "0", "SNAC_2D_BLOCK": "0"}, "max", [20480, 24576, 28672, 30720, 32768, 40960]),
    ("SNAC_2D_STAGE_MIN_F64", "k_rollout2d | tile kernel, 2D float64 rows", dict(kind=2, T=0, f32=0, layout=None, mode="rollout"),
     {"SNAC_2D_STAGE_MIN": "1", "SNAC_2D_TP": "0", "SNAC_2D_BLOCK": "0"}, {"SNAC_2D_STAGE": "0", "SNAC_2D_TP": "0", "SNAC_2D_BLOCK": "0"}, "min", [24576, 28672, 32768, 36864, 40960, 49152]),
    ("SNAC_1D_TP_MAX_F64", "k_rollout1dt | tile kernel, 1D float64 rows", dict(kind=1, T=0, f32=0, layout=None, mode="rollout"),
     {"SNAC_1D_TP_MAX": "10000000"}, {"SNAC_1D_TP": "0"}, "max", [32768, 40960, 49152, 57344, 65536, 81920]),
    ("SNAC_3D_BLOCK_MIN_F64", "k_rollout3db | k_rollout3d, 3D float64 rows", dict(kind=3, T=0, f32=0, layout=None, mode="rollout"),
     {"SNAC_3D_BLOCK_MIN": "0"}, {"SNAC_3D_BLOCK": "0"}, "min", [2048, 4096, 6144, 8192, 12288]),
    ("SNAC_STEP_VAR_FULL_F64", "k_step2d<VAR> | k_transition, 2D PPO rows per tick", dict(kind=2, T=1, f32=0, layout="ppo", mode="step"),
     {"SNAC_STEP_VAR_MIN": "1", "SNAC_STEP_VAR_HALF": "0"}, {"SNAC_STEP_VAR_MIN": "100000000"}, "min", [32768, 40960, 45056, 49152, 65536]),
    ("SNAC_STEP_VAR3_MIN", "k_step3d<VAR> | k_transition, 3D PPO rows per tick", dict(kind=3, T=1, f32=0, layout="ppo", mode="step"),
     {"SNAC_STEP_VAR3_MIN": "1"}, {"SNAC_STEP_VAR3_MIN": "100000000"}, "min", [8192, 16384, 24576, 32768, 65536]),
    ("SNAC_2D_BLOCK_MIN_F64", "k_rollout2db (a stepper wave + eight writer waves per 64 envs) | k_rollout2dt / tile kernel, 2D float64 rows", dict(kind=2, T=0, f32=0, layout=None, mode="rollout"),
     {"SNAC_2D_BLOCK_MIN_F64": "4", "SNAC_2D_BLOCK_MAX_F64": "100000000"}, {"SNAC_2D_BLOCK": "0"}, "min", [4096, 6144, 8192, 10240, 12288, 16384, 20480, 24576]),
    ("SNAC_2D_BLOCK_MAX_F64", "k_rollout2db | tile kernel / k_rollout2d, 2D float64 rows", dict(kind=2, T=0, f32=0, layout=None, mode="rollout"),
     {"SNAC_2D_BLOCK_MIN_F64": "4", "SNAC_2D_BLOCK_MAX_F64": "100000000"}, {"SNAC_2D_BLOCK": "0"}, "max", [28672, 32768, 40960, 49152, 57344, 65536]),
    ("SNAC_2D_BLOCK_MIN_F32", "k_rollout2db | k_rollout2dt, 2D float32 rows", dict(kind=2, T=0, f32=1, layout=None, mode="rollout"),
     {"SNAC_2D_BLOCK_MIN_F32": "4", "SNAC_2D_BLOCK_MAX_F32": "100000000"}, {"SNAC_2D_BLOCK": "0"}, "min", [4096, 8192, 12288, 16384, 24576]),
    ("SNAC_2D_BLOCK_MAX_F32", "k_rollout2db | k_rollout2d, 2D float32 rows", dict(kind=2, T=0, f32=1, layout=None, mode="rollout"),
     {"SNAC_2D_BLOCK_MIN_F32": "4", "SNAC_2D_BLOCK_MAX_F32": "100000000"}, {"SNAC_2D_BLOCK": "0"}, "max", [28672, 32768, 40960, 49152, 65536]),
    ("SNAC_3D_BLOCK_VAR_PLAN_F64", "k_rollout3db (plan rows in LDS, handed over by the stepper) | tile kernel, 3D float64 rows with the plan tail, 200 ticks",
     dict(kind=3, T=200, f32=0, layout="ppo", mode="rollout"), {"SNAC_3D_BLOCK_VAR_PLAN_F64": "4"}, {"SNAC_3D_BLOCK_VAR": "0"}, "min", [6144, 8192, 10240, 12288, 16384, 32768]),
    ("SNAC_3D_BLOCK_VAR_PLAN_F32", "the same, float32 rows", dict(kind=3, T=200, f32=1, layout="ppo", mode="rollout"),
     {"SNAC_3D_BLOCK_VAR_PLAN_F32": "4"}, {"SNAC_3D_BLOCK_VAR": "0"}, "min", [8192, 10240, 12288, 14336, 16384, 32768, 65536]),
    ("SNAC_STEP3D_SPAN_MIN", "k_step3ds (cooperative span loads) | k_step3d, 3D canonical rows per tick", dict(kind=3, T=1, f32=0, layout=None, mode="step"),
     {"SNAC_STEP3D_SPAN_MIN": "4"}, {"SNAC_STEP3D_SPAN": "0"}, "min", [32768, 65536, 81920, 98304, 131072, 262144]),
    # round 6: which FORM of a step kernel a batch size takes (form bits: 1 = non-temporal state loads, 2 = non-temporal row stores; what fits the Infinity Cache)
    ("SNAC_STEP3D_NTLOAD_MIN", "k_step3dq: form 1 (non-temporal span loads + plain rows) | form 2 (resident: plain loads + non-temporal rows), 3D canonical rows per tick",
     dict(kind=3, T=1, f32=0, layout=None, mode="step"), {"SNAC_STEP3D_FORM": "1"}, {"SNAC_STEP3D_FORM": "2"}, "min", [262144, 327680, 360448, 376832, 393216, 458752]),
    ("SNAC_STEP3D_HUGE_MIN", "k_step3dq: form 3 (both non-temporal) | form 1", dict(kind=3, T=1, f32=0, layout=None, mode="step"),
     {"SNAC_STEP3D_FORM": "3"}, {"SNAC_STEP3D_FORM": "1"}, "min", [491520, 524288, 557056, 589824, 655360]),
    ("SNAC_STEP2D_PLAIN_LO", "k_step2d: form 2 (resident) | form 1 (round 5's), 2D canonical rows per tick", dict(kind=2, T=1, f32=0, layout=None, mode="step"),
     {"SNAC_STEP2D_FORM": "2"}, {"SNAC_STEP2D_FORM": "1"}, "min", [12288, 16384, 20480, 24576, 32768, 49152]),
    ("SNAC_STEP2D_RES_HI", "k_step2d: form 2 (resident) | form 0 (plain loads, plain rows)", dict(kind=2, T=1, f32=0, layout=None, mode="step"),
     {"SNAC_STEP2D_FORM": "2"}, {"SNAC_STEP2D_FORM": "0"}, "max", [229376, 262144, 278528, 294912, 327680]),
    ("SNAC_STEP2D_HUGE_MIN", "k_step2d: form 2 (resident) | form 0, the upper end of form 0's range (SNAC_STEP2D_PLAIN_HI = this - 1)", dict(kind=2, T=1, f32=0, layout=None, mode="step"),
     {"SNAC_STEP2D_FORM": "2"}, {"SNAC_STEP2D_FORM": "0"}, "min", [393216, 458752, 475137, 491520, 524288]),
    ("SNAC_1D_TP_EB8_MIN", "k_rollout1dt in blocks of 8 envs | blocks of 4, 1D float64 rows", dict(kind=1, T=0, f32=0, layout=None, mode="rollout"),
     {"SNAC_1D_TP_EB8_MIN": "1", "SNAC_1D_TP_EB16": "100000000"}, {"SNAC_1D_TP_EB8_MIN": "100000000", "SNAC_1D_TP_EB16": "100000000"}, "min", [1024, 2048, 3072, 3584, 4096, 8192]),
    ("SNAC_1D_LANE_MIN_F64", "k_rollout1dl (lane = env) | k_rollout1dt, 1D float64 rows", dict(kind=1, T=0, f32=0, layout=None, mode="rollout"),
     {"SNAC_1D_LANE_MIN_F64": "1"}, {"SNAC_1D_LANE": "0"}, "min", [32768, 36864, 40960, 45056, 49152, 57344]),
    ("SNAC_1D_LANE_MIN_F32", "k_rollout1dl (lane = env) | k_rollout1dt, 1D float32 rows", dict(kind=1, T=0, f32=1, layout=None, mode="rollout"),
     {"SNAC_1D_LANE_MIN_F32": "1"}, {"SNAC_1D_LANE": "0"}, "min", [28672, 32768, 36864, 40960, 45056, 49152]),
    ("SNAC_1D_LANE_VAR_MIN", "k_rollout1dl<VARLD> | k_rollout1dt<VAR>, 1D 37-value PPO rows", dict(kind=1, T=0, f32=0, layout="ppo", mode="rollout"),
     {"SNAC_1D_LANE_VAR_MIN": "1"}, {"SNAC_1D_LANE": "0"}, "min", [16384, 20480, 24576, 28672, 32768, 40960]),
    ("SNAC_1D_LANE_VAR_SHORT_MIN", "k_rollout1dl<VARLD> | k_rollout1dt<VAR>, 1D 8-value L-Net rows", dict(kind=1, T=0, f32=0, layout="lnet1d", mode="rollout"),
     {"SNAC_1D_LANE_VAR_SHORT_MIN": "1"}, {"SNAC_1D_LANE": "0"}, "min", [32768, 40960, 45056, 49152, 57344, 65536]),
]


def run(work, n, env):
    e = dict(os.environ)
    e.update(env)
    code = WORKER % dict(root=ROOT, n=n, **work)
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    if out.returncode:
        raise RuntimeError(out.stderr[-800:])
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    quick = "quick" in sys.argv[1:]
    only = [a[5:] for a in sys.argv[1:] if a.startswith("only=")]      # only=SUBSTRING: the crossovers whose entry name contains it
    sys.path.insert(0, ROOT)
    from snac_amd import _lib

    table = _lib.tuning()
    exports = []
    for name, what, work, env_a, env_b, direction, sizes in CROSSOVERS:
        if only and not any(o in name for o in only):
            continue
        if quick:
            sizes = sizes[1:-1:2] if len(sizes) > 4 else sizes[::2]
        cur = table[name][0]
        print("== %s (now %d): %s" % (name, cur, what), flush=True)
        rows = []
        for n in sizes:
            a, b = run(work, n, env_a), run(work, n, env_b)
            rows.append((n, a["ms"], b["ms"]))
            print("   N = %7d   %-14s %8.4f ms    %-14s %8.4f ms    %s" % (n, a["kernel"], a["ms"], b["kernel"], b["ms"], "A" if a["ms"] <= b["ms"] else "B"), flush=True)
        # the crossover: direction max -> the largest N up to which A wins everywhere below; min -> the smallest N from which A wins everywhere above
        if direction == "max":
            good = [n for n, x, y in rows if x <= y]
            bad = [n for n, x, y in rows if x > y]
            new = max([n for n in good if not any(m < n for m in bad)], default=None)
            moved = new is not None and not (new <= cur < min([m for m in bad if m > new], default=1 << 30))
        else:
            good = [n for n, x, y in rows if x <= y]
            bad = [n for n, x, y in rows if x > y]
            new = min([n for n in good if not any(m > n for m in bad)], default=None)
            moved = new is not None and not (max([m for m in bad if m < new], default=0) < cur <= new)
        if new is None:
            print("   -> kernel A wins at none of these sizes on this box (entry left alone)")
        elif moved:
            print("   -> crossover on this box: %d (the table says %d)" % (new, cur))
            exports.append("export %s=%d" % (name, new))
        else:
            print("   -> the table's value stands")
    print("\n# overrides for this box (none: the table's defaults hold here)")
    for ln in exports:
        print(ln)


if __name__ == "__main__":
    main()
