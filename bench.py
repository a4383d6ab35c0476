"""bench.py -- env-steps/s of the fused rollout on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): 2D dynamic dense (Env/2D/DMP_Env_2D_dynamic_usedata_plan.py, plans = the converted
data_2d_dynamic_dense_envplan_500_train set), 65 536 envs per GPU, seed 1.  One bench "step" is one pass of the
reference driver loop (multiprocess.py:82-84): T = total_step = 600 vector steps with uniform random actions,
step sizes and plan indices from the counter RNG, auto-reset, writing the float64 observation, the reward and the
done flag of EVERY env-step to HBM -- one snac_rollout launch.  Inputs (state, plan table) are resident in HBM
before the timed region.  Multi-GPU: envs are sharded by global id (weak scaling, no data-path collective); the
only collective is one RCCL all-reduce of three int64 episodic sums per pass, inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic bytes per env-step (DESIGN.md "Roofline")
ALG_BYTES = {(2, "f64"): 481, (2, "f32"): 277, (1, "f64"): 88, (3, "f64"): 574}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(kind, dynamic, n, T, seed):
    """The C oracle (oracle/snac_oracle.c, OpenMP over the host cores) on the same workload, timed on the host:
    one full pass of n envs x T steps in chunks, float64 observations of every step written to a reused buffer."""
    import numpy as np

    from oracle import snac_oracle
    from snac_amd import plans

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    full = plans.dataset(kind, "dense", "train") if dynamic else plans.static_plan(kind, 0)[None]
    table = full.reshape(len(full), -1).astype(np.int32)
    L = snac_oracle.lib()
    chunk = 10
    best = None
    # a few thread counts up to what the process may use (a shared host rarely scales to all of them), and a
    # single thread (the scalar port); report the fastest
    counts = sorted({c for c in (1, 8, 32, 64, avail) if c <= avail})
    for cores, budget in [(c, 5.0) for c in counts]:
        orc = snac_oracle.OracleBatch(kind, dynamic, n, table, seed=seed)
        orc.reset()
        obs = np.zeros((chunk, n, orc.obs_dim), np.float64)
        rew = np.zeros((chunk, n), np.float32)
        done = np.zeros((chunk, n), np.uint8)
        args = (None, None, obs.ctypes.data, 0, rew.ctypes.data, done.ctypes.data, cores)
        L.orc_batch_rollout(orc.b, chunk, 0, *args)  # warm-up chunk
        t0 = time.perf_counter()
        steps, t = 0, chunk
        while t < T and time.perf_counter() - t0 < budget:
            L.orc_batch_rollout(orc.b, chunk, t, *args)
            t += chunk
            steps += chunk * n
        rate = steps / (time.perf_counter() - t0)
        if best is None or rate > best["value"]:
            best = dict(value=rate, unit="env-steps/s", cores=cores, kind="port",
                        sample="C oracle (oracle/snac_oracle.c), %d OpenMP thread(s), %d envs x %d vector steps of the same "
                               "workload, f64 obs/reward/done of every step written" % (cores, n, steps // n))
        del orc
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--kind", type=int, default=2)
    ap.add_argument("--static", action="store_true")
    ap.add_argument("--T", type=int, default=0, help="vector steps per pass (default: total_step)")
    ap.add_argument("--obs-f32", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from snac_amd import BatchedDMPEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RCCL ("nccl" on ROCm) in production; SNAC_BENCH_BACKEND=gloo lets the N > 1 path be exercised with several
    # ranks sharing one GPU (tests): the three int64 sums then take a CPU round trip.
    backend = os.environ.get("SNAC_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    def allreduce_(t, op=None):
        op = op or dist.ReduceOp.SUM
        if world == 1:
            return t
        if backend == "nccl":
            dist.all_reduce(t, op=op)
            return t
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
        return t

    n = args.envs
    dynamic = not args.static
    env = BatchedDMPEnv(args.kind, dynamic, n, device=dev, seed=1, env_id_base=rank * n,
                        obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
    T = args.T or env.total_step
    obs = torch.empty((T, n, env.obs_dim), dtype=env.obs_dtype, device=dev)
    env.reset()
    stats = torch.zeros(3, dtype=torch.int64, device=dev)

    def one_pass():
        env.rollout(T, obs="all", out=obs)
        s = allreduce_(env.stats_tensor())  # RCCL over xGMI: episodic [episodes, return sum, IoU fixed-point sum]
        stats.copy_(s)

    for _ in range(args.warmup):
        one_pass()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    pending = []
    for i in range(args.steps):
        ev[i][0].record()
        env.rollout(T, obs="all", out=obs)
        ev[i][1].record()
        s = env.stats_tensor()
        if world > 1 and backend == "nccl":
            # the pass's one exchange (24 bytes, RCCL): enqueued behind the rollout on RCCL's stream, it overlaps the next
            # pass instead of holding it up; every pass still performs it and all are complete before the clock stops
            try:
                pending.append((dist.all_reduce(s, op=dist.ReduceOp.SUM, async_op=True), s))
            except Exception:                                   # keep the measurement alive: fall back to the blocking form
                stats.copy_(allreduce_(s))
        else:
            stats.copy_(allreduce_(s))
    for work, s in pending:
        work.wait()
    if pending:
        stats.copy_(pending[-1][1])
    sync()
    dt = time.perf_counter() - t0
    dt = float(allreduce_(torch.tensor([dt], dtype=torch.float64, device=dev), dist.ReduceOp.MAX).item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)

    if rank == 0:
        total_steps = world * n * T * args.steps
        dkey = "f32" if args.obs_f32 else "f64"
        alg = ALG_BYTES.get((args.kind, dkey), ALG_BYTES[(2, "f64")])
        achieved = alg * n * T / (kern_ms * 1e-3) / 1e9
        s = stats.tolist()
        # measured HBM bytes per launch (rocprofv3 PMC passes of this same command, tools/profile.sh ->
        # profiles/traffic.json), turned into GB/s with the live launch duration; null for other workloads
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile) and (args.kind, dynamic, n, T, dkey) == (2, True, 65536, 600, "f64"):
            with open(tfile) as fh:
                traffic = json.load(fh)["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec at N=65536 envs (2D dynamic dense); bit-exact vs CPU",
            "value": total_steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if not args.obs_f32 else "f32",
            "data": "synthetic",
            "config": {"workload": "%dD %s %s, %d envs/GPU, %d vector steps/pass, uniform random actions (counter RNG), "
                                   "auto-reset, %s obs of every step written" % (
                                       args.kind, "dynamic" if dynamic else "static", "dense", n, T, dkey),
                       "envs_per_gpu": n, "vector_steps_per_pass": T, "env_steps_per_pass": n * T * world,
                       "parallelism": "env-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_rollout", "kernel_ms": kern_ms, "alg_bytes_per_env_step": alg},
            "episodic": {"episodes": s[0], "mean_return": (s[1] / s[0]) if s[0] else None,
                         "mean_iou": (s[2] / 2.0 ** 40 / s[0]) if s[0] else None},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.kind, dynamic, n, T, 1)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
