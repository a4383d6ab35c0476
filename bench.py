"""bench.py -- env-steps/s of the fused rollout on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts the N ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): 2D dynamic dense (Env/2D/DMP_Env_2D_dynamic_usedata_plan.py, plans = the converted
data_2d_dynamic_dense_envplan_500_train set), 65 536 envs per GPU, seed 1.  One bench "step" is one pass of the
reference driver loop (multiprocess.py:82-84): T = total_step = 600 vector steps with uniform random actions,
step sizes and plan indices from the counter RNG, auto-reset, writing the float64 observation, the reward and the
done flag of EVERY env-step to HBM -- one snac_rollout launch.  Inputs (state, plan table) are resident in HBM
before the timed region.  Multi-GPU: envs are sharded by global id (weak scaling, no data-path collective); the
only collective is one RCCL all-reduce of three int64 episodic sums per pass, inside the timed region.

`--gpus N` is binding: without WORLD_SIZE in the environment the process becomes a launcher (it never touches the
GPU) that starts N rank processes -- the one-command form of the reference driver (multiprocess.py:89-97); with
WORLD_SIZE set (torch.distributed.run) it must equal N, anything else is a non-zero exit.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# SURVEY.md section 8d's per-env-step figure: an UN-FUSED step (state and window read, scalars written back); f32 = the same with
# 4-byte observations.  Reported as roofline.contract -- the fused launch keeps state and window on chip, so since the trajectory
# memory went in this figure / time comes out ABOVE the 8 TB/s peak, which no launch can be
CONTRACT_BYTES = {(2, "f64"): 481, (2, "f32"): 277, (1, "f64"): 88, (1, "f32"): 60, (3, "f64"): 574, (3, "f32"): 370}
# per-env state a launch loads once and stores once: header 16 + episode 4 + grid record + episodic sums 24
STATE_BYTES = {1: 16 + 4 + 64 + 24, 2: 16 + 4 + 80 + 24, 3: 16 + 4 + 800 + 24}
# bytes the fused rollout really writes per env-step: the observation row + reward (4) + done (1); state stays on chip
WRITTEN_BYTES = {(2, "f64"): 413, (2, "f32"): 209, (1, "f64"): 61, (1, "f32"): 33, (3, "f64"): 413, (3, "f32"): 209}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HEADLINE = "env-steps/sec at N=65536 envs (2D dynamic dense); bit-exact vs CPU"


def parse_args(argv=None):
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
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: --gpus N without a torch.distributed environment
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, script=None, argv=None):
    """Start n rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), forward rank 0's JSON line and
    return the worst exit status.  This process has not initialised the GPU and never does.  (script / argv: the rank program and
    its arguments, for tests/test_bench_launcher.py, which starts eight stand-in ranks on the CPU.)"""
    script = script or os.path.abspath(__file__)
    argv = sys.argv[1:] if argv is None else list(argv)
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env, cwd=os.getcwd(),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    # rank 0 prints one short JSON line (far below the pipe buffer), so polling without draining cannot block it
    rc = 0
    while any(p.poll() is None for p in procs):
        failed = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if failed:                                                # one rank died: the others would wait in a collective
            rc = failed[0]
            for p in procs:
                if p.poll() is None:
                    p.kill()                                      # the exact children we started, nothing else
        time.sleep(0.05)
    out0 = procs[0].stdout.read()
    for p in procs:
        rc = rc or p.wait()
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if rc == 0 and len(lines) != 1:
        rc = 3
    for ln in lines:
        print(ln)
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------
# CPU baselines (SURVEY.md section 8d i-iii); rank 0 at N = 1 only
def _table(kind, dynamic):
    import numpy as np

    from snac_amd import plans

    full = plans.dataset(kind, "dense", "train") if dynamic else plans.static_plan(kind, 0)[None]
    return full.reshape(len(full), -1).astype(np.int32)


def _oracle_rate(kind, dynamic, n, T, seed, cores, budget):
    import numpy as np

    from oracle import snac_oracle

    L = snac_oracle.lib()
    chunk = 10
    orc = snac_oracle.OracleBatch(kind, dynamic, n, _table(kind, dynamic), seed=seed)
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
    return steps / (time.perf_counter() - t0), steps // n


def oracle_first_ticks(kind, dynamic, n, seed, env_id_base, ticks, cores):
    """What the CPU oracle says about the first `ticks` vector steps of this rank's shard (same seed, plan table, env ids as the
    measured batch): the reset observation, then obs [ticks, n, D] float64, reward [ticks, n] float32, done [ticks, n] uint8.
    The checker of the bench line's `parity_vs_oracle` (the oracle is test infrastructure: it checks, it is never the product)."""
    from oracle import snac_oracle

    orc = snac_oracle.OracleBatch(kind, dynamic, n, _table(kind, dynamic), seed=seed, env_id_base=env_id_base)
    first = orc.reset()
    o, r, d = orc.rollout(ticks, nthreads=cores)
    return first, o, r, d


def _python_loop_rate(kind, dynamic, seed, n=1024, budget=5.0):
    """(iii) the per-env Python loop shaped like VectorizedEnvWrapper.step (multiprocess.py:24-32): n independent env
    objects, a for-loop calling step() on each, np.asarray of the collected lists; the step size drawn per step with
    np.random.randint(1, 4) as the reference classes do; reset on done.  The env objects are oracle.OracleEnv (this leg
    is the checker, not the product)."""
    import numpy as np

    from oracle import snac_oracle

    table = _table(kind, dynamic)
    rs = np.random.RandomState(seed)
    envs = [snac_oracle.OracleEnv(kind, dynamic) for _ in range(n)]
    for e in envs:
        e.reset(table[rs.randint(0, len(table))])
    A = envs[0].e.num_actions
    ticks = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        actions = rs.randint(A, size=n)
        obs, rewards, dones = [], [], []
        for i, e in enumerate(envs):
            o, r, d = e.step(actions[i], rs.randint(1, 4))
            if d:
                e.reset(table[rs.randint(0, len(table))])
            obs.append(o), rewards.append(r), dones.append(d)
        np.asarray(obs), np.asarray(rewards), np.asarray(dones)
        ticks += 1
    return n * ticks / (time.perf_counter() - t0), ticks


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(kind, dynamic, n, T, seed):
    """The C oracle (oracle/snac_oracle.c) on the same workload, timed on the host: envs x T steps in chunks, float64
    observations of every step written to a reused buffer.  `value` = the best OpenMP thread count (i); also the single
    thread (ii) and the per-env Python loop at N = 1024 (iii)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a few thread counts up to what the process may use (a shared host rarely scales to all of them)
    counts = sorted({c for c in (1, 8, 32, 64, avail) if c <= avail})
    best, single = None, None
    what = "C oracle (oracle/snac_oracle.c), %d OpenMP thread(s), %d envs x %d vector steps of the same workload, " \
           "f64 obs/reward/done of every step written"
    for cores in counts:
        rate, ticks = _oracle_rate(kind, dynamic, n, T, seed, cores, 4.0)
        rec = dict(value=rate, unit="env-steps/s", cores=cores, kind="port", sample=what % (cores, n, ticks))
        if cores == 1:
            single = dict(value=rate, unit="env-steps/s", cores=1, sample=rec["sample"])
        if best is None or rate > best["value"]:
            best = rec
    best["single_thread"] = single
    best["cpu_model"] = _cpu_model()
    best["cores_available"] = avail
    rate, ticks = _python_loop_rate(kind, dynamic, seed)
    best["python_loop_n1024"] = dict(
        value=rate, unit="env-steps/s", cores=1,
        sample="python for-loop over 1024 env objects per vector step (shape of multiprocess.py:24-32), %d vector steps; the env objects "
               "are the C oracle behind ctypes, so this is an UPPER bound for that loop shape -- the reference's own pure-Python "
               "classes do 1.1-1.2e5 env-steps/s in it (BASELINE.md section 2)" % ticks)
    return best


# ------------------------------------------------------------------------------------------------
# the stdout line: compact, strict JSON, a few KB (tests/test_gpu_bench.py holds it below 6000 bytes)
def _sig(x, digits=6):
    """Floats to `digits` significant digits (recursively); NaN / inf -> None so the line stays strict JSON."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return str(x)


EXTRA_COLS = ["kernel_or_path", "us", "frac"]


def _extra_row(e):
    """One extra configuration as [kernel or path, microseconds per launch / tick / call, fraction of the 8 TB/s peak or null]."""
    if "kernel_ms" in e:
        us = e["kernel_ms"] * 1e3
    elif "us_per_tick" in e:
        us = e["us_per_tick"]
    elif "us_per_step" in e:
        us = e["us_per_step"]
    elif "us_per_call" in e:
        us = e["us_per_call"]
    else:
        us = e.get("us_per_vector_step")
    name = e.get("kernel") or ("mailbox" if str(e.get("path", "")).startswith("mailbox") else "launch")
    return [name, us, e.get("frac")]


def compact_line(out, extra_file):
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "backend", "rccl_ranks", "collective_check", "rccl_async_exchanges", "retimed", "ranks", "kernel_ms_per_rank",
            "trajectory_check", "trajectory_full_pass_check", "parity_vs_oracle", "ranks_on_distinct_devices", "episodic")
    line = {k: out[k] for k in keep if k in out}
    r = out["roofline"]
    line["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "alg_bytes_per_env_step",
                                          "peak_measured_write")}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": "C oracle, %d OpenMP threads, same workload, %s" % (cb["cores"], cb["sample"].split("thread(s), ", 1)[-1].split(" of the same", 1)[0]),
                                "cpu_model": cb.get("cpu_model"), "cores_available": cb.get("cores_available"),
                                "single_thread": {"value": (cb.get("single_thread") or {}).get("value")},
                                "python_loop_n1024": {"value": (cb.get("python_loop_n1024") or {}).get("value")}}
    cfgs = (out.get("extra") or {}).get("configs")
    if isinstance(cfgs, dict) and "error" not in cfgs:
        line["extra_cols"] = EXTRA_COLS
        line["extra"] = {k: _extra_row(v) for k, v in cfgs.items()}
    elif isinstance(cfgs, dict):
        line["extra"] = cfgs
    if isinstance(out.get("tiled_layout"), dict) and "kernel_ms" in out["tiled_layout"]:
        line["tiled_layout_kernel_ms"] = out["tiled_layout"]["kernel_ms"]
    line["timed_regions_ms_per_step"] = [t["ms_per_step"] for t in out.get("timed_regions", [])]   # both, when the region was retimed
    line["extra_file"] = os.path.basename(extra_file) if extra_file else None
    return _sig(line)


# ------------------------------------------------------------------------------------------------
def measured_write_peak(torch, dev, nbytes=1 << 31, ms_budget=50.0, alloc=None):
    """The box's write ceiling, live: hipMemsetAsync over a 2 GiB buffer on the current stream, ~50 ms of it, timed with
    events on that stream.  alloc: where the buffer comes from (default torch.empty = hipMalloc memory; snac_amd.trajmem.traj_empty =
    memory of the virtual-memory API, the kind the trajectory lies in).  GB/s or None."""
    import ctypes as C

    hip = None
    for name in ("libamdhip64.so.7", "libamdhip64.so"):          # already loaded by torch: dlopen returns that copy
        try:
            hip = C.CDLL(name)
            break
        except OSError:
            continue
    if hip is None or not hasattr(hip, "hipMemsetAsync"):
        return None
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    hip.hipMemsetAsync.restype = C.c_int
    buf = alloc((nbytes,), torch.uint8, dev) if alloc else torch.empty(nbytes, dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def run(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k):
            if hip.hipMemsetAsync(C.c_void_p(buf.data_ptr()), 0, nbytes, stream) != 0:
                return None
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    if run(2) is None:
        return None
    one = run(4)
    if not one:
        return None
    k = max(4, int(ms_budget / (one / 4)))
    ms = run(k)
    del buf
    return nbytes * k / (ms * 1e-3) / 1e9 if ms else None


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))                          # before any GPU call in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a number for the wrong job size\n"
                         % (args.gpus, world))
        sys.exit(2)

    import torch
    import torch.distributed as dist

    from snac_amd import BatchedDMPEnv, _lib

    # RCCL ("nccl" on ROCm) in production; SNAC_BENCH_BACKEND=gloo lets the N > 1 path be exercised with several
    # ranks sharing one GPU (tests): the three int64 sums then take a CPU round trip.
    backend = os.environ.get("SNAC_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        sys.stderr.write("bench.py: %d RCCL ranks need %d GPUs, this node shows %d\n" % (world, world, ndev))
        sys.exit(2)
    local = local % max(ndev, 1) if backend != "nccl" else local
    # SNAC_BENCH_FORCE_DIST=1: a ONE-rank run initialises the process group all the same and takes the N > 1 code path for every
    # exchange (RCCL group of one: the async all_reduce on RCCL's stream, work.wait(), the barriers) -- the way to execute that path
    # on a box with one GPU.  The reduced sums must then equal the local ones (`collective_check`).
    use_dist = world > 1 or os.environ.get("SNAC_BENCH_FORCE_DIST", "0") == "1"
    if use_dist:
        kw = {}
        if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:   # (forced one-rank group without a launcher)
            kw = dict(init_method="tcp://127.0.0.1:%d" % _free_port(), rank=rank, world_size=world)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), **kw)   # before any GPU call of this process
        else:
            dist.init_process_group(backend, **kw)
        assert dist.get_world_size() == args.gpus, "process group size %d != --gpus %d" % (dist.get_world_size(), args.gpus)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    def allreduce_(t, op=None):
        op = op or dist.ReduceOp.SUM
        if not use_dist:
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
    # Where the 16 GB trajectory tensor lies in HBM is worth 5-9 % of the pass (fast and slow regions of the address map,
    # snac_amd/placement.py): SNAC_BENCH_PLACE candidates (default 3, 0 / 1 = take the first allocation) are
    # allocated, the workload itself -- a scratch batch of the same shape -- is timed on each, the fastest is kept.
    # What those regions are (tools/wr_blocks.hip): slices of 32 GiB of the physical address space -- streams inside one slice write at
    # ~5.7 TB/s, spread over several at ~7.1 -- and a hipMalloc tensor is one physical run.  snac_traj_alloc (snac_amd/trajmem.py) backs
    # ONE virtual range with three runs a slice apart: SNAC_BENCH_MEMORY=vmm (default) | malloc; if such a block cannot be had the run
    # falls back to torch.empty and says so in `placement`.
    # N > 1 ranks: one block per rank (every candidate is an allocation + a probe per rank, inside the driver's time limit)
    place_n = int(os.environ.get("SNAC_BENCH_PLACE", "3" if world == 1 else "1"))
    memory = os.environ.get("SNAC_BENCH_MEMORY", "vmm")
    traj_alloc = None
    if memory == "vmm":
        try:
            from snac_amd import trajmem

            trajmem.traj_empty((1 << 20,), torch.uint8, dev)       # one small block: does this box / build do it at all?
            traj_alloc = trajmem.traj_empty
        except Exception as e:
            sys.stderr.write("bench.py: no virtual-memory trajectory block (%r), using torch.empty\n" % (e,))
            memory = "malloc (vmm failed: %r)" % (e,)

    def plain(shape):
        return traj_alloc(shape, env.obs_dtype, dev) if traj_alloc else torch.empty(shape, dtype=env.obs_dtype, device=dev)

    placement_report = None
    if place_n > 1:
        from snac_amd import placement

        try:
            probe = BatchedDMPEnv(args.kind, dynamic, n, device=dev, seed=3, env_id_base=rank * n,
                                  obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
            probe.reset()
            obs, placement_report = placement.fastest_tensor((T, n, env.obs_dim), env.obs_dtype, dev,
                                                             lambda t: probe.rollout(T, obs="all", out=t), candidates=place_n,
                                                             alloc=traj_alloc)
            del probe
        except Exception as e:                                   # the measurement must not depend on the probe: take a plain tensor
            sys.stderr.write("bench.py: placement probe failed (%r), using the first allocation\n" % (e,))
            torch.cuda.empty_cache()
            try:
                obs, placement_report = plain((T, n, env.obs_dim)), {"error": repr(e)}
            except Exception as e2:
                traj_alloc, memory = None, "malloc (vmm failed: %r)" % (e2,)
                obs, placement_report = torch.empty((T, n, env.obs_dim), dtype=env.obs_dtype, device=dev), {"error": repr(e)}
    else:
        obs = plain((T, n, env.obs_dim))
        placement_report = {}
    placement_report["memory"] = memory
    env.reset()
    stats = torch.zeros(3, dtype=torch.int64, device=dev)

    def one_pass():
        env.rollout(T, obs="all", out=obs)
        s = allreduce_(env.stats_tensor())  # RCCL over xGMI: episodic [episodes, return sum, IoU fixed-point sum]
        stats.copy_(s)

    # The write-ceiling probe (~60 ms of hipMemsetAsync) runs FIRST, on every rank: a GPU that comes from idle needs some tens
    # of milliseconds of load before it holds its sustained clocks (tools/b2b_time.py: the same launch takes 1.51 ms right
    # after an idle gap and 1.29 ms ten launches later), and W warm-up passes of 3 ms each do not get it there.
    wpeak_malloc = measured_write_peak(torch, dev)
    wpeak_vmm = None
    if traj_alloc:
        try:
            wpeak_vmm = measured_write_peak(torch, dev, alloc=traj_alloc)
        except Exception as e:
            sys.stderr.write("bench.py: write-ceiling probe on virtual-memory block failed (%r)\n" % (e,))
    wpeak = max([w for w in (wpeak_malloc, wpeak_vmm) if w] or [0.0]) or None
    # ... and neither does the probe: what brings the clocks up is the workload itself, launched back to back.  So an untimed
    # pre-roll of SNAC_BENCH_PREROLL_MS (default 60 ms, 0 = none) of passes is enqueued before the W warm-up passes, without
    # a host synchronisation in between (tools/b2b_time.py).
    preroll_ms = float(os.environ.get("SNAC_BENCH_PREROLL_MS", "60"))
    preroll_passes = 0
    if preroll_ms > 0:
        # on a scratch batch of the same shape: the measured batch runs exactly W + K passes, whatever the pre-roll's length
        pre = BatchedDMPEnv(args.kind, dynamic, n, device=dev, seed=2, env_id_base=rank * n,
                            obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
        pre.reset()
        a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        pre.rollout(T, obs="all", out=obs)
        b0.record()
        torch.cuda.synchronize()
        preroll_passes = max(1, min(400, int(preroll_ms / max(a0.elapsed_time(b0), 1e-3))))
        for _ in range(preroll_passes):
            pre.rollout(T, obs="all", out=obs)
    for _ in range(args.warmup):
        one_pass()

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region():
        """EXACTLY K passes between a barrier + synchronize on either side; returns (wall seconds, MAX over ranks; per-pass kernel ms of this rank)."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        gc.collect()                                                # (no collector pause of the host thread inside the timed region)
        gc.disable()
        try:
            return _timed_region_body(ev)
        finally:
            gc.enable()

    def _timed_region_body(ev):
        sync()
        t0 = time.perf_counter()
        pending = []
        for i in range(args.steps):
            ev[i][0].record()
            env.rollout(T, obs="all", out=obs)
            ev[i][1].record()
            s = env.stats_tensor(out=stats) if not use_dist else env.stats_tensor()   # no group: the sums land where they are kept
            if not use_dist:
                continue
            if backend == "nccl":
                # the pass's one exchange (24 bytes, RCCL): enqueued behind the rollout on RCCL's stream, it overlaps the next
                # pass instead of holding it up; every pass still performs it and all are complete before the clock stops
                try:
                    pending.append((dist.all_reduce(s, op=dist.ReduceOp.SUM, async_op=True), s))
                except Exception:                               # keep the measurement alive: fall back to the blocking form
                    stats.copy_(allreduce_(s))
            else:
                stats.copy_(allreduce_(s))
        for work, s in pending:
            work.wait()
        if pending:
            stats.copy_(pending[-1][1])
        sync()
        dt_ = time.perf_counter() - t0
        dt_ = float(allreduce_(torch.tensor([dt_], dtype=torch.float64, device=dev), dist.ReduceOp.MAX).item())
        return dt_, [a.elapsed_time(b) for a, b in ev], len(pending)

    # The boxes of this pool stall for tens of milliseconds now and then (a group of bench extras: 362 instead of 8.6 us per launch, twice
    # in this round's runs; round 4's driver run: 396 instead of 38) -- as long as the whole timed region of the headline.  A timed region
    # that shows such a stall (its slowest pass > 2 x its median pass, or its wall time > 1.3 x the sum of its kernels + 1 ms per pass) is
    # timed ONCE more, by every rank together, and the second region is the one reported; both are in `timed_regions`, `retimed` says so.
    # SNAC_BENCH_RETIME=0 switches that off.
    def stalled(dt_, ms_):
        srt = sorted(ms_)
        med = srt[len(srt) // 2]
        wall = (not use_dist or backend == "nccl") and dt_ * 1e3 > 1.3 * sum(ms_) + 1.0 * len(ms_)   # (gloo: the exchange itself takes milliseconds of host time)
        return bool(srt[-1] > 2.0 * med or wall)

    dt, per_step_ms, n_async = timed_region()
    timed_regions = [{"ms_per_step": dt / args.steps * 1e3, "kernel_ms_per_step": [round(x, 4) for x in per_step_ms], "stalled": stalled(dt, per_step_ms)}]
    retimed = False
    if os.environ.get("SNAC_BENCH_RETIME", "1") != "0":
        flag = float(allreduce_(torch.tensor([1.0 if timed_regions[0]["stalled"] else 0.0], dtype=torch.float64, device=dev), dist.ReduceOp.MAX).item())
        if flag > 0:
            dt, per_step_ms, n_async = timed_region()
            timed_regions.append({"ms_per_step": dt / args.steps * 1e3, "kernel_ms_per_step": [round(x, 4) for x in per_step_ms], "stalled": stalled(dt, per_step_ms)})
            retimed = True
    # a group of one: what came back from the collective must be this rank's own sums
    collective_check = bool(torch.equal(stats, env.stats_tensor())) if (use_dist and world == 1) else None
    kern_ms = sum(per_step_ms) / max(args.steps, 1)
    # every rank's own kernel time (events on its launch stream), and the proof of how many ranks the collective saw
    per_rank = torch.zeros(world, dtype=torch.float64, device=dev)
    per_rank[rank] = kern_ms
    per_rank = allreduce_(per_rank).tolist()
    ranks_seen = int(allreduce_(torch.ones(1, dtype=torch.int64, device=dev)).item())
    # which physical GPU each rank ran on: PCI domain:bus:device packed into one int64 per rank, summed into place
    try:
        pr = torch.cuda.get_device_properties(local)
        packed = (int(getattr(pr, "pci_domain_id", 0)) << 16) | (int(getattr(pr, "pci_bus_id", 0)) << 8) | int(getattr(pr, "pci_device_id", 0))
    except Exception:
        packed = -1
    ids = torch.zeros(world, dtype=torch.int64, device=dev)
    ids[rank] = packed
    ranks_devices = ["cuda:%d pci %04x:%02x:%02x" % (i % max(ndev, 1) if backend != "nccl" else i, v >> 16, (v >> 8) & 0xff, v & 0xff) if v >= 0 else None
                     for i, v in enumerate(allreduce_(ids).tolist())]

    kernel_name = _lib.lib().snac_last_kernel().decode()        # what the timed launches went to, as the library's dispatch says

    # integrity of what the timed passes wrote (after the clock stopped): the last step's rows in the trajectory tensor must be the
    # batch's current observation, read through a different kernel into ordinary memory
    traj_ok = bool(torch.equal(obs[T - 1], env.observe()))
    if not traj_ok:
        sys.stderr.write("bench.py: the trajectory tensor's last step differs from observe() -- the measurement is INVALID\n")
    # ... "bit-exact vs CPU" as something THIS run proves: a fresh batch with the measured batch's seed, table and env ids makes one
    # full pass (the same launch shape, the same kernel) into the measured trajectory block, and its reset observation and the first
    # PARITY_TICKS ticks of observations, rewards and done flags are compared byte for byte with the CPU oracle's.  Every rank checks
    # its own shard; false anywhere marks the line invalid (exit status 4).  (The -m gpu tests compare whole passes; this is the
    # run's own certificate, a fraction of a second.)
    PARITY_TICKS = 10
    parity = None
    if os.environ.get("SNAC_BENCH_PARITY", "1") != "0":
        try:
            first, oo, ro, do = oracle_first_ticks(args.kind, dynamic, n, 1, rank * n, PARITY_TICKS, min(16, os.cpu_count() or 1))
            fresh = BatchedDMPEnv(args.kind, dynamic, n, device=dev, seed=1, env_id_base=rank * n,
                                  obs_dtype=torch.float32 if args.obs_f32 else torch.float64)
            g0 = fresh.reset()
            go, gr, gd = fresh.rollout(T, obs="all", out=obs)
            same_kernel = _lib.lib().snac_last_kernel().decode() == kernel_name
            cast = (lambda x: x.astype("float32")) if args.obs_f32 else (lambda x: x)
            parity = bool(same_kernel
                          and g0.cpu().numpy().tobytes() == cast(first).tobytes()
                          and go[:PARITY_TICKS].cpu().numpy().tobytes() == cast(oo).tobytes()
                          and gr[:PARITY_TICKS].cpu().numpy().tobytes() == ro.tobytes()
                          and gd[:PARITY_TICKS].cpu().numpy().view("uint8").tobytes() == do.tobytes())
            del fresh, go, gr, gd, g0, first, oo, ro, do
            if not parity:
                traj_ok = False
                sys.stderr.write("bench.py: rank %d: the first %d ticks differ from the CPU oracle -- the measurement is INVALID\n" % (rank, PARITY_TICKS))
        except Exception as e:
            sys.stderr.write("bench.py: oracle parity leg could not run (%r)\n" % (e,))
            parity = "not run: %r" % (e,)
    # every rank's own account, gathered into rank 0's line: kernel time, how its trajectory block is backed and how fast the
    # allocator measured it, its integrity and parity checks -- a slow or wrong rank is explainable from the line
    blk = None
    try:
        from snac_amd import trajmem as _tm

        blk = _tm.describe(obs)
    except Exception:
        pass
    lay_code = {"one run": 1, "three runs 32 GiB apart": 2, "measured: two slices in turn": 3}.get((blk or {}).get("layout"), 0)
    mine = torch.zeros((world, 8), dtype=torch.float64, device=dev)
    mine[rank] = torch.tensor([kern_ms, lay_code, (blk or {}).get("us_per_gib", {}).get("block", 0.0), (blk or {}).get("us_per_gib", {}).get("fast", 0.0),
                               (blk or {}).get("windows_slow", 0), (blk or {}).get("rebuilds", 0), 1.0 if traj_ok else 0.0,
                               1.0 if parity is True else (0.0 if parity is False else -1.0)], dtype=torch.float64)
    mine = allreduce_(mine).tolist()
    lay_names = {0: "hipMalloc", 1: "one run", 2: "three runs 32 GiB apart", 3: "measured: two slices in turn"}
    per_rank_report = [{"rank": i, "kernel_ms": round(r[0], 4), "block_layout": lay_names[int(r[1])], "block_us_per_gib": r[2], "fast_us_per_gib": r[3],
                        "windows_slow": int(r[4]), "rebuilds": int(r[5]), "trajectory_check": bool(r[6]), "parity_vs_oracle": (None if r[7] < 0 else bool(r[7]))}
                       for i, r in enumerate(mine)]
    all_ok = all(r["trajectory_check"] for r in per_rank_report)
    parity_all = None if any(r["parity_vs_oracle"] is None for r in per_rank_report) else all(r["parity_vs_oracle"] for r in per_rank_report)
    # N RCCL ranks must sit on N distinct GPUs
    devices_ok = backend != "nccl" or world == 1 or len(set(ranks_devices)) == world
    if not devices_ok:
        sys.stderr.write("bench.py: %d RCCL ranks on %d distinct GPUs (%r) -- the measurement is INVALID\n" % (world, len(set(ranks_devices)), ranks_devices))

    # ... and every row of one more pass against a SECOND kernel: a twin of the batch rolls out through the tile kernel k_rollout
    # (an output that is not 16-byte aligned cannot take k_rollout2d's 16-byte stores, snac_hip.hip roll2d_ok()) into ordinary memory;
    # all T x N rows, rewards and done flags of the two passes must be equal.  After the clock has stopped; one GPU, rank 0.
    full_check = None
    if world == 1 and os.environ.get("SNAC_BENCH_FULLCHECK", "1") != "0":
        try:
            twin = env.fork(torch.arange(n, device=dev))
            raw = torch.empty(T * n * env.obs_dim + 1, dtype=env.obs_dtype, device=dev)
            o1, r1, d1 = env.rollout(T, obs="all", out=obs)
            o2, r2, d2 = twin.rollout(T, obs="all", out=raw[1:].view(T, n, env.obs_dim))
            full_check = bool(o2.data_ptr() % 16 != 0 and torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
                              and torch.equal(env._hdr, twin._hdr) and torch.equal(env._grid, twin._grid))
            del twin, raw, o2, r2, d2, o1, r1, d1
            torch.cuda.empty_cache()
            if not full_check:
                traj_ok = False
                sys.stderr.write("bench.py: a full pass differs between the two rollout kernels -- the measurement is INVALID\n")
        except Exception as e:
            sys.stderr.write("bench.py: full-pass check could not run (%r)\n" % (e,))
            full_check = "not run: %r" % (e,)

    # the same workload into the tile-major trajectory layout (obs="tiled": [N / 64][T][64][D], SNAC_OBS_TILED) -- reported
    # beside the headline, never as `value`: 2D only, after the clock stopped; its tensor is placed like the headline's (the
    # tile-major stream is the one that really profits from a fast region: 5.65 against 7.1 TB/s store-only)
    def tiled_extra():
        nonlocal obs
        del obs
        torch.cuda.empty_cache()
        tshape = (n // 64, T, 64, env.obs_dim)
        if place_n > 1:
            tv, trep = placement.fastest_tensor(tshape, env.obs_dtype, dev, lambda t: env.rollout(T, obs="tiled", out=t), candidates=place_n,
                                                alloc=traj_alloc)
        else:
            tv, trep = plain(tshape), None
        for _ in range(12):
            env.rollout(T, obs="tiled", out=tv)
        tev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a_, b_ in tev:
            a_.record()
            env.rollout(T, obs="tiled", out=tv)
            b_.record()
        torch.cuda.synchronize()
        tms = sum(a_.elapsed_time(b_) for a_, b_ in tev) / max(args.steps, 1)
        return {"layout": "[N/64][T][64][obs_dim] (rollout(obs='tiled'))", "kernel_ms": tms, "value": n * T / (tms * 1e-3), "unit": "env-steps/s per GPU",
                 "written": WRITTEN_BYTES[(args.kind, "f32" if args.obs_f32 else "f64")] * n * T / (tms * 1e-3) / 1e9, "placement": trep}

    # ---- the other BASELINE configs, the float32 rows and the per-tick step(), driver-timed: after the headline clock has stopped,
    # rank 0 of a one-GPU run only, into the headline's own trajectory block (no further allocation); kernel time from events on the
    # launch stream, averaged over back-to-back launches.  `frac` prices the bytes the launch must move -- rows + reward + done
    # written, the state once in and once out (rollouts); SURVEY.md 8d's per-step figure (step(): the state does cross HBM every
    # tick there) -- against the 8 TB/s peak.
    def extra_configs():
        flat = obs.reshape(-1).view(torch.uint8)

        def view(shape, dt):
            nb = 1
            for d in shape:
                nb *= int(d)
            nb *= 4 if dt == torch.float32 else 8
            if nb > flat.numel():
                return None
            return flat[:nb].view(dt).view(shape)

        GROUPS = 5

        def timed(fn, reps):
            """Device time of one call of fn(): GROUPS groups of `reps` back-to-back calls, every group bracketed by its own pair of
            events on the launch stream; the MEDIAN group's mean is the figure, min / max are reported beside it, and a run whose
            slowest group is more than 1.5 x the median is marked `outlier` (round 4: one group of 100 launches once carried a single
            36 ms stall on the driver's box -- the SMI sampler, a clock event -- and read as 396 instead of 38 us per tick)."""
            for _ in range(max(3, reps // 4)):
                fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(GROUPS)]
            # the interpreter's cyclic garbage collector off while the launches are enqueued: a full collection in this process takes tens
            # of milliseconds, during which the device runs dry -- the "stall" of round 4's driver run and of this round's first runs (always
            # the same group of the same extra: the collector triggers on an allocation count); a stepping loop of 6000 launches on its own
            # shows no such gap
            gc.collect()
            gc.disable()
            try:
                for a_, b_ in evs:
                    a_.record()
                    for _ in range(reps):
                        fn()
                    b_.record()
                torch.cuda.synchronize()
            finally:
                gc.enable()
            in_order = [a_.elapsed_time(b_) / reps for a_, b_ in evs]
            per = sorted(in_order)
            spread.append({"min": per[0], "median": per[GROUPS // 2], "max": per[-1], "groups": GROUPS, "launches_per_group": reps,
                           "in_order": [round(x, 6) for x in in_order],
                           "outlier": bool(per[-1] > 1.5 * per[GROUPS // 2])})
            return per[GROUPS // 2]

        spread = []                                                # timed()'s last account: spread[-1] goes into the entry it belongs to

        def host_timed(loop, steps):
            """Host-timed loops (the single-env facade, the numpy wrapper): GROUPS repetitions of `steps` calls, median rate."""
            per = []
            for _ in range(GROUPS):
                t0 = time.perf_counter()
                loop(steps)
                per.append((time.perf_counter() - t0) / steps)
            per.sort()
            return per[GROUPS // 2], {"min_us": per[0] * 1e6, "median_us": per[GROUPS // 2] * 1e6, "max_us": per[-1] * 1e6, "groups": GROUPS,
                                      "calls_per_group": steps, "outlier": bool(per[-1] > 1.5 * per[GROUPS // 2])}

        res = {}

        def last_kernel():
            return _lib.lib().snac_last_kernel().decode()

        def rollout_cfg(name, kind, dyn, nn, f32, reps, note=None, plans=0, layout=None, TT=0, tail=None):
            import numpy as np

            dt = torch.float32 if f32 else torch.float64
            kw = {}
            if plans:
                kw["plans"] = np.zeros((plans, 26, 26))
            if layout:
                kw["layout"] = layout
            if tail:
                kw["obs_tail"] = tail
            e = BatchedDMPEnv(kind, dyn, nn, device=dev, seed=1, obs_dtype=dt, **kw)
            if plans:
                e.generate_plans(0, plans, seed=5)
            e.reset()
            TT = TT or e.total_step
            buf = view((TT, nn, e.obs_dim), dt)
            if buf is None:
                return
            rw = torch.empty((TT, nn), dtype=torch.float32, device=dev)
            dn = torch.empty((TT, nn), dtype=torch.uint8, device=dev)
            ms = timed(lambda: e.rollout(TT, obs="all", out=buf, reward_out=rw, done_out=dn), reps)
            esz = 4 if f32 else 8
            written = e.obs_dim * esz + 5                           # the row + reward + done
            algb = written + 2.0 * STATE_BYTES[kind] / TT
            gbs = algb * nn * TT / (ms * 1e-3) / 1e9
            res[name] = {"kernel": last_kernel(), "kernel_ms": ms, "vector_steps": TT, "values_per_row": e.obs_dim, "plans": e.num_plans,
                         "env_steps_per_s": nn * TT / (ms * 1e-3), "written_GBs": written * nn * TT / (ms * 1e-3) / 1e9,
                         "alg_bytes_per_env_step": algb, "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS, "launches": GROUPS * reps, "timing": spread[-1]}
            if note:
                res[name]["note"] = note

        def edges_cfg(name, kind, parents, reps, nodes=False):
            """snac_transition as one search wave (Env/2D/DMP_ENV_2D_dynamic_MCTS.py:117-175): `parents`
            random rows of a 2^20-row node pool, one child each with a random action, written to fresh rows with the child's
            observation.  Bytes per edge: the source record in, the destination record out, row + reward + done + indices."""
            import ctypes as C

            pool = 1 << 20
            e = BatchedDMPEnv(kind, True, pool, device=dev, seed=1)
            e.reset()
            e.rollout(20, obs=None)
            m = parents
            src = torch.randint(0, pool - m, (m,), device=dev, dtype=torch.int32)
            dst = (pool - m + torch.arange(m, device=dev, dtype=torch.int32)).contiguous()
            acts = torch.randint(0, e.num_actions, (m,), device=dev).to(torch.int8)
            ob = view((m, e.obs_dim), torch.float64)
            rw = torch.empty(m, dtype=torch.float32, device=dev)
            dn = torch.empty(m, dtype=torch.uint8, device=dev)
            vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
            recs = None
            if nodes:                                               # the same pool as ONE 128-byte record per node (snac_amd.NodePool2D, k_edges2dp)
                from snac_amd import NodePool2D

                recs = NodePool2D(e, pool)
                recs.load()

            def call():
                if recs is not None:
                    _lib.check(e._lib.snac_transition_nodes2d(C.byref(e._desc), C.byref(e._state), vp(recs.records), pool, m, vp(src), vp(dst), 0, vp(acts),
                                                              None, vp(ob), vp(rw), vp(dn), e._stream()))
                    return
                _lib.check(e._lib.snac_transition(C.byref(e._desc), C.byref(e._state), m, vp(src), vp(dst), 0, vp(acts), None, vp(ob),
                                                  vp(rw), vp(dn), e._stream()))

            ms = timed(call, reps)
            rec = {1: 64, 2: 80, 3: 800}[kind] + 20                  # grid record + header + episode counter
            algb = 2 * rec + e.obs_dim * 8 + 5 + 8 + 1
            gbs = algb * m / (ms * 1e-3) / 1e9
            res[name] = {"kernel": last_kernel(), "kernel_ms": ms, "edges_per_s": m / (ms * 1e-3), "alg_bytes_per_edge": algb, "achieved_GBs": gbs,
                         "frac": gbs / HBM_PEAK_GBS, "launches": GROUPS * reps, "timing": spread[-1], "note": "random parents in a 2^20-row pool, one child per parent into fresh rows"}

        def gather_cfg(name, nn, cap, batch, reps):
            """snac_replay_gather (row f1; script/DQN/2d/DQN_2d_dynamic.py:145-166): float32 (s, s', plan) minibatch of `batch` random
            transitions out of a float64 ring of `cap` ticks x `nn` envs."""
            from snac_amd import ReplayRing

            e = BatchedDMPEnv(2, True, nn, device=dev, seed=1)
            e.reset()
            ring = ReplayRing(e, cap)
            ring.collect(cap)
            slot = torch.randint(1, cap, (batch,), device=dev)
            ei = torch.randint(0, nn, (batch,), device=dev)
            import ctypes as C
            import itertools

            # NEW samples every launch, as a trainer draws them: one repeated set would keep its 70 MB of rows in the 256 MB Infinity Cache
            # from launch to launch (35 us against 39: tools/gather_probe.py).  32 sets = 2.2 GB of rows between two uses of a set.
            sets = itertools.cycle([(torch.randint(1, cap, (batch,), device=dev).to(torch.int32), torch.randint(0, nn, (batch,), device=dev).to(torch.int32))
                                    for _ in range(32)])
            so = torch.empty((batch, e.obs_dim), dtype=torch.float32, device=dev)
            sn = torch.empty_like(so)
            pl = torch.empty((batch, 400), dtype=torch.float32, device=dev)
            vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

            def call():
                slot32, ei32 = next(sets)
                _lib.check(e._lib.snac_replay_gather(C.byref(e._desc), C.byref(e._state), cap, vp(ring.obs), vp(ring.first), vp(ring.plan_idx),
                                                     vp(slot32), vp(ei32), batch, vp(so), vp(sn), vp(pl), e._stream()))

            ms = timed(call, reps)
            tm = spread[-1]
            ms_wrapped = timed(lambda: ring.gather(slot, ei), reps)
            algb = 2 * 408 + 2 * 204 + 1600 + 8 + 3                  # two f64 rows in, two f32 rows + 400 f32 plan cells out, indices, flags
            gbs = algb * batch / (ms * 1e-3) / 1e9
            res[name] = {"kernel": "k_gather", "kernel_ms": ms, "samples_per_s": batch / (ms * 1e-3), "alg_bytes_per_sample": algb, "achieved_GBs": gbs,
                         "frac": gbs / HBM_PEAK_GBS, "launches": GROUPS * reps, "timing": tm, "wrapper_ms": ms_wrapped,
                         "note": "kernel_ms: snac_replay_gather alone (raw C ABI), new random samples every launch; wrapper_ms: ReplayRing.gather, which also gathers action / reward / done with torch"}

        def facade_cfg(name, steps=3000):
            """The single-env drop-in class as a DQN script drives it (Env/2D/DMP_Env_2D_dynamic_usedata_plan.py: one env.step(action) per
            loop turn, np.random step sizes, reset on done), host-timed: a doorbell and an acknowledgement of the env's resident wavefront per
            step.  For scale only: the reference's own class does 110 k steps/s on one core of the BUILD container (another CPU, BASELINE.md
            section 2); this box's own host is timed in cpu_baseline.python_loop_n1024.  The drop-in classes are the PARITY surface,
            BatchedDMPEnv is the THROUGHPUT surface."""
            import numpy as np

            from snac_amd.envs import deep_mobile_printing_2d1r_dynamic

            e = deep_mobile_printing_2d1r_dynamic("data_2d_dynamic_dense_envplan_500_train.pkl")
            np.random.seed(1)
            e.reset()
            acts = np.random.RandomState(0).randint(0, 5, steps)
            for i in range(200):
                if e.step(int(acts[i]))[2]:
                    e.reset()
            def loop(k):
                for i in range(k):
                    if e.step(int(acts[i]))[2]:
                        e.reset()

            per, tm = host_timed(loop, steps)
            rate = 1.0 / per
            mbs = e._env.mailbox_stats()
            res[name] = {"facade_steps_per_s": rate, "us_per_step": 1e6 / rate, "reference_steps_per_s_one_core": 110300.0, "timing": tm,
                         "path": "mailbox (resident wavefront, snac_mailbox_step)" if getattr(e, "_mbox", False) else "launch (snac_step_scalar + wait)",
                         "mailbox": mbs,
                         "note": "deep_mobile_printing_2d1r(data_path).step(a) of snac_amd.envs, one env, host-timed; reference figure: BASELINE.md section 2 "
                                 "(measured in the build container, other CPU); mailbox.last_step_us: the resident wave's own account of its last step "
                                 "(the rest of a step is the bus: 2.5 us for a bare doorbell echo, tools/bar_probe.hip)"}

        def vector_cfg(name, nn, steps=1500):
            """VectorizedEnvWrapper.step(actions) (multiprocess.py:15-32 on the HIP path: numpy in, numpy out), host-timed, with the
            reference's default --num_envs 3, 64 envs (one resident wave) and 256 (four: the largest batch of the resident path; round 5: one launch + one wait per
            vector step).  The reference wrapper steps its N envs one after the other at ~9 us each."""
            import numpy as np

            from snac_amd.vector import VectorizedEnvWrapper

            table = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "snac_amd", "data", "plans.npz"))["2d_dense_train"]
            w = VectorizedEnvWrapper((2, True, table), num_envs=nn)
            np.random.seed(1)
            w.reset()
            acts = np.random.RandomState(0).randint(0, 5, (steps + 100, nn))
            for i in range(100):
                w.step(acts[i])
            def loop(k):
                for i in range(k):
                    w.step(acts[100 + i])

            per, tm = host_timed(loop, steps)
            dt_ = per * steps
            res[name] = {"vector_steps_per_s": steps / dt_, "env_steps_per_s": steps * nn / dt_, "us_per_vector_step": 1e6 * dt_ / steps, "num_envs": nn, "timing": tm,
                         "path": "mailbox (resident wavefront, an env per lane: snac_mailbox_step_n)" if w._mrows is not None else "launch (snac_step + wait)",
                         "reference_env_steps_per_s_one_core": 110300.0,
                         "note": "snac_amd.vector.VectorizedEnvWrapper.step(actions): host numpy in and out, no auto-reset (stepped past done like the "
                                 "reference's loop); the reference's wrapper does its N env.step() calls in turn on one core"}

        def step_cfg(name, kind, nn, reps, layout=None):
            e = BatchedDMPEnv(kind, True, nn, device=dev, seed=1, **({"layout": layout} if layout else {}))
            e.reset()
            out = (view((nn, e.obs_dim), torch.float64), torch.empty(nn, dtype=torch.float32, device=dev), torch.empty(nn, dtype=torch.uint8, device=dev))
            ms = timed(lambda: e.step(auto_reset=True, out=out), reps)
            algb = CONTRACT_BYTES[(kind, "f64")] + (e.obs_dim - (7 if kind == 1 else 51)) * 8    # a layout variant's tail is written too
            gbs = algb * nn / (ms * 1e-3) / 1e9
            res[name] = {"kernel": last_kernel(), "us_per_tick": ms * 1e3, "values_per_row": e.obs_dim, "env_steps_per_s": nn / (ms * 1e-3), "alg_bytes_per_env_step": algb, "achieved_GBs": gbs,
                         "frac": gbs / HBM_PEAK_GBS, "launches": GROUPS * reps, "timing": spread[-1],
                         "note": "device time per tick, launches enqueued back to back" + ("" if nn >= 262144 else "; at this batch size a tick is the launch gap (about 2.5 us) plus its bytes at the rate of a plain copy (2D: 38 MB = 6.0 us, 3D: 51 MB = 8.1 us)")}

        def reset_cfg(name, kind, nn, reps):
            """snac_reset of every env (row a1 of the path: reset()): header, episode counter, the record zeroed, the reset observation written --
            bytes per env: row + 16 + 4 + 4 (episode counter in and out) + the record (64 / 80 / 800)."""
            e = BatchedDMPEnv(kind, True, nn, device=dev, seed=1)
            e.reset()
            ms = timed(lambda: e.reset(), reps)
            algb = e.obs_dim * 8 + 24 + {1: 64, 2: 80, 3: 800}[kind]
            gbs = algb * nn / (ms * 1e-3) / 1e9
            res[name] = {"kernel": last_kernel(), "us_per_call": ms * 1e3, "envs_per_s": nn / (ms * 1e-3), "alg_bytes_per_env": algb, "achieved_GBs": gbs,
                         "frac": gbs / HBM_PEAK_GBS, "launches": GROUPS * reps, "timing": spread[-1], "note": "reset() of the whole batch, rows allocated by the call"}

        def default_alloc_cfg(name, reps):
            """The headline pass as a user calls it -- env.rollout(T) with NO out= --: the observation tensor comes from the cache of
            measured trajectory blocks (snac_amd/trajmem.py cached_empty: built on the first call, recycled afterwards)."""
            from snac_amd import trajmem

            e = BatchedDMPEnv(2, True, n, device=dev, seed=1)
            e.reset()
            c0 = trajmem.cache_stats()
            t0 = time.perf_counter()
            o = e.rollout(T)[0]
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            lay = trajmem.layout_of(o)
            del o
            ms = timed(lambda: e.rollout(T), reps)
            c1 = trajmem.cache_stats()
            algb = 413 + 2.0 * STATE_BYTES[2] / T
            gbs = algb * n * T / (ms * 1e-3) / 1e9
            res[name] = {"kernel": last_kernel(), "kernel_ms": ms, "vs_headline": ms / kern_ms, "first_call_s": first, "block_layout": lay,
                         "blocks_built": c1["built"] - c0["built"], "blocks_reused": c1["reused"] - c0["reused"],
                         "env_steps_per_s": n * T / (ms * 1e-3), "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS, "launches": GROUPS * reps,
                         "timing": spread[-1], "note": "BatchedDMPEnv.rollout(T) without out=: rows, reward and done allocated by the call itself"}
            trajmem.cache_trim()

        default_alloc_cfg("headline_default_alloc", 12)
        rollout_cfg("c2_1d_static_n4096_T750", 1, False, 4096, False, 40,
                    "time-parallel kernel k_rollout1dt (one wave per env, lane = tick, the rows of 8 envs through an LDS tile as 448-byte runs: "
                    "0.29 -> 0.045 ms); the pass writes only 187 MB and is bound by its vector instructions (5.4 per env-step by the counters), not by the HBM rate")
        # the same kind at the headline's batch size: from 45 056 envs there is a 64-env wave for every SIMD and the rollout is lane-per-env
        # (k_rollout1dl, round 6: 0.72 ms on the time-parallel / tile kernels before)
        rollout_cfg("rollout_1d_dynamic_n65536_T750", 1, True, 65536, False, 24)
        rollout_cfg("c5_3d_dynamic_n16384_T1000", 3, True, 16384, False, 24)
        rollout_cfg("headline_f32_obs", 2, True, 65536, True, 24)
        # the headline on a table of 2000 GENERATED plans (snac_make_plans: what generate_plans() is for) and in the observation layout
        # of the script/PPO env copies (451 values per row: window, counters, the 400 plan cells)
        rollout_cfg("headline_2000_generated_plans", 2, True, 65536, False, 24, plans=2000)
        rollout_cfg("headline_ppo_layout_451_values", 2, True, 65536, False, 6, layout="ppo", TT=60)
        # the batch sizes the on-policy scripts use: the same 451-value rows and the canonical ones on the time-parallel kernel
        rollout_cfg("ppo_layout_n4096_T600", 2, True, 4096, False, 10, layout="ppo")
        rollout_cfg("ppo_layout_n1024_T600", 2, True, 1024, False, 20, layout="ppo")
        rollout_cfg("small_batch_n4096_T600", 2, True, 4096, False, 30)
        # the middle batches (at most one lane-per-env wave per SIMD): the block kernel k_rollout2db, blocks of 128 envs with two stepper waves
        rollout_cfg("mid_batch_n20480_T600", 2, True, 20480, False, 20)
        rollout_cfg("ppo_layout_1d_n1024_T750", 1, True, 1024, False, 30, layout="ppo")
        rollout_cfg("ppo_layout_3d_n16384_T200", 3, True, 16384, False, 6, layout="ppo", TT=200)
        # 3D rows that carry their own record (reward, done, position, counters, plan row: 59 values), what a replay writer of the 3D classes stores
        rollout_cfg("record_rows_3d_n16384_T1000", 3, True, 16384, False, 12, tail=("record",))
        rollout_cfg("small_batch_n1024_T600", 2, True, 1024, False, 30)
        for kind in (2, 3):
            for nn in (65536, 524288):
                step_cfg("step_%dd_dynamic_n%d" % (kind, nn), kind, nn, 200)
        # 1D: k_step1d since the end of round 6 (the tile kernel before: 39 us).  By SURVEY 8d's 88 bytes per env-step; what the tick moves is
        # 163 -- the record is 64 bytes whatever the step reads of it, the header goes in and out
        step_cfg("step_1d_dynamic_n524288", 1, 524288, 200)
        step_cfg("step_2d_ppo_layout_n65536", 2, 65536, 100, layout="ppo")   # what a trainer that steps 65 536 envs per tick reads: 451-value rows
        step_cfg("step_3d_ppo_layout_n65536", 3, 65536, 100, layout="ppo")
        reset_cfg("reset_2d_n524288", 2, 524288, 50)                # k_reset since the end of round 6 (the tile kernel k_aux before: 81 / 412 us)
        reset_cfg("reset_3d_n524288", 3, 524288, 30)
        for kind in (1, 2, 3):                                      # (1D: k_edges1d since the end of round 6; the tile kernel before: 54 us)
            edges_cfg("transition_%dd_524288_edges" % kind, kind, 524288, 20)
        edges_cfg("transition_2d_nodes_524288_edges", 2, 524288, 20, nodes=True)
        gather_cfg("replay_gather_65536", 65536, 64, 65536, 20)
        facade_cfg("facade_2d_dynamic_one_env")
        vector_cfg("vector_wrapper_3_envs", 3)
        vector_cfg("vector_wrapper_64_envs", 64)
        vector_cfg("vector_wrapper_256_envs", 256)
        return res

    extras = None
    if world == 1 and rank == 0 and os.environ.get("SNAC_BENCH_EXTRAS", "1") != "0" and (args.kind, dynamic, n, args.obs_f32) == (2, True, 65536, False) and T == 600:
        try:
            extras = extra_configs()
        except Exception as e:                                   # informational only: never take the headline line down with it
            sys.stderr.write("bench.py: extra configurations failed (%r)\n" % (e,))
            extras = {"error": repr(e)}

    try:
        tiled = tiled_extra() if (world == 1 and args.kind == 2 and n % 64 == 0 and not args.static and os.environ.get("SNAC_BENCH_TILED", "1") != "0") else None
    except Exception as e:                                       # informational only: never take the headline line down with it
        sys.stderr.write("bench.py: tile-major extra measurement failed (%r)\n" % (e,))
        tiled = {"error": repr(e)}

    if rank == 0:
        total_steps = world * n * T * args.steps
        dkey = "f32" if args.obs_f32 else "f64"
        # algorithmic bytes of the FUSED launch: what any implementation of T fused steps has to move through HBM -- every
        # observation row + reward + done written, the state loaded and stored once (2D: 413 + 248 / 600 = 413.4 B per env-step)
        alg = WRITTEN_BYTES[(args.kind, dkey)] + 2.0 * STATE_BYTES[args.kind] / T
        achieved = alg * n * T / (kern_ms * 1e-3) / 1e9
        contract = CONTRACT_BYTES[(args.kind, dkey)]
        achieved_contract = contract * n * T / (kern_ms * 1e-3) / 1e9
        s = stats.tolist()
        # measured HBM bytes per launch (rocprofv3 PMC passes of this same command, tools/profile.sh ->
        # profiles/traffic.json), turned into GB/s with the live launch duration; null for other workloads
        traffic, traffic_source = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        headline = (args.kind, dynamic, n, T, dkey) == (2, True, 65536, 600, "f64")
        if os.path.exists(tfile) and headline:
            with open(tfile) as fh:
                tj = json.load(fh)
            now = _lib.kernel_source_sha16()
            if tj.get("source_sha16") == now:                    # the counters were taken with THIS kernel source
                traffic = tj["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9
                traffic_source = "profiles/traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command, kernel source (snac_dev.h + k_roll2d.hip) sha256 %s), bytes per launch / the live kernel time" % now
            else:
                traffic_source = "none: profiles/traffic.json belongs to kernel source %s, this library was built from %s" % (tj.get("source_sha16"), now)
        written = WRITTEN_BYTES[(args.kind, dkey)] * n * T / (kern_ms * 1e-3) / 1e9
        what = "%dD %s dense" % (args.kind, "dynamic" if dynamic else "static")
        out = {
            "metric": HEADLINE if headline else "env-steps/sec at N=%d envs (%s, %s obs); bit-exact vs CPU" % (n, what, dkey),
            "value": total_steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dkey,
            "data": "synthetic",
            "config": {"workload": "%s, %d envs/GPU, %d vector steps/pass, uniform random actions (counter RNG), "
                                   "auto-reset, %s obs of every step written" % (what, n, T, dkey),
                       "envs_per_gpu": n, "vector_steps_per_pass": T, "env_steps_per_pass": n * T * world,
                       "parallelism": "env-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "frac_traffic": (traffic / HBM_PEAK_GBS) if traffic else None,
                         "peak_measured_write": wpeak,            # hipMemsetAsync, the faster of the two kinds of memory:
                         "peak_measured_write_by_memory": {"malloc": wpeak_malloc, "vmm": wpeak_vmm},
                         "written": written,                      # output bytes of the launch / its duration, GB/s
                         "frac_of_measured_write": ((traffic or written) / wpeak) if wpeak else None,
                         "kernel": kernel_name, "kernel_ms": kern_ms, "alg_bytes_per_env_step": alg,
                         "contract": {"alg_bytes_per_env_step": contract, "achieved": achieved_contract,
                                      "frac": achieved_contract / HBM_PEAK_GBS,
                                      "note": "SURVEY.md 8d's un-fused per-step figure (state, window and scalars through HBM every step); "
                                              "rounds 1-2 reported this as achieved / frac (0.83-0.86 then)"}},
            "backend": backend if use_dist else None,
            "rccl_ranks": ranks_seen if (use_dist and backend == "nccl") else None,
            "collective_check": collective_check,                 # SNAC_BENCH_FORCE_DIST=1 at one rank: reduced sums == local sums
            "rccl_async_exchanges": n_async if use_dist else None,   # per-pass all_reduce(async_op=True) calls that completed
            "retimed": retimed,                                   # the first timed region showed a stall of the box and the region was timed once more
            "timed_regions": timed_regions,                       # every timed region of this run (one, or two when retimed), rank 0's kernels
            "ranks": ranks_seen,
            "kernel_ms_per_rank": per_rank,
            "ranks_devices": ranks_devices,                       # every rank's cuda:<local> PCI bus id: N ranks on N distinct GPUs
            "kernel_ms_per_step": [round(x, 4) for x in per_step_ms],   # rank 0's launches, in order
            "preroll_passes": preroll_passes,                     # untimed, before the W warm-up passes (clock ramp)
            "trajectory_check": traj_ok and all_ok,               # obs[T - 1] == observe() after the timed passes (every rank), and:
            "trajectory_full_pass_check": full_check,             # one more pass == the same pass by the tile kernel, all T x N rows
            "parity_vs_oracle": parity_all,                       # reset + the first 10 ticks of a fresh pass into the measured block == the CPU oracle, byte for byte (every rank)
            "per_rank": per_rank_report,                          # every rank's kernel time, trajectory block and checks
            "ranks_on_distinct_devices": devices_ok,
            "placement": placement_report,                        # rank 0's choice among SNAC_BENCH_PLACE candidate tensors
            "tiled_layout": tiled,                                # rank 0, informational: the build's own trajectory layout
            "extra": {"configs": extras},                         # rank 0, one GPU: the other configs / dtypes / step(), driver-timed
            "episodic": {"episodes": s[0], "mean_return": (s[1] / s[0]) if s[0] else None,
                         "mean_iou": (s[2] / 2.0 ** 40 / s[0]) if s[0] else None},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.kind, dynamic, n, T, 1)
        # The driver parses the LAST stdout line and keeps only a tail of stdout: the line stays a few KB (round 5's 22 KB line was
        # cut and could not be parsed).  Everything else -- extras with their timing spreads, placement, per-rank accounts, timed
        # regions -- goes to the side file, whose path the line names.
        extra_file = os.environ.get("SNAC_BENCH_EXTRA_FILE") or os.path.join(ROOT, "bench_extra.json")
        try:
            with open(extra_file, "w") as fh:
                json.dump(out, fh, indent=1)
                fh.write("\n")
        except OSError as e:
            sys.stderr.write("bench.py: could not write %s (%r)\n" % (extra_file, e))
            extra_file = None
        print(json.dumps(compact_line(out, extra_file), separators=(",", ":"), allow_nan=False))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if not (traj_ok and all_ok and devices_ok) or parity_all is False or collective_check is False:
        sys.exit(4)


if __name__ == "__main__":
    main()
