"""GPU: the bench.py contract (one JSON line, required keys) and its N > 1 path: two ranks under
torch.distributed.run, sharing the one GPU of the test box (collectives via gloo for that reason; production uses RCCL),
must report the same global episodic sums as one process holding all the envs."""
import json
import os
import socket
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _run(cmd, env=None):
    e = dict(os.environ)
    e["SNAC_BENCH_RETIME"] = "0"                                  # (a retimed region runs K more passes: the episodic sums compared below would differ)
    e.update(env or {})
    out = subprocess.run(cmd, cwd=helpers.ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_line_and_two_rank_path():
    one = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--envs", "8192", "--T", "120"])
    assert KEYS <= set(one) and "cpu_baseline" in one
    assert one["n_gpus"] == 1 and one["unit"] == "env-steps/s" and one["dtype"] == "f64" and one["scaling"] == "weak"
    assert one["vs_baseline"] is None and one["higher_is_better"] is True and "workload" in one["config"]
    r = one["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["unit"] == "GB/s"
    assert one["cpu_baseline"]["kind"] == "port" and one["cpu_baseline"]["cores"] >= 1
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--envs", "4096",
                "--T", "120", "--no-cpu"], env={"SNAC_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["env_steps_per_pass"] == 2 * 4096 * 120
    # 2 ranks x 4096 envs (ids 0..8191) == 1 rank x 8192 envs: same global episodic sums after the same 3 passes
    assert two["episodic"] == one["episodic"]
    assert two["ranks"] == 2 and len(two["kernel_ms_per_rank"]) == 2 and min(two["kernel_ms_per_rank"]) > 0
    cb = one["cpu_baseline"]
    assert cb["single_thread"]["cores"] == 1 and cb["single_thread"]["value"] > 0 and cb["python_loop_n1024"]["value"] > 0
    assert cb["cpu_model"] and "upper bound" in cb["python_loop_n1024"]["sample"].lower()
    assert one["extra"]["configs"] is None                            # the extras belong to the headline shape (N = 65 536, T = 600)
    assert one["ranks_devices"] and one["ranks_devices"][0].startswith("cuda:0 pci ")
    assert r["peak_measured_write"] is None or 1000 < r["peak_measured_write"] < 8000


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no torch.distributed environment: the process becomes a launcher, starts two ranks
    (sharing this box's one GPU, hence gloo for the three int64 sums) and forwards rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SNAC_BENCH_BACKEND"] = "gloo"
    env["SNAC_BENCH_RETIME"] = "0"
    cmd = [sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--T", "120", "--no-cpu"]
    out = subprocess.run(cmd + ["--gpus", "2", "--envs", "4096"], cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    two = json.loads(lines[0])
    one = _run(cmd + ["--envs", "8192"])
    assert two["n_gpus"] == 2 and two["ranks"] == 2 and two["backend"] == "gloo"
    assert two["episodic"] == one["episodic"]
    assert len(two["ranks_devices"]) == 2 and all(d and d.startswith("cuda:") for d in two["ranks_devices"])
    assert two["extra"]["configs"] is None and two["tiled_layout"] is None          # N > 1: the headline only
    # four ranks (the GPU box admits six processes on its card: this test process, its launcher -- which never touches the GPU --
    # and four ranks; BASELINE config 4's eight ranks on one card would trip that guard, the 8-way split itself runs on CPU in
    # tests/test_dist_gloo.py): 4 x 2048 envs == 1 x 8192, inside a wall-time bound that a per-rank start-up of seconds would break
    import time

    t0 = time.perf_counter()
    out = subprocess.run(cmd + ["--gpus", "4", "--envs", "2048"], cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    assert out.returncode == 0, out.stderr[-2000:]
    four = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert four["n_gpus"] == 4 and four["ranks"] == 4 and len(four["kernel_ms_per_rank"]) == 4 and len(four["ranks_devices"]) == 4
    assert four["episodic"] == one["episodic"] and four["config"]["env_steps_per_pass"] == 4 * 2048 * 120
    assert wall < 240, wall
    # RCCL needs one GPU per rank: two RCCL ranks on a one-GPU box must fail loudly, not report a 1-GPU number
    if __import__("torch").cuda.device_count() == 1:
        env["SNAC_BENCH_BACKEND"] = "nccl"
        bad = subprocess.run(cmd + ["--gpus", "2", "--envs", "4096"], cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_rccl_int64_all_reduce_single_rank():
    """The collective the N > 1 path uses (int64 SUM + barrier) on the RCCL backend, world size 1."""
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.tensor([3, -7, 1 << 45], dtype=torch.int64, device="cuda:0")
        dist.all_reduce(t)
        dist.barrier()
        m = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        assert t.tolist() == [3, -7, 1 << 45] and m.item() == 1.5
    finally:
        dist.destroy_process_group()


def test_overlapped_all_reduce_form_on_rccl():
    """bench.py enqueues its per-pass exchange with async_op=True on RCCL and waits for all of them before the clock stops.
    One GPU cannot host two RCCL ranks, so the call form is exercised on a 1-rank RCCL group: int64 [3] tensor, several
    in flight, results intact."""
    import subprocess
    import sys

    code = r'''
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
pending = []
for i in range(5):
    s = torch.tensor([i, 2 * i, 3 * i], dtype=torch.int64, device="cuda")
    pending.append((dist.all_reduce(s, op=dist.ReduceOp.SUM, async_op=True), s))
for w, s in pending:
    w.wait()
torch.cuda.synchronize()
assert [s.tolist() for _, s in pending] == [[i, 2 * i, 3 * i] for i in range(5)]
dist.destroy_process_group()
print("OK")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-300:], r.stderr[-800:])


def test_bench_takes_its_rccl_path_with_one_rank():
    """SNAC_BENCH_FORCE_DIST=1: bench.py initialises the RCCL ("nccl") process group before any GPU call although there is one
    rank, and every pass performs the N > 1 exchange exactly as an 8-rank run does -- all_reduce(async_op=True) of the three int64
    sums on RCCL's stream behind the rollout, work.wait() for all of them before the clock stops, the barriers either side, the
    MAX over ranks of the wall time.  What came back must be the rank's own sums, and the line must say RCCL saw one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SNAC_BENCH_FORCE_DIST="1", SNAC_BENCH_BACKEND="nccl", SNAC_BENCH_RETIME="0")
    cmd = [sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--envs", "8192", "--T", "120", "--no-cpu"]
    out = subprocess.run(cmd, cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    forced = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert forced["backend"] == "nccl" and forced["rccl_ranks"] == 1 and forced["ranks"] == 1 and forced["n_gpus"] == 1
    assert forced["collective_check"] is True and forced["rccl_async_exchanges"] == 4
    assert forced["ranks_on_distinct_devices"] is True and forced["parity_vs_oracle"] is True and forced["trajectory_check"] is True
    env.pop("SNAC_BENCH_FORCE_DIST")
    plain = _run(cmd, env={"SNAC_BENCH_FORCE_DIST": "0"})
    assert plain["backend"] is None and plain["rccl_ranks"] is None and plain["collective_check"] is None
    assert plain["episodic"] == forced["episodic"]                    # the same passes, with and without the group


def test_a_stalled_timed_region_is_timed_once_more():
    """The boxes of the pool stall for tens of milliseconds now and then; a headline region that shows it (slowest pass > 2 x the median
    pass) is timed once more and both regions are reported.  Forced here with a pass count of 3 and a sleep-free trick: the first
    region of a cold process carries the clock ramp when the pre-roll is switched off and the batch is tiny."""
    one = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "0", "--envs", "8192", "--T", "120", "--no-cpu"],
               env={"SNAC_BENCH_RETIME": "1", "SNAC_BENCH_PREROLL_MS": "0"})
    assert isinstance(one["retimed"], bool) and 1 <= len(one["timed_regions"]) <= 2
    assert one["retimed"] == (len(one["timed_regions"]) == 2)
    assert one["timed_regions"][0]["stalled"] == one["retimed"]
    last = one["timed_regions"][-1]
    assert abs(last["ms_per_step"] - one["ms_per_step"]) < 1e-9 and len(last["kernel_ms_per_step"]) == 3
