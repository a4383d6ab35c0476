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


LINE_LIMIT = 6000                                                 # bytes: the driver keeps a tail of stdout and parses its last line


def _strict(text):
    def bad(c):
        raise ValueError("non-standard JSON constant %r" % c)
    return json.loads(text, parse_constant=bad)


def _parse(stdout, extra_path):
    """The bench contract on stdout: ONE JSON object, on the LAST line, strict JSON, a few KB, carrying the roofline fraction (and the
    CPU baseline when the run timed one).  Returns the full account of the side file the line names, after checking that the line's
    figures are that account's."""
    rows = [ln for ln in stdout.splitlines() if ln.strip()]
    lines = [ln for ln in rows if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    assert rows[-1] is lines[0], "the JSON object must be the last stdout line"
    assert len(lines[0].encode()) < LINE_LIMIT, len(lines[0])
    line = _strict(lines[0])
    assert KEYS <= set(line)
    assert isinstance(line["roofline"]["frac"], float) and 0 < line["roofline"]["frac"] < 1.0
    assert line["extra_file"] == os.path.basename(extra_path)
    with open(extra_path) as fh:
        full = _strict(fh.read())
    for k in ("metric", "n_gpus", "steps", "warmup", "scaling", "dtype", "config", "episodic", "ranks", "backend", "retimed",
              "parity_vs_oracle", "trajectory_check"):
        assert line[k] == full[k] or k == "episodic", k
    for k in ("value", "ms_per_step"):
        assert abs(line[k] - full[k]) <= 1e-5 * abs(full[k]), k
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert ("cpu_baseline" in line) == ("cpu_baseline" in full)
    if "cpu_baseline" in line:
        cb = line["cpu_baseline"]
        assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1 and cb["single_thread"]["value"] > 0
    full["_line"] = line
    return full


def _bench(cmd, env, check=True):
    import tempfile

    e = dict(env)
    fd, path = tempfile.mkstemp(prefix="snac_bench_", suffix=".json")
    os.close(fd)
    os.unlink(path)
    e["SNAC_BENCH_EXTRA_FILE"] = path
    try:
        out = subprocess.run(cmd, cwd=helpers.ROOT, env=e, capture_output=True, text=True, timeout=900)
        if not check:
            return out, None
        assert out.returncode == 0, out.stderr[-2000:]
        return out, _parse(out.stdout, path)
    finally:
        if os.path.exists(path):
            os.unlink(path)


def _run(cmd, env=None):
    e = dict(os.environ)
    e["SNAC_BENCH_RETIME"] = "0"                                  # (a retimed region runs K more passes: the episodic sums compared below would differ)
    e.update(env or {})
    return _bench(cmd, e)[1]


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
    out, two = _bench(cmd + ["--gpus", "2", "--envs", "4096"], env)
    one = _run(cmd + ["--envs", "8192"])
    assert two["n_gpus"] == 2 and two["ranks"] == 2 and two["backend"] == "gloo"
    assert two["episodic"] == one["episodic"]
    assert len(two["ranks_devices"]) == 2 and all(d and d.startswith("cuda:") for d in two["ranks_devices"])
    assert two["extra"]["configs"] is None and two["tiled_layout"] is None          # N > 1: the headline only
    # four ranks (the GPU box admits six processes on its card: this test process, its launcher -- which never touches the GPU --
    # and four ranks; BASELINE config 4's eight ranks on one card would trip that guard, the 8-way split itself runs on CPU in
    # tests/test_dist_gloo.py, the launcher's own eight-process form in tests/test_bench_launcher.py): 4 x 2048 envs == 1 x 8192
    out, four = _bench(cmd + ["--gpus", "4", "--envs", "2048"], env)
    assert four["n_gpus"] == 4 and four["ranks"] == 4 and len(four["kernel_ms_per_rank"]) == 4 and len(four["ranks_devices"]) == 4
    assert four["episodic"] == one["episodic"] and four["config"]["env_steps_per_pass"] == 4 * 2048 * 120
    # RCCL needs one GPU per rank: two RCCL ranks on a one-GPU box must fail loudly, not report a 1-GPU number
    if __import__("torch").cuda.device_count() == 1:
        env["SNAC_BENCH_BACKEND"] = "nccl"
        bad, _ = _bench(cmd + ["--gpus", "2", "--envs", "4096"], env, check=False)
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
    out, forced = _bench(cmd, env)
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
