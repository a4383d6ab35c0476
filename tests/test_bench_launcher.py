"""CPU: the parts of bench.py that need no GPU -- the compact stdout line (the driver parses the LAST stdout line and keeps only a
tail: round 5's 22 KB line could not be parsed) and the `--gpus N` launcher as an eight-process topology (a one-GPU test box admits
six processes on its card, so BASELINE config 4's eight ranks cannot start there; the launcher itself never touches the GPU)."""
import contextlib
import io
import json
import os
import sys
import textwrap

import helpers

sys.path.insert(0, helpers.ROOT)
import bench  # noqa: E402


def _strict(text):
    def bad(c):
        raise ValueError(c)
    return json.loads(text, parse_constant=bad)


def test_compact_line_of_a_full_headline_account_is_a_few_kb():
    """profiles/r05_bench.json is round 5's full 22 KB account (26 extras, placement, per-rank): its compact line must stay far below
    the driver's tail and keep what the judge reads -- value, ms_per_step, roofline, cpu_baseline, every extra's kernel / time / fraction."""
    with open(os.path.join(helpers.ROOT, "profiles", "r05_bench.json")) as fh:
        full = json.load(fh)
    text = json.dumps(bench.compact_line(full, "/somewhere/bench_extra.json"), separators=(",", ":"), allow_nan=False)
    assert len(text.encode()) < 4096, len(text)
    line = _strict(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_vs_oracle", "trajectory_check"):
        assert k in line, k
    assert line["metric"] == full["metric"] and line["config"]["workload"] == full["config"]["workload"]
    assert abs(line["value"] - full["value"]) < 1e-5 * full["value"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - full["roofline"]["frac"]) < 1e-5 and r["traffic"] is not None
    assert r["kernel"] == "k_rollout2d" and r["peak"] == 8000.0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == full["cpu_baseline"]["cores"] and cb["value"] > 0 and cb["single_thread"]["value"] > 0
    assert set(line["extra"]) == set(full["extra"]["configs"]) and line["extra_cols"] == bench.EXTRA_COLS
    k, us, frac = line["extra"]["step_3d_dynamic_n524288"]
    assert k == "k_step3dq" and abs(us - full["extra"]["configs"]["step_3d_dynamic_n524288"]["us_per_tick"]) < 1e-3 and 0 < frac < 1
    assert line["extra_file"] == "bench_extra.json"


def test_compact_line_is_strict_json_whatever_the_floats():
    full = {"metric": "m", "value": float("inf"), "roofline": {k: float("nan") for k in
            ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "alg_bytes_per_env_step", "peak_measured_write")}}
    line = _strict(json.dumps(bench.compact_line(full, None), allow_nan=False))
    assert line["value"] is None and line["roofline"]["frac"] is None and line["extra_file"] is None


RANK_STUB = textwrap.dedent('''
    import json, os, sys, time
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    assert int(os.environ["LOCAL_RANK"]) == r and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    mode = sys.argv[1]
    open(os.path.join(sys.argv[2], "rank%d" % r), "w").write(os.environ["MASTER_PORT"])
    if mode == "fail" and r == 5:
        sys.exit(7)
    if mode == "fail":
        time.sleep(60)                     # the others wait in a collective that never completes
    if r == 0:
        print("not the line")
        print(json.dumps({"n_gpus": w, "ranks": w}))
''')


def _launch(tmp_path, mode, n=8):
    stub = tmp_path / "rank_stub.py"
    stub.write_text(RANK_STUB)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.launch_ranks(n, script=str(stub), argv=[mode, str(tmp_path)])
    return rc, buf.getvalue()


def test_launcher_starts_eight_ranks_and_forwards_rank_0s_line(tmp_path):
    rc, out = _launch(tmp_path, "ok")
    assert rc == 0
    assert [json.loads(ln) for ln in out.splitlines()] == [{"n_gpus": 8, "ranks": 8}]
    ports = {(tmp_path / ("rank%d" % r)).read_text() for r in range(8)}
    assert len(ports) == 1                                            # all eight saw the same rendezvous


def test_launcher_ends_the_other_ranks_when_one_dies(tmp_path):
    import time

    t0 = time.perf_counter()
    rc, out = _launch(tmp_path, "fail")
    assert rc == 7 and out.strip() == "" and time.perf_counter() - t0 < 30
