"""Shared test helpers: golden loading, plan tables, oracle access."""
import json
import os
import sys

import numpy as np

TESTS = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(TESTS)
GOLDEN = os.path.join(TESTS, "golden")
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

DIMS = {1: dict(A=3, W=5, D=7, T=(750, 750)), 2: dict(A=5, W=49, D=51, T=(600, 600)), 3: dict(A=8, W=49, D=51, T=(1300, 1000))}

_cache = {}


def plans_npz():
    if "plans" not in _cache:
        _cache["plans"] = np.load(os.path.join(ROOT, "snac_amd", "data", "plans.npz"))
    return _cache["plans"]


def static_plans_npz():
    if "static" not in _cache:
        _cache["static"] = np.load(os.path.join(GOLDEN, "static_plans.npz"))
    return _cache["static"]


def digests():
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        return json.load(f)


def plan_table(dim, dyn, tag):
    """Full plan table [P, cells] int32 as the reference holds it (bordered 26x26 for 2D/3D)."""
    if dyn:
        a = plans_npz()["%dd_%s" % (dim, tag)]
        return np.ascontiguousarray(a.reshape(len(a), -1), np.int32)
    pc = int(tag[1:])
    a = static_plans_npz()["%dd_p%d" % (dim, pc)]
    return np.ascontiguousarray(a.reshape(1, -1), np.int32)


def traj_file(dim, dyn):
    return np.load(os.path.join(GOLDEN, "traj_%dd_%s.npz" % (dim, "dynamic" if dyn else "static")))


def golden_cases():
    """Yield (dim, dyn, case_name, dict-of-arrays)."""
    for dim in (1, 2, 3):
        for dyn in (False, True):
            z = traj_file(dim, dyn)
            for name in z["cases"].tolist():
                rec = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}
                yield dim, dyn, name, rec


def case_ids():
    out = []
    for dim in (1, 2, 3):
        for dyn in (False, True):
            z = traj_file(dim, dyn)
            out += [(dim, dyn, n) for n in z["cases"].tolist()]
    return out


def load_case(dim, dyn, name):
    z = traj_file(dim, dyn)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def obs_from_golden(rec, t, dim):
    """float64 primary observation of golden step t."""
    return np.concatenate([rec["win"][t].astype(np.float64), rec["sc"][t]])


def oracle():
    from oracle import snac_oracle

    return snac_oracle


# ---- figures the parity files measure in passing (build times, rates): noted here, judged in tests/test_zz_gpu_perf.py, which sorts last
# and REPORTS -- no test before it can fail on a clock
_PERF_NOTES = {}


def same_bytes(a, b):
    """a.tobytes() == b.tobytes() without the two copies (the large batches' rows are hundreds of megabytes)."""
    a, b = np.ascontiguousarray(a).reshape(-1), np.ascontiguousarray(b).reshape(-1)
    if a.nbytes != b.nbytes:
        return False
    word = np.uint64 if a.nbytes % 8 == 0 else np.uint8
    return bool(np.array_equal(a.view(word), b.view(word)))


def perf_note(name, value):
    _PERF_NOTES[name] = value


def perf_notes():
    return dict(_PERF_NOTES)
