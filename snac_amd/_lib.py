"""ctypes binding of libsnac_hip.so (include/snac_hip.h).  There is no CPU fallback: if the HIP
library is missing or a GPU is not available the product raises."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNAC_HIP_LIB") or os.path.join(HERE, "libsnac_hip.so")  # override: A/B builds
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

SNAC_OK = 0
ABI_VERSION = 12
ENV_1D, ENV_2D, ENV_3D = 1, 2, 3
OBS_F64, OBS_F32 = 0, 1
OBS_NONE, OBS_ALL, OBS_LAST, OBS_TILED = 0, 1, 2, 3
FLAG_NEED_RESET = 1
RULE_BRICK_GT, RULE_TIME_GT = 1, 2
SCALARS_DEFAULT, SCALARS_RAW, SCALARS_NORM = 0, 1, 2
TAIL_POSITION, TAIL_PLAN, TAIL_RECORD = 1, 2, 4

EXPORTS = ("snac_version", "snac_last_error", "snac_env_sizes", "snac_obs_dim", "snac_reset", "snac_reset_scalar", "snac_step",
           "snac_step_scalar", "snac_rollout",
           "snac_rollout_rec", "snac_replay_gather", "snac_make_plans", "snac_observe", "snac_iou", "snac_export_grid", "snac_transition",
           "snac_import_state", "snac_obs_equal", "snac_discounted_return", "snac_plans_from_grids", "snac_mailbox_create", "snac_mailbox_row", "snac_mailbox_touch", "snac_mailbox_step", "snac_mailbox_step_n", "snac_mailbox_reward", "snac_mailbox_done",
           "snac_mailbox_quit", "snac_mailbox_settle", "snac_mailbox_destroy", "snac_mailbox_stats", "snac_stream_sync", "snac_rollout_tiled", "snac_replay_gather_tiled", "snac_traj_alloc",
           "snac_traj_alloc_ex", "snac_traj_free", "snac_traj_layout", "snac_traj_describe", "snac_traj_reserved_bytes", "snac_last_kernel", "snac_tuning",
           "snac_nodes2d_pack", "snac_nodes2d_unpack", "snac_transition_nodes2d")


class Sizes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "obs_dim", "num_actions", "total_step", "half_window", "env_height", "env_width", "plan_height", "plan_width",
        "grid_elems", "grid_elem_bytes", "plan_elems", "plan_elem_bytes")]


class EnvDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dynamic", C.c_int32), ("num_envs", C.c_int32), ("num_plans", C.c_int32),
                ("obs_dtype", C.c_int32), ("static_plan", C.c_int32), ("seed", C.c_uint64), ("env_id_base", C.c_int64),
                ("total_step", C.c_int32), ("rules", C.c_int32), ("frame_value", C.c_int32), ("obs_scalars", C.c_int32),
                ("obs_tail", C.c_int32), ("reserved", C.c_int32)]


class RolloutRecord(C.Structure):
    _fields_ = [("actions", C.c_void_p), ("step_size", C.c_void_p), ("plan_idx", C.c_void_p), ("first", C.c_void_p)]


class State(C.Structure):
    _fields_ = [("hdr", C.c_void_p), ("episode", C.c_void_p), ("grid", C.c_void_p), ("plans", C.c_void_p),
                ("plan_tb", C.c_void_p), ("stat_episodes", C.c_void_p), ("stat_return", C.c_void_p),
                ("stat_iou_fx", C.c_void_p)]


TRAJ_INFO_WINDOWS = 64


class TrajInfo(C.Structure):
    """snac_traj_info (include/snac_hip.h): what snac_traj_alloc measured while it built a block."""
    _fields_ = ([(n, C.c_int32) for n in ("layout", "rebuilds", "pool_groups", "probe_launches", "windows", "windows_slow")]
                + [(n, C.c_float) for n in ("self_us_per_gib", "fast_us_per_gib", "slow_us_per_gib", "window_max_us_per_gib",
                                            "window_mean_us_per_gib", "block_us_per_gib", "build_ms")]
                + [("bytes", C.c_uint64), ("window_us", C.c_float * TRAJ_INFO_WINDOWS)])


class SnacError(RuntimeError):
    pass


def build(force=False):
    """Compile libsnac_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    src = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".inc")) or f == "Makefile"]
    src.append(os.path.join(INCLUDE, "snac_hip.h"))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        # one object per translation unit, compiled in parallel (snac_amd/csrc/Makefile: `fast`)
        jobs = str(max(1, min(8, os.cpu_count() or 1)))
        subprocess.check_call(["make", "-s", "-C", CSRC, "JOBS=" + jobs] + (["clean", "fast"] if force else ["fast"]))
    return LIB_PATH


def kernel_source_sha16():
    """First 16 hex digits of the sha256 over the headline kernel's source (snac_dev.h + k_roll2d.hip): what profiles/traffic.json
    is stamped with, so that bench.py reports counter traffic only for the kernel source the counters were taken with."""
    import hashlib

    h = hashlib.sha256()
    for f in ("snac_dev.h", "k_roll2d.hip"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SnacError("libsnac_hip.so is not built (%s); run __graft_entry__.build() or `make -C snac_amd/csrc`. "
                            "There is no CPU fallback." % LIB_PATH)
        # torch first: libsnac_hip.so needs libamdhip64.so.7 and must bind to the HIP runtime that PyTorch-ROCm
        # bundles (same SONAME) -- loading the system copy beside it would put two runtimes in one process.
        import torch  # noqa: F401

        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.snac_version.restype = C.c_int
        L.snac_last_error.restype = C.c_char_p
        L.snac_last_kernel.restype = C.c_char_p
        L.snac_stream_sync.argtypes = [vp]
        L.snac_env_sizes.argtypes = [C.c_int, C.c_int, C.POINTER(Sizes)]
        L.snac_obs_dim.argtypes = [C.POINTER(EnvDesc)]
        L.snac_reset.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), vp, vp, vp, vp]
        L.snac_reset_scalar.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, vp, vp]
        L.snac_step_scalar.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_uint32, C.c_int32, C.c_int32, C.c_int, vp, vp, vp, vp]
        L.snac_step.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_uint32, vp, vp, C.c_int, vp, vp, vp, vp]
        L.snac_rollout.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, C.c_uint32, vp, vp, C.c_int, vp, vp, vp, vp]
        L.snac_rollout_rec.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, C.c_uint32, vp, vp, C.c_int, vp, vp, vp,
                                       C.POINTER(RolloutRecord), vp]
        L.snac_replay_gather.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp]
        L.snac_replay_gather_tiled.argtypes = L.snac_replay_gather.argtypes
        L.snac_rollout_tiled.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, C.c_uint32, vp, vp, C.c_int32, C.c_int32, vp, vp, vp,
                                         C.POINTER(RolloutRecord), vp]
        L.snac_make_plans.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int64, vp, vp, vp]
        L.snac_observe.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), vp, vp]
        L.snac_iou.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), vp, vp]
        L.snac_export_grid.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), vp, vp]
        L.snac_transition.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, vp]
        L.snac_nodes2d_pack.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), vp, C.c_int32, vp, C.c_int32, vp, vp]
        L.snac_nodes2d_unpack.argtypes = [C.POINTER(EnvDesc), vp, C.c_int32, vp, C.c_int32, C.POINTER(State), vp, vp]
        L.snac_transition_nodes2d.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), vp, C.c_int32, C.c_int32, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, vp]
        L.snac_import_state.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.snac_plans_from_grids.argtypes = [C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, C.c_int32, C.POINTER(State), C.c_int32, vp, vp, vp, vp]
        L.snac_mailbox_create.argtypes = [C.POINTER(EnvDesc), C.c_uint32, C.POINTER(vp)]
        L.snac_mailbox_row.argtypes, L.snac_mailbox_row.restype = [vp], vp
        L.snac_mailbox_reward.argtypes, L.snac_mailbox_reward.restype = [vp], vp
        L.snac_mailbox_done.argtypes, L.snac_mailbox_done.restype = [vp], vp
        L.snac_mailbox_step_n.argtypes = [vp, C.POINTER(EnvDesc), C.POINTER(State), vp, vp]
        L.snac_mailbox_touch.argtypes = [vp]
        L.snac_mailbox_step.argtypes = [vp, C.POINTER(EnvDesc), C.POINTER(State), C.c_int32, C.c_int32]
        L.snac_mailbox_quit.argtypes = [vp]
        L.snac_mailbox_settle.argtypes = [vp]
        L.snac_mailbox_destroy.argtypes = [vp]
        L.snac_mailbox_stats.argtypes = [vp, C.POINTER(C.c_uint32 * 8)]
        L.snac_obs_equal.argtypes = [C.POINTER(EnvDesc), vp, vp, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp]
        L.snac_discounted_return.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
        L.snac_traj_alloc.argtypes = [C.c_size_t, C.c_int, C.POINTER(vp)]
        L.snac_traj_alloc_ex.argtypes = [C.c_size_t, C.c_int, C.c_size_t, vp, C.POINTER(vp)]
        L.snac_traj_free.argtypes = [vp]
        L.snac_traj_layout.argtypes = [vp]
        L.snac_traj_describe.argtypes = [vp, C.POINTER(TrajInfo)]
        L.snac_traj_reserved_bytes.restype = C.c_uint64
        for n in EXPORTS:
            getattr(L, n)
        if L.snac_version() != ABI_VERSION:
            raise SnacError("libsnac_hip.so ABI version mismatch")
        _lib = L
    return _lib


def tuning():
    """The library's dispatch table as {environment variable: (effective value, what it decides)} (snac_tuning)."""
    buf = C.create_string_buffer(16384)
    check(lib().snac_tuning(buf, len(buf)))
    out = {}
    for ln in buf.value.decode().splitlines():
        kv, _, what = ln.partition("  # ")
        k, _, v = kv.partition("=")
        out[k] = (int(v), what)
    return out


def check(rc):
    if rc != SNAC_OK:
        raise SnacError("libsnac_hip: %s (code %d)" % (lib().snac_last_error().decode(), rc))


def env_sizes(kind, dynamic):
    s = Sizes()
    check(lib().snac_env_sizes(kind, int(bool(dynamic)), C.byref(s)))
    return s
