"""A/B timing of snac_rollout through the raw C ABI: python tools/ab_time.py <libsnac_hip.so> [kind] [N] [T] [reps] [obs_mode]
(works with any ABI version whose snac_env_desc / snac_state layouts match; used to compare builds on one box)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from snac_amd import plans  # noqa: E402


class Desc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dynamic", C.c_int32), ("num_envs", C.c_int32), ("num_plans", C.c_int32),
                ("obs_dtype", C.c_int32), ("static_plan", C.c_int32), ("seed", C.c_uint64), ("env_id_base", C.c_int64),
                ("total_step", C.c_int32), ("rules", C.c_int32), ("frame_value", C.c_int32), ("obs_scalars", C.c_int32),
                ("obs_tail", C.c_int32), ("reserved", C.c_int32)]


class State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("hdr", "episode", "grid", "plans", "plan_tb", "stat_episodes", "stat_return", "stat_iou_fx")]


def main():
    lib = sys.argv[1]
    kind = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 600
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    obs_mode = int(sys.argv[6]) if len(sys.argv) > 6 else 1      # 1: every observation written, 0: none (phase 1 alone)
    want_rd = int(sys.argv[7]) if len(sys.argv) > 7 else 3       # bit 0: reward written, bit 1: done written
    L = C.CDLL(lib)
    table = plans.dataset(kind, "dense", "train")
    packed, tb = plans.pack_plans(kind, table)
    dev = "cuda"
    ge = {1: 32, 2: 20, 3: 400}[kind]
    D = 7 if kind == 1 else 51
    hdr = torch.zeros((N, 4), dtype=torch.int32, device=dev)
    epi = torch.full((N,), -1, dtype=torch.int32, device=dev)
    grid = torch.zeros((N, ge), dtype=torch.int32 if kind == 2 else torch.int16, device=dev)
    d_plans = torch.from_numpy(packed.view(np.int32) if kind == 2 else packed).to(dev)
    d_tb = torch.from_numpy(tb).to(dev)
    stats = torch.zeros((3, N), dtype=torch.int64, device=dev)
    desc = Desc(kind, 1, N, len(packed), 0, 0, 1, 0, 0, 0)
    st = State(hdr.data_ptr(), epi.data_ptr(), grid.data_ptr(), d_plans.data_ptr(), d_tb.data_ptr(), stats[0].data_ptr(),
               stats[1].data_ptr(), stats[2].data_ptr())
    obs = torch.empty((T, N, D), dtype=torch.float64, device=dev)
    rew = torch.empty((T, N), dtype=torch.float32, device=dev)
    done = torch.empty((T, N), dtype=torch.uint8, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    assert L.snac_reset(C.byref(desc), C.byref(st), None, None, None, stream) == 0

    def call(t0):
        assert L.snac_rollout(C.byref(desc), C.byref(st), T, C.c_uint32(t0), None, None, obs_mode, vp(obs), vp(rew) if want_rd & 1 else None,
                              vp(done) if want_rd & 2 else None, stream) == 0

    call(0); call(T)
    torch.cuda.synchronize()
    times = []
    for i in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); call((2 + i) * T); b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    print("%s kind=%d N=%d T=%d: min %.3f ms  median %.3f ms  checksum %d" % (
        os.path.basename(lib), kind, N, T, min(times), sorted(times)[len(times) // 2], int(stats.sum().item()) & 0xFFFFFFFF))


if __name__ == "__main__":
    main()
