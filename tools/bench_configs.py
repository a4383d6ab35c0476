"""Secondary timings on the GPU box (not the driver's bench line): every BASELINE config's env type as a fused
rollout, the float32-observation variant, and the per-tick step() API.  Prints one line per measurement."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv  # noqa: E402

ALG = {1: 88, 2: 481, 3: 574}       # SURVEY.md 8d's un-fused contract figures per env-step
WRITTEN = {1: 61, 2: 413, 3: 413}   # what the fused rollout writes: observation row + reward + done


def env_obs_bytes(kind):
    return 8 * (7 if kind == 1 else 51)


def rollout_time(kind, dynamic, n, T, obs_dtype=torch.float64, reps=5, obs="all"):
    env = BatchedDMPEnv(kind, dynamic, n, seed=1, obs_dtype=obs_dtype)
    env.reset()
    buf = env.alloc_trajectory(T, candidates=2)[0] if obs == "all" else None   # snac_traj_alloc memory (two 32 GiB slices)
    env.rollout(T, obs=obs, out=buf)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        a.record()
        env.rollout(T, obs=obs, out=buf)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def step_time(kind, dynamic, n, ticks=200, explicit=False):
    env = BatchedDMPEnv(kind, dynamic, n, seed=1)
    env.reset()
    acts = torch.randint(0, env.num_actions, (n,), dtype=torch.int8, device=env.device) if explicit else None
    ks = torch.randint(1, 4, (n,), dtype=torch.int8, device=env.device) if explicit else None
    for _ in range(10):
        env.step(acts, ks, auto_reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks):
        env.step(acts, ks, auto_reset=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / ticks * 1e3


def main():
    rows = [("1D static p0   N=4096   T=750 ", 1, False, 4096, 750), ("1D static p0   N=65536  T=750 ", 1, False, 65536, 750),
            ("2D dynamic     N=65536  T=600 ", 2, True, 65536, 600), ("2D static      N=65536  T=600 ", 2, False, 65536, 600),
            ("2D dynamic     N=524288 T=75  ", 2, True, 524288, 75), ("3D dynamic     N=16384  T=1000", 3, True, 16384, 1000),
            ("3D dynamic     N=65536  T=250 ", 3, True, 65536, 250), ("3D static      N=16384  T=1300", 3, False, 16384, 1300)]
    for name, kind, dyn, n, T in rows:
        ms = rollout_time(kind, dyn, n, T)
        print("rollout f64 %s  %8.3f ms  %.3e env-steps/s  written %.0f GB/s  (contract figure %.0f GB/s)" % (
            name, ms, n * T / ms * 1e3, WRITTEN[kind] * n * T / ms / 1e6, ALG[kind] * n * T / ms / 1e6))
    # 3D with the reference's own action mix [0.2 x 4 moves, 0.05 x 4 builds] (Env/3D/DMP_simulator_3d_static_circle.py:361-362):
    # explicit int8 actions drawn with that distribution on the device, step sizes from the counter RNG
    for dyn, T in ((True, 1000), (False, 1300)):
        n = 16384
        env = BatchedDMPEnv(3, dyn, n, seed=1)
        env.reset()
        g = torch.Generator(device=env.device).manual_seed(1)
        probs = torch.tensor([0.2] * 4 + [0.05] * 4, device=env.device)
        acts = torch.multinomial(probs, T * n, replacement=True, generator=g).to(torch.int8).reshape(T, n)
        buf = env.alloc_trajectory(T, candidates=2)[0]
        env.rollout(T, actions=acts, out=buf)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            a.record(); env.rollout(T, actions=acts, out=buf); b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        st = env.episodic_stats()
        print("rollout f64 3D %s reference action mix [0.2x4, 0.05x4] N=16384 T=%d  %8.3f ms  %.3e env-steps/s  (%d episodes, mean IoU %.4f)" % (
            "dynamic" if dyn else "static ", T, best, n * T / best * 1e3, st["episodes"], st["iou_fx_sum"] / 2.0 ** 40 / max(st["episodes"], 1)))
    ms = rollout_time(2, True, 65536, 600, obs_dtype=torch.float32)
    print("rollout f32 2D dynamic     N=65536  T=600   %8.3f ms  %.3e env-steps/s  alg(277 B) %.0f GB/s" % (ms, 65536 * 600 / ms * 1e3, 277 * 65536 * 600 / ms / 1e6))
    ms = rollout_time(2, True, 65536, 600, obs="last")
    print("rollout obs=last 2D dynamic N=65536 T=600   %8.3f ms  %.3e env-steps/s (reward/done only)" % (ms, 65536 * 600 / ms * 1e3))
    for n in (1, 4096, 65536, 524288):
        for explicit in (False, True):
            ms = step_time(2, True, n, explicit=explicit)
            print("step() 2D dynamic N=%-7d %s  %8.4f ms/tick  %.3e env-steps/s" % (n, "explicit a,k" if explicit else "counter RNG ", ms, n / ms * 1e3))


def graph_step_time(n, ticks=300):
    """step() with explicit inputs replayed as a hipGraph (torch.cuda.CUDAGraph) vs the eager call."""
    env = BatchedDMPEnv(2, True, n, seed=1)
    env.reset()
    a = torch.randint(0, 5, (n,), dtype=torch.int8, device=env.device)
    k = torch.randint(1, 4, (n,), dtype=torch.int8, device=env.device)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.step(a, k, auto_reset=True)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.step(a, k, auto_reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ticks):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / ticks * 1e3


def gather_time(n=65536, cap=64, batch=65536, reps=20):
    """snac_replay_gather: float32 (s, s', plan) minibatch out of a float64 ring."""
    from snac_amd import ReplayRing

    env = BatchedDMPEnv(2, True, n, seed=1)
    env.reset()
    ring = ReplayRing(env, cap)
    ring.collect(cap)
    slot = torch.randint(1, cap, (batch,), device=env.device)
    ei = torch.randint(0, n, (batch,), device=env.device)
    ring.gather(slot, ei)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ring.gather(slot, ei)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    moved = batch * (2 * 408 + 2 * 204 + 1600)           # read 2 f64 rows, write 2 f32 rows + 400 f32 plan cells
    return ms, moved / ms / 1e6


def transition_time(kind, pool=1 << 20, parents=1 << 17, reps=20):
    """snac_transition as one search wave: every action of `parents` random pool rows expanded into fresh rows, with the
    observation of every child (out of place, gathered sources).  Uses the raw ABI call: the Python wrapper's argument
    checks cost more than the launch."""
    import ctypes as C

    from snac_amd import _lib

    env = BatchedDMPEnv(kind, True, pool, seed=1)
    env.reset()
    env.rollout(20, obs=None)
    A = env.num_actions
    m = parents * A
    src = torch.randint(0, pool - m, (parents,), device=env.device, dtype=torch.int32).repeat_interleave(A).contiguous()
    dst = (pool - m + torch.arange(m, device=env.device, dtype=torch.int32)).contiguous()
    acts = torch.arange(A, device=env.device, dtype=torch.int8).repeat(parents).contiguous()
    obs = torch.empty((m, env.obs_dim), dtype=torch.float64, device=env.device)
    rew = torch.empty(m, dtype=torch.float32, device=env.device)
    done = torch.empty(m, dtype=torch.uint8, device=env.device)
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def call():
        _lib.check(env._lib.snac_transition(C.byref(env._desc), C.byref(env._state), m, vp(src), vp(dst), 0, vp(acts), None, vp(obs),
                                            vp(rew), vp(done), env._stream()))

    call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, m


def extras():
    for kind, rec in ((1, 64), (2, 80), (3, 800)):
        ms, m = transition_time(kind, parents=(1 << 17) if kind != 3 else (1 << 16))
        per = 2 * (16 + 4 + rec) + env_obs_bytes(kind) + 5       # read + write one state record, write obs + reward + done
        print("transition %dD dynamic: %d edges (random parents x all actions, pool 2^20)  %8.3f ms  %.3e edges/s  alg %.0f GB/s"
              % (kind, m, ms, m / ms * 1e3, per * m / ms / 1e6))
    for n in (65536, 524288):
        ms = step_time(3, True, n)
        print("step() 3D dynamic N=%-7d counter RNG   %8.4f ms/tick  %.3e env-steps/s" % (n, ms, n / ms * 1e3))
    for n in (1, 4096, 65536):
        print("step() as hipGraph replay 2D dynamic N=%-7d explicit a,k  %8.4f ms/tick" % (n, graph_step_time(n)))
    ms, gbs = gather_time()
    print("replay gather 65536 samples from a 64-tick x 65536-env f64 ring  %8.3f ms  %.0f GB/s moved" % (ms, gbs))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extras":
        extras()
    else:
        main()
        extras()
