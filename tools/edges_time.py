"""snac_transition alone (raw C ABI, device time over back-to-back launches): one search wave of m tree edges on a node pool of 2^20 rows.
Modes: `one` = every edge has its own random parent (what bench.py's extras time), `all` = random parents x all actions (children of a
parent share its record: tools/bench_configs.py).  SNAC_EDGES3D=0 keeps 3D edges on k_transition3d.

    gpurun -- python tools/edges_time.py [kind] [m] [one|all] [reps] [nodes]        nodes: 2D on one-record-per-node pools (snac_transition_nodes2d)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv, _lib  # noqa: E402


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
    mode = sys.argv[3] if len(sys.argv) > 3 else "one"
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    pool = 1 << 20
    env = BatchedDMPEnv(kind, True, pool, seed=1)
    env.reset()
    env.rollout(20, obs=None)
    A = env.num_actions
    dev = env.device
    if mode == "all":
        parents = m // A
        m = parents * A
        src = torch.randint(0, pool - m, (parents,), device=dev, dtype=torch.int32).repeat_interleave(A).contiguous()
        acts = torch.arange(A, device=dev, dtype=torch.int8).repeat(parents).contiguous()
    else:
        src = torch.randint(0, pool - m, (m,), device=dev, dtype=torch.int32)
        acts = torch.randint(0, A, (m,), device=dev).to(torch.int8)
    dst = (pool - m + torch.arange(m, device=dev, dtype=torch.int32)).contiguous()
    obs = torch.empty((m, env.obs_dim), dtype=torch.float64, device=dev)
    rew = torch.empty(m, dtype=torch.float32, device=dev)
    done = torch.empty(m, dtype=torch.uint8, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    nodes = None
    if len(sys.argv) > 5 and sys.argv[5] == "nodes":
        from snac_amd import NodePool2D

        nodes = NodePool2D(env, pool)
        nodes.load()

    def call():
        if nodes is not None:
            _lib.check(env._lib.snac_transition_nodes2d(C.byref(env._desc), C.byref(env._state), vp(nodes.records), pool, m, vp(src), vp(dst), 0, vp(acts), None,
                                                        vp(obs), vp(rew), vp(done), env._stream()))
            return
        _lib.check(env._lib.snac_transition(C.byref(env._desc), C.byref(env._state), m, vp(src), vp(dst), 0, vp(acts), None, vp(obs), vp(rew), vp(done), env._stream()))

    for _ in range(10):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    rec = {1: 64, 2: 80, 3: 800}[kind] + 20
    per = 2 * rec + env.obs_dim * 8 + 5 + 9
    if len(sys.argv) <= 5 or sys.argv[5] != "nodes" or True:
        pass
    print("%dD %s: %d edges (%s) %.4f ms  %.3e edges/s  %.0f GB/s of %d B per edge = %.2f of 8 TB/s" % (
        kind, _lib.lib().snac_last_kernel().decode(), m, mode, ms, m / ms * 1e3, per * m / ms / 1e6, per, per * m / ms / 1e6 / 8000))


if __name__ == "__main__":
    main()
