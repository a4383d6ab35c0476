"""Which kernel takes which call: snac_rollout (T = 4 ticks, rows into plain memory), snac_step, snac_reset (all envs / with a mask), snac_observe,
snac_iou and snac_transition (a wave of N / 2 edges) for every kind, row type and layout over a ladder of batch sizes, as the library's dispatch table (snac_hip.hip KNOBS, environment overrides included) decides it on
this box.  One line per (entry point, kind, rows): the batch sizes at which the kernel CHANGES.  Run it after touching a threshold:

    gpurun -- python tools/dispatch_table.py > gpurun_out/dispatch.txt        (copied to profiles/r06_dispatch.txt)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from snac_amd import BatchedDMPEnv, _lib  # noqa: E402

LADDER = [4, 64, 256, 1024, 2048, 3072, 3584, 4096, 8192, 12288, 16384, 20480, 24576, 28672, 32768, 36864, 40960, 45056, 49152, 57344, 65536, 81920,
          98304, 131072, 196608, 262144, 278528, 376832, 475136, 475140, 524288, 557056, 786432, 1048576]


def kernel_of(call):
    call()
    torch.cuda.synchronize()
    return _lib.lib().snac_last_kernel().decode()


def changes(names):
    out, prev = [], None
    for n, k in names:
        if k != prev:
            out.append("%d: %s" % (n, k))
            prev = k
    return "  |  ".join(out)


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
    print("# tools/dispatch_table.py: the kernel behind each call by batch size (first batch size of the ladder at which it changes); ladder:",
          " ".join(str(n) for n in LADDER if n <= top))
    for kind in (1, 2, 3):
        for dt, dn in ((torch.float64, "f64"), (torch.float32, "f32")):
            for layout in (None, "ppo"):
                roll, step, rst, rstm, obsv, iou, edge = [], [], [], [], [], [], []
                for n in LADDER:
                    if n > top or (layout and n > 262144):           # (the 451-value rows of a million envs are 3.6 GB a tick)
                        continue
                    kw = {"layout": layout} if layout else {}
                    env = BatchedDMPEnv(kind, True, n, seed=1, obs_dtype=dt, **kw)
                    env.reset()
                    T = 4
                    buf = torch.empty((T, n, env.obs_dim), dtype=dt, device="cuda")
                    rew = torch.empty((T, n), dtype=torch.float32, device="cuda")
                    done = torch.empty((T, n), dtype=torch.uint8, device="cuda")
                    roll.append((n, kernel_of(lambda: env.rollout(T, obs="all", out=buf, reward_out=rew, done_out=done))))
                    out = (buf[0], rew[0], done[0])
                    step.append((n, kernel_of(lambda: env.step(None, None, auto_reset=True, out=out))))
                    if n <= 524288:                                  # the other calls of the path: reset, reset(mask), observe, iou, one wave of tree edges
                        mask = (torch.arange(n, device="cuda") % 3 == 0).to(torch.uint8)
                        rst.append((n, kernel_of(lambda: env.reset())))
                        rstm.append((n, kernel_of(lambda: env.reset(mask=mask))))
                        obsv.append((n, kernel_of(lambda: env.observe())))
                        iou.append((n, kernel_of(lambda: env.iou())))
                        if not layout and n >= 8:
                            m = n // 2
                            src = torch.arange(m, device="cuda", dtype=torch.int32)
                            dst = (m + torch.arange(m, device="cuda")).to(torch.int32)
                            acts = torch.zeros(m, dtype=torch.int8, device="cuda")
                            edge.append((m, kernel_of(lambda: env.transition(acts, None, src, dst, t=0))))
                    del env, buf, rew, done, out
                tag = "%dD %s %s rows" % (kind, dn, layout or "canonical")
                print("rollout  %-24s %s" % (tag, changes(roll)), flush=True)
                print("step     %-24s %s" % (tag, changes(step)), flush=True)
                print("reset    %-24s %s" % (tag, changes(rst)), flush=True)
                print("reset(m) %-24s %s" % (tag, changes(rstm)), flush=True)
                print("observe  %-24s %s" % (tag, changes(obsv)), flush=True)
                print("iou      %-24s %s" % (tag, changes(iou)), flush=True)
                if edge:
                    print("edges    %-24s %s" % (tag, changes(edge)), flush=True)
    env = {k: v for k, v in os.environ.items() if k.startswith("SNAC_")}
    print("# environment overrides in effect:", env or "none")


if __name__ == "__main__":
    main()
