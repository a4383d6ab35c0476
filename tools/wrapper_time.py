"""VectorizedEnvWrapper.step (numpy in, numpy out) per batch size, resident waves (up to 256 envs) against the launch path.
    gpurun -- python tools/wrapper_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from snac_amd import VectorizedEnvWrapper, plans  # noqa: E402

table = plans.dataset(2, "dense", "train")
for n in (3, 16, 64, 65, 128, 192, 256, 1024, 4096):
    for mailbox in ("1", "0"):
        if n > 256 and mailbox == "1":
            continue
        os.environ["SNAC_MAILBOX"] = mailbox
        env = VectorizedEnvWrapper((2, True, table), num_envs=n)
        np.random.seed(0)
        env.reset()
        T = 200 if n >= 4096 else 2000
        acts = np.random.RandomState(1).randint(0, 5, (T + 100, n))
        for i in range(100):
            env.step(acts[i])
        per = []
        for _ in range(5):
            t0 = time.perf_counter()
            for i in range(T):
                env.step(acts[100 + i])
            per.append((time.perf_counter() - t0) / T)
        per.sort()
        path = "resident waves" if env._mrows is not None else "launch + wait"
        print("VectorizedEnvWrapper.step N=%5d  %-14s  %7.2f us per vector step (min %.2f max %.2f) -> %.3e env-steps/s"
              % (n, path, per[2] * 1e6, per[0] * 1e6, per[-1] * 1e6, n / per[2]), flush=True)
        if env._mrows is not None:
            env.batched.mailbox_close()
        del env
torch.cuda.synchronize()
