import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snac_amd import VectorizedEnvWrapper, plans
for n in (16, 256, 1024, 4096, 65536):
    env = VectorizedEnvWrapper((2, True, plans.dataset(2, "dense", "train")), num_envs=n)
    np.random.seed(0)
    env.reset()
    acts = np.random.randint(5, size=n)
    for _ in range(5): env.step(acts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    T = 50 if n >= 4096 else 500
    for _ in range(T):
        env.step(np.random.randint(5, size=n))
    dt = (time.perf_counter() - t0) / T
    print("VectorizedEnvWrapper.step N=%d: %.3f ms/tick (numpy in, numpy out: %.1f MB D2H per tick) -> %.3e env-steps/s" % (n, dt * 1e3, n * 413 / 1e6, n / dt))
