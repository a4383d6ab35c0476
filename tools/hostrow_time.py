import sys, time
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from snac_amd import BatchedDMPEnv
for n in (1, 256, 4096, 65536):
    env = BatchedDMPEnv(2, True, n, seed=1)
    env.reset()
    host = env.new_host_obs()
    dev = env._new_obs()
    for name, out in (("device row + .cpu()", dev), ("host row + sync", host)):
        for rep in range(2):
            t0 = time.perf_counter()
            K = 200 if n < 65536 else 30
            for i in range(K):
                env.step_scalar(i % 5, 1 + i % 3, auto_reset=True, out=out)
                if out is dev:
                    x = dev.cpu().numpy()
                else:
                    env.sync(); x = host.numpy().copy()
            dt = (time.perf_counter() - t0) / K
        print("N=%6d %-22s %8.1f us/tick  %.2f GB/s" % (n, name, dt * 1e6, n * 51 * 8 / dt / 1e9), flush=True)
