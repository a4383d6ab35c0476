import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from snac_amd import BatchedDMPEnv, _lib
kind = int(sys.argv[1]); n = int(sys.argv[2]); layout = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "none" else None
f32 = len(sys.argv) > 4 and sys.argv[4] == "f32"
kw = dict(layout=layout) if layout else {}
e = BatchedDMPEnv(kind, True, n, seed=1, obs_dtype=torch.float32 if f32 else torch.float64, **kw)
e.reset()
out = (torch.empty((n, e.obs_dim), dtype=e.obs_dtype, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda"))
acts = torch.randint(0, e.num_actions, (n,), dtype=torch.int8, device="cuda")
ks = torch.randint(1, 4, (n,), dtype=torch.int8, device="cuda")
for mode in ("counter RNG", "explicit"):
    a_, k_ = (None, None) if mode == "counter RNG" else (acts, ks)
    for _ in range(50): e.step(a_, k_, auto_reset=True, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300): e.step(a_, k_, auto_reset=True, out=out)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 300 * 1e3
    byt = n * (e.obs_dim * (4 if f32 else 8) + 5)
    print("%dD step N=%d layout=%s %s %s: %.1f us/tick  %.2f TB/s written  kernel %s" % (kind, n, layout, "f32" if f32 else "f64", mode, us, byt / us / 1e6, _lib.lib().snac_last_kernel().decode()))
