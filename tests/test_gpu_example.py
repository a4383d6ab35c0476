"""GPU: the end-to-end example (examples/dqn_batched.py) runs: env ticks, ring appends, gathers and torch's own kernels
interleave on one stream; the ring it fills is consistent with the env's own outputs."""
import importlib.util
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def test_dqn_example_runs_and_learns_something_finite():
    spec = importlib.util.spec_from_file_location("dqn_batched", os.path.join(helpers.ROOT, "examples", "dqn_batched.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    losses, returns = mod.run(envs=512, ticks=60, batch=256, replace=20, prefill=32, log=lines.append)
    assert len(losses) == 60 and all(np.isfinite(losses)) and len(returns) == 3 and len(lines) == 3


def test_ring_append_matches_a_plain_step():
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    a, b = BatchedDMPEnv(2, True, 300, seed=8), BatchedDMPEnv(2, True, 300, seed=8)
    a.reset(); b.reset()
    ring = ReplayRing(a, 16)
    g = torch.Generator().manual_seed(0)
    for t in range(40):                                           # wraps the ring twice
        acts = torch.randint(0, 5, (300,), generator=g, dtype=torch.int8).cuda()
        ring.append(acts)
        o, r, d = b.step(acts, auto_reset=True)
        slot = (ring.head - 1) % ring.cap
        assert torch.equal(ring.obs_at(slot), o) and torch.equal(ring.reward[slot], r) and torch.equal(ring.done[slot].bool(), d)
        assert torch.equal(ring.action[slot], acts)
