"""GPU: the per-tick step() launch is hipGraph-capturable (no allocation, sync or host read inside the C ABI call), so a
launch-bound policy loop can be replayed as a graph.  Inputs are static device tensors refilled between replays."""
import pytest

pytestmark = pytest.mark.gpu


def test_step_replays_as_a_graph():
    import torch
    from snac_amd import BatchedDMPEnv

    n = 4096
    env = BatchedDMPEnv(2, True, n, seed=3)
    ref = BatchedDMPEnv(2, True, n, seed=3)
    env.reset()
    ref.reset()
    a = torch.zeros(n, dtype=torch.int8, device=env.device)
    k = torch.ones(n, dtype=torch.int8, device=env.device)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up launch outside capture (torch's capture protocol)
        env.step(a, k, auto_reset=True)
    torch.cuda.current_stream().wait_stream(side)
    ref.step(a, k, auto_reset=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        obs, rew, done = env.step(a, k, auto_reset=True)
    gen = torch.Generator(device=env.device)
    gen.manual_seed(0)
    for _ in range(300):
        a.copy_(torch.randint(0, 5, (n,), device=env.device, generator=gen).to(torch.int8))
        k.copy_(torch.randint(1, 4, (n,), device=env.device, generator=gen).to(torch.int8))
        g.replay()
        o2, r2, d2 = ref.step(a, k, auto_reset=True)
        assert torch.equal(obs, o2) and torch.equal(rew, r2) and torch.equal(done, d2)
    assert torch.equal(env.environment_memory(), ref.environment_memory())
    assert env.episodic_stats() == ref.episodic_stats() and env.episodic_stats()["episodes"] > 0
