"""GPU: inputs the reference never produces must not corrupt state: explicit step sizes outside {1,2,3} are clamped, actions
outside [0, A) only advance count_step, an over-long time limit is refused."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,dyn", [(1, False), (2, True), (3, True), (3, False)])
def test_out_of_range_step_sizes_are_clamped(kind, dyn):
    import torch
    from snac_amd import BatchedDMPEnv

    N, T = 2048, 40
    a, b = BatchedDMPEnv(kind, dyn, N, seed=4), BatchedDMPEnv(kind, dyn, N, seed=4)
    a.reset(); b.reset()
    g = torch.Generator().manual_seed(kind)
    acts = torch.randint(0, a.num_actions, (T, N), generator=g, dtype=torch.int8).cuda()
    wild = torch.randint(-128, 128, (T, N), generator=g, dtype=torch.int16).to(torch.int8).cuda()
    oa, ra, da = a.rollout(T, actions=acts, step_size=wild)
    ob, rb, db = b.rollout(T, actions=acts, step_size=wild.clamp(1, 3))
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    assert torch.equal(a._grid, b._grid) and torch.equal(a._hdr, b._hdr)
    lo, hi = (2, 31) if kind == 1 else (3, 22)
    pos = a.position
    assert int(pos[:, 0].min()) >= lo and int(pos[:, 0].max()) <= hi
    if kind != 1:
        assert int(pos[:, 1].min()) >= lo and int(pos[:, 1].max()) <= hi
    # the single-step and transition entry points clamp too
    o1, _, _ = a.step(acts[0], wild[0]); o2, _, _ = b.step(acts[0], wild[0].clamp(1, 3))
    assert torch.equal(o1, o2)
    o1, _, _ = a.transition(acts[1], wild[1]); o2, _, _ = b.transition(acts[1], wild[1].clamp(1, 3))
    assert torch.equal(o1, o2) and torch.equal(a._grid, b._grid)


@pytest.mark.parametrize("kind", [1, 2, 3])
def test_bad_actions_only_advance_the_clock(kind):
    import torch
    from snac_amd import BatchedDMPEnv

    N = 512
    env = BatchedDMPEnv(kind, False, N, seed=2)
    env.reset()
    env.rollout(15)
    grid, pos, cb, cs = env._grid.clone(), env.position.clone(), env.count_brick.clone(), env.count_step.clone()
    bad = torch.tensor([-128, -1, env.num_actions, 100], dtype=torch.int8).repeat(N // 4).cuda()
    _, r, _ = env.step(bad, torch.ones(N, dtype=torch.int8).cuda())
    assert torch.equal(env._grid, grid) and torch.equal(env.position, pos) and torch.equal(env.count_brick, cb)
    assert torch.equal(env.count_step, cs + 1) and float(r.abs().sum()) == 0.0


def test_time_limit_above_3000_is_refused():
    from snac_amd import BatchedDMPEnv, SnacError

    env = BatchedDMPEnv(2, True, 8, total_step=3001)
    with pytest.raises(SnacError, match="total_step"):
        env.reset()
    BatchedDMPEnv(2, True, 8, total_step=3000).reset()


def test_step_into_preallocated_outputs():
    import torch
    from snac_amd import BatchedDMPEnv

    a, b = BatchedDMPEnv(2, True, 1000, seed=6), BatchedDMPEnv(2, True, 1000, seed=6)
    a.reset(); b.reset()
    bufs = (torch.empty((1000, 51), dtype=torch.float64, device="cuda"), torch.empty(1000, dtype=torch.float32, device="cuda"),
            torch.empty(1000, dtype=torch.uint8, device="cuda"))
    for _ in range(30):
        o1, r1, d1 = a.step(auto_reset=True)
        o2, r2, d2 = b.step(auto_reset=True, out=bufs)
        assert o2.data_ptr() == bufs[0].data_ptr() and torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    with pytest.raises(ValueError):
        b.step(out=(bufs[0][:10], bufs[1], bufs[2]))
    with pytest.raises(ValueError):
        b.step(out=(bufs[0].float(), bufs[1], bufs[2]))
