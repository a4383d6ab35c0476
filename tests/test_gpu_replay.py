"""GPU: SURVEY.md section 8 rows f1 / f2 -- the device-resident replay ring (what the reference's DQN scripts keep in
a python deque) and the snapshot / fork primitives (the MCTS variants' functional transition, batched)."""
import numpy as np
import pytest

import helpers
import rng_spec

pytestmark = pytest.mark.gpu


def _reference_style_memory(dim, dyn, n, T, seed, table):
    """Restatement of the prefill loop of script/DQN/2d/DQN_2d_dynamic.py:184-199 on the CPU oracle, for n independent
    envs: state = env.reset(); prev = state[0]; each step store (prev, action, reward, next_state[0], plan); prev = next;
    on done start over from reset().  Actions / step sizes / plan indices follow the counter RNG (one tick at a time)."""
    orc = helpers.oracle().OracleBatch(dim, dyn, n, table, seed=seed)
    prev = orc.reset()
    reset_obs = prev.copy()
    A = helpers.DIMS[dim]["A"]
    mem = []
    for t in range(T):
        w = rng_spec.words(seed, rng_spec.STREAM_STEP, np.arange(n, dtype=np.uint64), np.uint64(t))
        acts = rng_spec.action_of(w, A)
        nxt, rew, done = orc.step(t, auto_reset=True)
        pidx = orc.state()["plan_idx"]
        mem.append(dict(s=prev.copy(), a=acts.astype(np.int64), r=rew.copy(), s_next=nxt.copy(), done=done.copy(), plan_idx=pidx.copy()))
        prev = np.where(done[:, None] != 0, reset_obs, nxt)       # after done the reference calls reset() again
    return mem


@pytest.mark.parametrize("kind", [(1, True), (2, False), (2, True), (3, True)], ids=str)
@pytest.mark.parametrize("f32", [False, True], ids=["f64ring", "f32ring"])
@pytest.mark.parametrize("layout", ["ticks", "tiled"])
def test_ring_holds_the_reference_replay_tuples(kind, f32, layout):
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    dim, dyn = kind
    n, T, seed = 24, 90, 6
    tag = ("sin_train" if dim == 1 else "dense_train") if dyn else "p0"
    table = helpers.plan_table(dim, dyn, tag)
    full = table.reshape(len(table), 30) if dim == 1 else table.reshape(len(table), 26, 26)
    env = BatchedDMPEnv(dim, dyn, n, plans=full, seed=seed, obs_dtype=torch.float32 if f32 else torch.float64)
    env.reset()
    ring = ReplayRing(env, capacity_ticks=128, layout=layout)
    ring.collect(40)
    ring.collect(T - 40)
    assert ring.valid_ticks() == T and len(ring) == T * n
    mem = _reference_style_memory(dim, dyn, n, T, seed, table)
    slots = np.repeat(np.arange(T), n)
    envs = np.tile(np.arange(n), T)
    got = ring.gather(slots, envs)
    s = got["s"].cpu().numpy().reshape(T, n, -1)
    s_next = got["s_next"].cpu().numpy().reshape(T, n, -1)
    plan = got["plan"].cpu().numpy().reshape(T, n, -1)
    for t in range(T):
        m = mem[t]
        assert np.array_equal(s[t], m["s"].astype(np.float32)), t           # float32 as torch.FloatTensor(s) makes it
        assert np.array_equal(s_next[t], m["s_next"].astype(np.float32)), t
        want_plan = (full[m["plan_idx"]] if dim == 1 else full[m["plan_idx"]][:, 3:23, 3:23].reshape(n, -1)).astype(np.float32)
        assert np.array_equal(plan[t], want_plan), t
    assert np.array_equal(got["action"].cpu().numpy().reshape(T, n), np.stack([m["a"] for m in mem]))
    assert np.array_equal(got["reward"].cpu().numpy().reshape(T, n), np.stack([m["r"] for m in mem]))
    assert np.array_equal(got["done"].cpu().numpy().reshape(T, n).astype(np.uint8), np.stack([m["done"] for m in mem]))


@pytest.mark.parametrize("layout", ["ticks", "tiled"])
def test_ring_wraps_and_samples(layout):
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    n, cap, seed = 16, 32, 2
    table = helpers.plan_table(2, True, "dense_train")
    env = BatchedDMPEnv(2, True, n, plans=table.reshape(-1, 26, 26), seed=seed)
    env.reset()
    ring = ReplayRing(env, cap, layout=layout)
    for _ in range(5):
        ring.collect(20)                                              # 100 ticks through a 32-slot ring
    T = 100
    assert ring.valid_ticks() == cap - 1
    mem = _reference_style_memory(2, True, n, T, seed, table)
    slots = ring.slots().cpu().numpy()                                # oldest first: ticks T-31 .. T-1
    assert len(slots) == cap - 1 and slots[-1] == (T - 1) % cap
    for j, slot in enumerate(slots):
        t = T - (cap - 1) + j
        got = ring.gather(np.full(n, slot), np.arange(n), with_plan=False)
        assert np.array_equal(got["s"].cpu().numpy(), mem[t]["s"].astype(np.float32)), t
        assert np.array_equal(got["s_next"].cpu().numpy(), mem[t]["s_next"].astype(np.float32)), t
    g = torch.Generator(device=env.device)
    g.manual_seed(0)
    b = ring.sample(256, generator=g)
    assert b["s"].shape == (256, 51) and b["plan"].shape == (256, 20, 20) and b["s"].dtype == torch.float32
    assert set(np.unique(b["plan"].cpu().numpy()).tolist()) <= {0.0, 1.0}
    with pytest.raises(ValueError):
        ring.collect(cap + 1)


def test_snapshot_restore_and_fork():
    import torch
    from snac_amd import BatchedDMPEnv

    env = BatchedDMPEnv(3, True, 32, seed=8)
    env.reset()
    env.rollout(40, obs=None)
    sd = env.state_dict()
    o1, r1, d1 = env.rollout(25)
    mem1 = env.environment_memory().clone()
    env.load_state_dict(sd)
    o2, r2, d2 = env.rollout(25)
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(mem1, env.environment_memory())
    # fork: children = copies of chosen parents, each stepped with its own action (functional transition, batched)
    env.load_state_dict(sd)
    parents = [3, 3, 3, 3, 3, 3, 3, 3, 9, 9]
    acts = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 4, 0], dtype=torch.int8)
    ks = torch.tensor([3, 3, 3, 3, 1, 1, 1, 1, 2, 2], dtype=torch.int8)
    child = env.fork(parents)
    before = env.environment_memory().clone()
    oc, rc, dc = child.step(acts, ks)
    assert torch.equal(env.environment_memory(), before)              # the parents are untouched
    for j, (p, a, k) in enumerate(zip(parents, acts.tolist(), ks.tolist())):
        env.load_state_dict(sd)
        o, r, d = env.step(torch.full((32,), a, dtype=torch.int8), torch.full((32,), k, dtype=torch.int8))
        assert torch.equal(o[p], oc[j]) and r[p] == rc[j] and d[p] == dc[j]
        assert torch.equal(env.environment_memory()[p], child.environment_memory()[j])
    with pytest.raises(ValueError):
        BatchedDMPEnv(3, True, 8).load_state_dict(sd)


@pytest.mark.parametrize("kind", [(2, True), (3, True)], ids=str)
def test_sequence_sampling_stays_inside_one_episode(kind):
    """DRQN-style windows (Memory.get_batch, script/DRQN/2d/DRQN_2D_dynamic_training.py:131-143): every window is L consecutive
    transitions of one env within one episode, and its tuples are the reference's (chained: s[j+1] == s_next[j])."""
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    dim, dyn = kind
    n, cap, L, B = 64, 96, 8, 500
    env = BatchedDMPEnv(dim, dyn, n, seed=12)
    env.reset()
    ring = ReplayRing(env, cap)
    ring.collect(cap)
    ring.collect(40)                                              # wrapped: windows may cross the physical end of the ring
    g = torch.Generator(device="cuda").manual_seed(1)
    out = ring.sample_sequences(B, L, generator=g)
    assert out["s"].shape == (B, L, env.obs_dim) and out["action"].shape == (B, L) and out["done"].shape == (B, L)
    slots, ei = out["slot"], out["env"]
    assert torch.equal((slots[:, 1:] - slots[:, :-1]) % cap, torch.ones((B, L - 1), dtype=slots.dtype, device=slots.device))
    first = ring.first[slots, ei[:, None]]
    assert int(first[:, 1:].sum()) == 0                          # no episode starts inside a window ...
    assert not bool(out["done"][:, :-1].any())                   # ... so only the last step may be terminal
    assert torch.equal(out["s"][:, 1:], out["s_next"][:, :-1])   # consecutive tuples chain
    flat = ring.gather(slots.reshape(-1), ei[:, None].expand(B, L).reshape(-1), with_plan=False)
    assert torch.equal(flat["s_next"].view(B, L, -1), out["s_next"]) and torch.equal(flat["reward"].view(B, L), out["reward"])
    assert torch.equal(out["plan"], ring.gather(slots[:, 0], ei)["plan"])
    oldest = (ring.head - ring.valid_ticks()) % cap
    assert bool((((slots - oldest) % cap) < ring.valid_ticks()).all())          # only addressable slots
    with pytest.raises(ValueError):
        ring.sample_sequences(8, cap + 1)


def test_ring_attached_in_mid_episode_and_guard_against_outside_steps():
    """slot 0's predecessor is slot cap - 1 of the ring; it is seeded with the envs' current observation, so a
    ring attached to envs in mid-episode returns a real `s` for its first transitions; stepping the envs outside the ring
    between collect() calls is refused."""
    import torch

    from snac_amd import BatchedDMPEnv, ReplayRing, SnacError

    env = BatchedDMPEnv(2, True, 64, seed=8)
    env.reset()
    env.rollout(37)                                              # mid-episode
    before = env.observe().clone()
    ring = ReplayRing(env, 16)
    ring.collect(5)
    assert int(ring.first[0].sum()) < 64                         # most envs did not start an episode at slot 0
    b = ring.gather(torch.zeros(64, dtype=torch.int32), torch.arange(64, dtype=torch.int32), with_plan=False)
    cont = ring.first[0] == 0
    assert torch.equal(b["s"][cont], before[cont].float()) and torch.equal(b["s_next"], ring.obs_at(0).float())
    env.step(auto_reset=True)                                    # behind the ring's back
    with pytest.raises(SnacError, match="outside the ring"):
        ring.collect(1)


@pytest.mark.parametrize("n", [5, 64, 200, 4100])
def test_tiled_ring_equals_the_tick_ring(n):
    """The same collection into both ring layouts (ragged batches, ring wraps, mid-episode attach): every row, every sampled
    minibatch and every DRQN window identical."""
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    envs = [BatchedDMPEnv(2, True, n, seed=12, total_step=40) for _ in range(2)]
    rings = []
    for e, layout in zip(envs, ("ticks", "tiled")):
        e.reset()
        e.rollout(13, obs=None)                                       # attach in mid-episode
        # the tile-major ring in trajectory memory (snac_traj_alloc), the tick ring in torch.empty memory
        rings.append(ReplayRing(e, 48, layout=layout, place_candidates=3 if layout == "tiled" else 0,
                                memory="vmm" if layout == "tiled" else "malloc"))
    assert rings[1].memory == "vmm" and rings[1].obs.data_ptr() % (2 << 20) == 0 and rings[0].memory == "malloc"
    for T in (20, 30, 48, 7):
        for r in rings:
            r.collect(T)
    a, b = rings
    assert tuple(b.obs.shape) == ((n + 63) // 64, 48, 64, 51)
    for slot in range(48):
        assert torch.equal(a.obs_at(slot), b.obs_at(slot))
    for name in ("reward", "done", "action", "step_size", "plan_idx", "first"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    slots = torch.randint(0, 48, (500,), device="cuda")
    idx = torch.randint(0, n, (500,), device="cuda")
    assert torch.equal(a.row(slots, idx), b.row(slots, idx))
    ga, gb = torch.Generator(device="cuda"), torch.Generator(device="cuda")
    ga.manual_seed(3); gb.manual_seed(3)
    sa, sb = a.sample(300, generator=ga), b.sample(300, generator=gb)
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    qa, qb = a.sample_sequences(40, 6, generator=ga), b.sample_sequences(40, 6, generator=gb)
    assert all(torch.equal(qa[k], qb[k]) for k in qa)


@pytest.mark.parametrize("layout", ["ticks", "tiled"])
def test_a_ring_of_more_than_2_31_values(layout):
    """A ring of 65 536 envs x 700 ticks holds 2.34e9 float32 values (the gather's row offsets pass 2^31): samples from its far end
    equal those of a small ring that follows the batch's last 64 envs alone (env_id_base keys the counter RNG)."""
    import torch
    from snac_amd import BatchedDMPEnv, ReplayRing

    n, cap, w = 65536, 700, 64
    table = helpers.plan_table(2, True, "dense_train")
    full = table.reshape(len(table), 26, 26)
    big = BatchedDMPEnv(2, True, n, plans=full, seed=8, obs_dtype=torch.float32, total_step=150)
    small = BatchedDMPEnv(2, True, w, plans=full, seed=8, obs_dtype=torch.float32, total_step=150, env_id_base=n - w)
    big.reset()
    small.reset()
    rb, rs = ReplayRing(big, cap, layout=layout), ReplayRing(small, cap, layout=layout)
    assert rb.obs.numel() > 2 ** 31
    for T in (300, 400):
        rb.collect(T)
        rs.collect(T)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    slot = torch.randint(cap - 80, cap, (4096,), device="cuda", generator=g)
    e = torch.randint(0, w, (4096,), device="cuda", generator=g)
    a, b = rb.gather(slot, e + (n - w)), rs.gather(slot, e)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(rb.obs_at(cap - 1)[n - w:], rs.obs_at(cap - 1))
