"""GPU: randomized parity (hypothesis) -- arbitrary env kind, batch size (ragged tiles included), tick counts, seeds,
action distributions and launch splits; the HIP path must equal the CPU oracle bit for bit, and results must not depend
on how a rollout is cut into launches or on the observation dtype chosen for other launches."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import helpers

pytestmark = pytest.mark.gpu

KIND = st.sampled_from([(1, False), (1, True), (2, False), (2, True), (3, False), (3, True)])


def _tables(dim, dyn):
    tag = ("sin_val" if dim == 1 else "sparse_test") if dyn else ("p2" if dim == 1 else "p1")
    t = helpers.plan_table(dim, dyn, tag)
    return t, (t.reshape(len(t), 30) if dim == 1 else t.reshape(len(t), 26, 26))


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(kind=KIND, n=st.integers(1, 150), seed=st.integers(0, 2**62), base=st.integers(0, 2**40), cuts=st.lists(st.integers(1, 60), min_size=1, max_size=4),
       explicit=st.booleans(), bias=st.floats(0.05, 0.9))
def test_random_rollouts_match_the_oracle(kind, n, seed, base, cuts, explicit, bias):
    import torch
    from snac_amd import BatchedDMPEnv

    dim, dyn = kind
    table, full = _tables(dim, dyn)
    env = BatchedDMPEnv(dim, dyn, n, plans=full, seed=seed, env_id_base=base)
    orc = helpers.oracle().OracleBatch(dim, dyn, n, table, seed=seed, env_id_base=base)
    assert env.reset().cpu().numpy().tobytes() == orc.reset().tobytes()
    A = helpers.DIMS[dim]["A"]
    rng = np.random.default_rng(seed % (2**32))
    t0 = 0
    for T in cuts:
        a = k = None
        if explicit:   # the last action (drop / a build) with probability `bias`, the rest uniform
            a = np.where(rng.random((T, n)) < bias, A - 1, rng.integers(0, A, size=(T, n))).astype(np.int8)
            k = rng.integers(1, 4, size=(T, n)).astype(np.int8)
        og, rg, dg = env.rollout(T, actions=None if a is None else torch.from_numpy(a), step_size=None if k is None else torch.from_numpy(k))
        oc, rc, dc = orc.rollout(T, t0=t0, actions=a, step_size=k)
        assert og.cpu().numpy().tobytes() == oc.tobytes()
        assert rg.cpu().numpy().tobytes() == rc.tobytes()
        assert np.array_equal(dg.cpu().numpy().view(np.uint8), dc)
        t0 += T
    st_ = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(n, -1), st_["grid"])
    assert np.array_equal(env.count_brick.cpu().numpy(), st_["cb"]) and np.array_equal(env.plan_idx.cpu().numpy(), st_["plan_idx"])
    assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()
    s = orc.stats()
    e = env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))


@settings(max_examples=15, deadline=None, suppress_health_check=list(HealthCheck))
@given(kind=KIND, n=st.integers(1, 200), seed=st.integers(0, 2**31), split=st.integers(1, 79))
def test_launch_boundaries_are_invisible(kind, n, seed, split):
    """One rollout of 80 ticks == two rollouts of split + (80 - split) ticks == 80 step() calls."""
    import torch
    from snac_amd import BatchedDMPEnv

    dim, dyn = kind
    _, full = _tables(dim, dyn)
    envs = [BatchedDMPEnv(dim, dyn, n, plans=full, seed=seed) for _ in range(3)]
    for e in envs:
        e.reset()
    o1, r1, d1 = envs[0].rollout(80)
    oa, ra, da = envs[1].rollout(split)
    ob, rb, db = envs[1].rollout(80 - split)
    assert torch.equal(o1, torch.cat([oa, ob])) and torch.equal(r1, torch.cat([ra, rb])) and torch.equal(d1, torch.cat([da, db]))
    for t in range(80):
        o, r, d = envs[2].step(auto_reset=True)
        assert torch.equal(o, o1[t]) and torch.equal(r, r1[t]) and torch.equal(d, d1[t])
    assert torch.equal(envs[0].environment_memory(), envs[1].environment_memory())
    assert torch.equal(envs[0].environment_memory(), envs[2].environment_memory())


@settings(max_examples=30, deadline=None, suppress_health_check=list(HealthCheck))
@given(kind=KIND, pool=st.integers(2, 300), seed=st.integers(0, 2**40), warm=st.integers(0, 40), waves=st.integers(1, 4),
       data=st.data())
def test_random_tree_edges_match_the_oracle(kind, pool, seed, warm, waves, data):
    """snac_transition on arbitrary pools: random sources (shared parents allowed), random distinct destinations disjoint
    from the sources of other edges (or equal to the edge's own source), explicit or counter-RNG step sizes."""
    import torch
    from snac_amd import BatchedDMPEnv

    dim, dyn = kind
    table = helpers.plan_table(dim, dyn, ("dense_train" if dim > 1 else "sin_train") if dyn else "p0")
    full = table.reshape((-1, 30) if dim == 1 else (-1, 26, 26))
    env = BatchedDMPEnv(dim, dyn, pool, plans=full, seed=seed)
    orc = helpers.oracle().OracleBatch(dim, dyn, pool, table, seed=seed)
    env.reset(); orc.reset()
    if warm:
        env.rollout(warm, obs=None); orc.rollout(warm, obs=None)
    rng = np.random.default_rng(seed % (2**32))
    for w in range(waves):
        n_dst = data.draw(st.integers(1, max(1, pool // 2)))
        dst = rng.choice(pool, n_dst, replace=False).astype(np.int32)
        free = np.setdiff1d(np.arange(pool), dst)                      # rows that are not written: legal sources for anyone
        inplace = rng.random(n_dst) < 0.3
        src = np.where(inplace | (len(free) == 0), dst, rng.choice(free if len(free) else dst, n_dst)).astype(np.int32)
        acts = rng.integers(0, env.num_actions, n_dst).astype(np.int8)
        ks = rng.integers(1, 4, n_dst).astype(np.int8) if data.draw(st.booleans()) else None
        o, r, d = env.transition(acts, ks, src, dst, t=w)
        oo, ro, do = orc.transition(acts, ks, src, dst, t=w)
        assert o.cpu().numpy().tobytes() == oo.tobytes() and r.cpu().numpy().tobytes() == ro.tobytes()
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), do)
    st_o = orc.state()
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(pool, -1), st_o["grid"])
    assert np.array_equal(env.count_step.cpu().numpy(), st_o["cs"]) and np.array_equal(env.count_brick.cpu().numpy(), st_o["cb"])
    assert env.iou().cpu().numpy().tobytes() == orc.iou().tobytes()
