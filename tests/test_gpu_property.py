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
    assert np.array_equal(env.environment_memory().cpu().numpy().reshape(n, -1), st_["grid"].astype(np.float64))
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
