"""GPU: 2D node pools with one record per node (snac_nodes2d_pack / _unpack / snac_transition_nodes2d, snac_amd.NodePool2D) against
the batch-layout path (snac_transition on a BatchedDMPEnv pool -- itself oracle-checked in tests/test_gpu_mcts.py) AND against the CPU
oracle directly: the same search-shaped workload on both, every wave's rows, rewards and done flags equal, the states equal afterwards."""
import ctypes as C

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _tag(dyn):
    return "dense_train" if dyn else "p0"


def _pools(dyn, pool, seed, dtype=None):
    import torch
    from snac_amd import BatchedDMPEnv, NodePool2D

    table = helpers.plan_table(2, dyn, _tag(dyn))
    full = table.reshape(-1, 26, 26)
    kw = {"obs_dtype": dtype} if dtype is not None else {}
    env = BatchedDMPEnv(2, dyn, pool, plans=full, seed=seed, **kw)
    twin = BatchedDMPEnv(2, dyn, pool, plans=full, seed=seed, **kw)
    orc = helpers.oracle().OracleBatch(2, dyn, pool, table, seed=seed)
    env.reset(); twin.reset(); orc.reset()
    env.rollout(37, obs=None); twin.rollout(37, obs=None); orc.rollout(37, obs=None)
    nodes = NodePool2D(env, pool)
    assert nodes.load() == pool
    return env, twin, orc, nodes, torch


def _same_records(nodes, twin, rows=None):
    import torch

    rows = torch.arange(twin.num_envs, device=twin.device) if rows is None else torch.as_tensor(rows, device=twin.device)
    r = nodes.records[rows]
    assert torch.equal(r[:, :4].contiguous().view(torch.uint8), twin._hdr[rows].contiguous().view(torch.uint8).reshape(len(rows), 16))
    assert torch.equal(r[:, 4], twin._episode[rows].to(torch.int32))
    assert torch.equal(r[:, 8:28].contiguous().view(torch.uint8), twin._grid[rows].contiguous().view(torch.uint8).reshape(len(rows), 80))
    assert int(r[:, 5:8].abs().sum()) == 0 and int(r[:, 28:].abs().sum()) == 0


@pytest.mark.parametrize("dyn", [False, True])
def test_pack_unpack_round_trip_and_decoded_fields(dyn):
    env, twin, orc, nodes, torch = _pools(dyn, 1000, 5)
    _same_records(nodes, env)
    assert torch.equal(nodes.position, env.position.to(nodes.position.dtype)) and torch.equal(nodes.count_brick, env.count_brick.to(torch.int32))
    assert torch.equal(nodes.count_step, env.count_step.to(torch.int32)) and torch.equal(nodes.plan_idx, env.plan_idx.to(torch.int32))
    assert torch.equal(nodes.total_brick, env.total_brick.to(torch.int32)) and torch.equal(nodes.need_reset, env.need_reset)
    # gathered rows both ways
    rng = np.random.default_rng(1)
    rows = rng.permutation(1000)[:300].astype(np.int32)
    nrows = rng.permutation(1000)[:300].astype(np.int32)
    nodes.load(rows=rows, node_rows=nrows)
    got = nodes.records[torch.as_tensor(nrows.astype(np.int64), device=env.device)]
    assert torch.equal(got[:, 8:28].contiguous().view(torch.uint8), env._grid[torch.as_tensor(rows.astype(np.int64), device=env.device)].contiguous().view(torch.uint8).reshape(300, 80))
    other = type(env)(2, dyn, 1000, plans=env.plans_full, seed=99)
    other.reset()
    nodes.load()                                                     # records = env's rows again
    nodes.store(env=other)
    assert torch.equal(other._hdr, env._hdr) and torch.equal(other._grid, env._grid) and torch.equal(other._episode, env._episode)


@pytest.mark.parametrize("dyn,f32", [(False, False), (True, False), (True, True)])
def test_edges_on_node_records_equal_the_batch_pool_and_the_oracle(dyn, f32):
    import torch

    pool, roots = 6000, 500
    env, twin, orc, nodes, torch = _pools(dyn, pool, 11, torch.float32 if f32 else None)
    rng = np.random.default_rng(7)
    A, used = 5, roots
    cast = (lambda x: x.astype(np.float32)) if f32 else (lambda x: x)
    for wave, m in enumerate([256, 64, 1000, 4, 3, 130, 1, 777, 2048]):     # whole tiles, ragged tiles, m % 4 != 0 (rows value by value)
        if wave % 3 == 2:
            m = min(m, used)
        parents = rng.integers(0, used, m).astype(np.int32)                 # children share parents
        if wave % 3 == 2:                                                   # in place on distinct rows
            parents = rng.permutation(used)[:m].astype(np.int32)
            children = parents.copy()
        else:
            children = (used + np.arange(m)).astype(np.int32)
            used += m
        acts = rng.integers(0, A, m).astype(np.int8)
        ks = rng.integers(1, 4, m).astype(np.int8) if wave % 2 == 0 else None   # None: the counter RNG keyed by (edge, t)
        o1, r1, d1 = nodes.transition(acts, ks, src=parents, dst=children, t=wave)
        o2, r2, d2 = twin.transition(torch.from_numpy(acts), None if ks is None else torch.from_numpy(ks), src=parents, dst=children, t=wave)
        assert env._lib.snac_last_kernel() in (b"k_edges2d", b"k_transition2d", b"k_transition")   # the twin's launch came last
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), (wave, m)
        oo, ro, do = orc.transition(acts, ks, src=parents, dst=children, t=wave)
        assert o1.cpu().numpy().tobytes() == cast(oo).tobytes() and r1.cpu().numpy().tobytes() == ro.tobytes(), (wave, m)
        assert np.array_equal(d1.cpu().numpy().astype(np.uint8), do), (wave, m)
        _same_records(nodes, twin, np.unique(children).astype(np.int64))
    _same_records(nodes, twin)
    # back into a batch: evaluate / observe work on the unpacked rows
    nodes.store()
    assert torch.equal(env._hdr, twin._hdr) and torch.equal(env._grid, twin._grid) and torch.equal(env.observe(), twin.observe())


def test_no_indices_unaligned_rows_and_argument_errors():
    import torch
    from snac_amd import BatchedDMPEnv, NodePool2D, SnacError, _lib

    env, twin, orc, nodes, torch = _pools(True, 512, 3)
    acts = torch.randint(0, 5, (512,), dtype=torch.int8, device="cuda")
    ks = torch.randint(1, 4, (512,), dtype=torch.int8, device="cuda")
    o1, r1, d1 = nodes.transition(acts, ks)                          # identity rows
    o2, r2, d2 = twin.transition(acts, ks)
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    _same_records(nodes, twin)
    # an observation buffer that is not 16-byte aligned: rows value by value
    L = env._lib
    raw = torch.zeros(512 * 51 + 1, dtype=torch.float64, device="cuda")
    ob = raw[1:].view(512, 51)
    rw, dn = torch.empty(512, dtype=torch.float32, device="cuda"), torch.empty(512, dtype=torch.uint8, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(L.snac_transition_nodes2d(C.byref(env._desc), C.byref(env._state), vp(nodes.records), 512, 512, None, None, 9, vp(acts), vp(ks), vp(ob),
                                         vp(rw), vp(dn), env._stream()))
    o2, r2, d2 = twin.transition(acts, ks, t=9)
    assert ob.data_ptr() % 16 != 0 and torch.equal(ob, o2) and torch.equal(rw, r2) and torch.equal(dn.view(torch.bool), d2)
    # errors: a misaligned pool, another kind, too many edges, clashing rows
    assert L.snac_transition_nodes2d(C.byref(env._desc), C.byref(env._state), C.c_void_p(nodes.records.data_ptr() + 64), 511, 4, None, None, 0, vp(acts), vp(ks),
                                     None, None, None, env._stream()) != 0 and b"128-byte" in L.snac_last_error()
    assert L.snac_transition_nodes2d(C.byref(env._desc), C.byref(env._state), vp(nodes.records), 512, 513, None, None, 0, vp(acts), vp(ks), None, None, None,
                                     env._stream()) != 0
    e3 = BatchedDMPEnv(3, True, 8, seed=1)
    with pytest.raises(ValueError):
        NodePool2D(e3, 8)
    with pytest.raises(ValueError):
        nodes.transition(acts[:4], ks[:4], src=[0, 1, 2, 3], dst=[1, 9, 10, 11])       # record 1 is read by another edge
    with pytest.raises(ValueError):
        nodes.transition(acts[:4], ks[:4], src=[0, 1, 2, 3], dst=[9, 9, 10, 11])
    with pytest.raises(ValueError):
        nodes.transition(acts[:4], ks[:4], src=[0, 1, 2, 600], dst=[9, 8, 10, 11])


def test_full_size_wave_on_a_million_records_equals_the_batch_pool():
    """The bench's shape (transition_2d_nodes_524288_edges): a 2^20-record pool, 524 288 random-parent edges into fresh records -- at that
    size the oracle is out of reach, so the same wave runs on the batch pool (k_edges2d, oracle-checked at small sizes) and every row,
    reward, done flag and resulting record must be equal; pack -> unpack over the whole pool is the identity."""
    import torch
    from snac_amd import BatchedDMPEnv, NodePool2D

    pool, m = 1 << 20, 524288
    env = BatchedDMPEnv(2, True, pool, seed=1)
    env.reset()
    env.rollout(20, obs=None)
    twin = env.fork(torch.arange(pool, device=env.device))
    nodes = NodePool2D(env, pool)
    assert nodes.load() == pool
    g = torch.Generator(device="cuda").manual_seed(3)
    src = torch.randint(0, pool - m, (m,), device="cuda", dtype=torch.int32, generator=g)
    dst = (pool - m + torch.arange(m, device="cuda", dtype=torch.int32)).contiguous()
    acts = torch.randint(0, 5, (m,), device="cuda", generator=g).to(torch.int8)
    ks = torch.randint(1, 4, (m,), device="cuda", generator=g).to(torch.int8)
    o1, r1, d1 = nodes.transition(acts, ks, src=src, dst=dst, check=False)
    assert env._lib.snac_last_kernel() == b"k_edges2dp"
    o2, r2, d2 = twin.transition(acts, ks, src=src, dst=dst)
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    other = BatchedDMPEnv(2, True, pool, seed=2)
    other.reset()
    assert nodes.store(env=other) == pool
    assert torch.equal(other._hdr, twin._hdr) and torch.equal(other._grid, twin._grid) and torch.equal(other._episode, twin._episode)
    # a second wave from the children (in place), counter-RNG step sizes
    o1, r1, d1 = nodes.transition(acts, None, src=dst, dst=dst, t=7, check=False)
    o2, r2, d2 = twin.transition(acts, None, src=dst, dst=dst, t=7)
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
