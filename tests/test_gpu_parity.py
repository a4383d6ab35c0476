"""GPU parity: the HIP path (through the C ABI, via snac_amd.BatchedDMPEnv) against the CPU oracle and
against the golden trajectories recorded from the reference.  Bit-exact: integer / index work, and the
float64 observation scalars are single IEEE divisions (compared as raw bytes)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

KINDS = [(1, False), (1, True), (2, False), (2, True), (3, False), (3, True)]


def _ids(v):
    return "%dd_%s" % (v[0], "dyn" if v[1] else "sta") if isinstance(v, tuple) else str(v)


def _table(dim, dyn):
    if dyn:
        return helpers.plan_table(dim, True, "sin_train" if dim == 1 else "dense_train")
    return helpers.plan_table(dim, False, "p0")


def _full(dim, table):
    return table.reshape(len(table), 30) if dim == 1 else table.reshape(len(table), 26, 26)


def _make(dim, dyn, n, seed=1, base=0, obs_dtype=None, table=None):
    import torch
    from snac_amd import BatchedDMPEnv

    table = _table(dim, dyn) if table is None else table
    env = BatchedDMPEnv(dim, dyn, n, plans=_full(dim, table), seed=seed, env_id_base=base,
                        obs_dtype=obs_dtype or torch.float64)
    orc = helpers.oracle().OracleBatch(dim, dyn, n, table, seed=seed, env_id_base=base)
    return env, orc


def _same_bits(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def _check_state(env, orc):
    st = orc.state()
    H, W = (1, 34) if env.kind == 1 else (26, 26)
    mem = env.environment_memory().cpu().numpy()
    assert np.array_equal(mem.reshape(env.num_envs, -1), st["grid"])
    pos = env.position.cpu().numpy()
    assert np.array_equal(pos[:, 0], st["pos"][:, 0])
    if env.kind != 1:
        assert np.array_equal(pos[:, 1], st["pos"][:, 1])
    assert np.array_equal(env.count_brick.cpu().numpy(), st["cb"])
    assert np.array_equal(env.count_step.cpu().numpy(), st["cs"])
    assert np.array_equal(env.total_brick.cpu().numpy(), st["tb"])
    assert np.array_equal(env.plan_idx.cpu().numpy(), st["plan_idx"])
    assert np.array_equal(env.episode.cpu().numpy(), st["episode"])
    assert np.array_equal(env.need_reset.cpu().numpy().astype(np.uint8), st["need_reset"])
    assert np.array_equal(env.episode_return.cpu().numpy(), st["ep_return"])
    assert _same_bits(env.iou().cpu().numpy(), orc.iou())
    s = orc.stats()
    e = env.episodic_stats()
    assert (e["episodes"], e["return_sum"], e["iou_fx_sum"]) == (int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum()))


@pytest.mark.parametrize("kind", KINDS, ids=_ids)
def test_step_explicit_inputs_vs_oracle(kind):
    """step() with explicit actions and step sizes (the parity mode), auto-reset on."""
    import torch

    dim, dyn = kind
    n, S = 192, 700 if dim != 3 else 400
    env, orc = _make(dim, dyn, n, seed=7, base=1000)
    rng = np.random.default_rng(dim * 10 + dyn)
    A = helpers.DIMS[dim]["A"]
    # per-env action mix so that drops / builds dominate in some envs and episodes end by count_brick
    p = rng.dirichlet(np.ones(A) * 0.6, size=n)
    cum = np.cumsum(p, axis=1)
    o_g = env.reset().cpu().numpy()
    o_c = orc.reset()
    assert _same_bits(o_g, o_c)
    for t in range(S):
        a = (rng.random(n)[:, None] > cum).sum(axis=1).clip(0, A - 1).astype(np.int8)
        k = rng.integers(1, 4, size=n).astype(np.int8)
        og, rg, dg = env.step(torch.from_numpy(a), torch.from_numpy(k), auto_reset=True)
        oc, rc, dc = orc.step(t, a, k, auto_reset=True)
        assert _same_bits(og.cpu().numpy(), oc), (kind, t)
        assert _same_bits(rg.cpu().numpy(), rc), (kind, t)
        assert np.array_equal(dg.cpu().numpy().astype(np.uint8), dc), (kind, t)
    _check_state(env, orc)
    assert _same_bits(env.observe().cpu().numpy(), _observe_all(orc))


def _observe_all(orc):
    """Current observation of every oracle env (orc_observe), [n, obs_dim]."""
    L = helpers.oracle().lib()
    out = np.zeros((orc.n, orc.obs_dim))
    for i in range(orc.n):
        L.orc_observe(orc.b.contents.envs[i], out[i].ctypes.data)
    return out


@pytest.mark.parametrize("kind", KINDS, ids=_ids)
def test_rollout_counter_rng_vs_oracle(kind):
    """rollout(): T fused steps, actions / step sizes / plan indices from the counter RNG."""
    dim, dyn = kind
    n = 320
    T = helpers.DIMS[dim]["T"][1 if dyn else 0]
    env, orc = _make(dim, dyn, n, seed=3, base=5_000_000_000)
    assert _same_bits(env.reset().cpu().numpy(), orc.reset())
    og, rg, dg = env.rollout(T)
    oc, rc, dc = orc.rollout(T, nthreads=8)
    assert _same_bits(og.cpu().numpy(), oc)
    assert _same_bits(rg.cpu().numpy(), rc)
    assert np.array_equal(dg.cpu().numpy().astype(np.uint8), dc)
    _check_state(env, orc)
    # a second launch continues the same streams (tick t0 = T) and equals stepping one by one
    og, rg, dg = env.rollout(37, obs="last")
    oc, rc, dc = orc.rollout(37, t0=T, obs="last", nthreads=8)
    assert _same_bits(og.cpu().numpy(), oc) and _same_bits(rg.cpu().numpy(), rc)
    for t in range(5):
        og, rg, dg = env.step(auto_reset=True)
        oc, rc, dc = orc.step(T + 37 + t, auto_reset=True)
        assert _same_bits(og.cpu().numpy(), oc) and _same_bits(rg.cpu().numpy(), rc)
    _check_state(env, orc)


@pytest.mark.parametrize("kind", KINDS, ids=_ids)
def test_rollout_explicit_streams_and_f32_obs(kind):
    import torch

    dim, dyn = kind
    n, T = 96, 150
    env, orc = _make(dim, dyn, n, seed=11, obs_dtype=torch.float32)
    rng = np.random.default_rng(99)
    A = helpers.DIMS[dim]["A"]
    a = rng.integers(0, A, size=(T, n)).astype(np.int8)
    k = rng.integers(1, 4, size=(T, n)).astype(np.int8)
    env.reset()
    orc.reset()
    og, rg, dg = env.rollout(T, actions=torch.from_numpy(a), step_size=torch.from_numpy(k))
    oc, rc, dc = orc.rollout(T, actions=a, step_size=k)
    assert og.dtype == torch.float32
    assert _same_bits(og.cpu().numpy(), oc.astype(np.float32))   # f32 obs := (float) of the f64 value
    assert _same_bits(rg.cpu().numpy(), rc)
    assert np.array_equal(dg.cpu().numpy().astype(np.uint8), dc)


@pytest.mark.parametrize("kind", KINDS, ids=_ids)
def test_golden_replay_through_hip(kind):
    """Every golden trajectory of this env type as one env of a batch: explicit actions, the step sizes the
    reference drew, the plan indices it picked; compared with what the reference returned."""
    import torch
    from snac_amd import BatchedDMPEnv

    dim, dyn = kind
    z = helpers.traj_file(dim, dyn)
    names = z["cases"].tolist()
    recs = [helpers.load_case(dim, dyn, nm) for nm in names]
    # one plan table holding every plan set used by the cases, with per-case row offsets
    tags = []
    for nm in names:
        tag = nm.split(".")[0]
        if tag not in tags:
            tags.append(tag)
    tables = [helpers.plan_table(dim, dyn, t) for t in tags]
    offs = np.cumsum([0] + [len(t) for t in tables])
    table = np.concatenate(tables)
    case_off = [int(offs[tags.index(nm.split(".")[0])]) for nm in names]
    n = len(names)
    S = len(recs[0]["actions"])
    env = BatchedDMPEnv(dim, dyn, n, plans=_full(dim, table))
    A = torch.from_numpy(np.stack([r["actions"] for r in recs], axis=1)).to(env.device)
    K = torch.from_numpy(np.stack([r["step_size"] for r in recs], axis=1)).to(env.device)
    starts = [dict((int(s), e) for e, s in enumerate(r["ep_start"])) for r in recs]
    obs_all = torch.empty((S, n, env.obs_dim), dtype=torch.float64, device=env.device)
    rew_all = torch.empty((S, n), dtype=torch.float32, device=env.device)
    done_all = torch.empty((S, n), dtype=torch.bool, device=env.device)
    iou_at = {}
    for t in range(S):
        mask = np.zeros(n, np.uint8)
        pidx = np.zeros(n, np.int16)
        for i in range(n):
            if t in starts[i]:
                mask[i] = 1
                pidx[i] = case_off[i] + max(int(recs[i]["ep_plan_idx"][starts[i][t]]), 0)
        if mask.any():
            o = env.reset(mask, pidx).cpu().numpy()
            for i in np.nonzero(mask)[0]:
                e = starts[i][t]
                want = np.concatenate([recs[i]["ep_reset_win"][e].astype(np.float64), recs[i]["ep_reset_sc"][e]])
                assert o[i].tobytes() == want.tobytes()
                assert int(env.total_brick[i]) == recs[i]["ep_total_brick"][e]
        obs_all[t], rew_all[t], done_all[t] = env.step(A[t], K[t])
        ends = [i for i in range(n) if (t + 1) in starts[i] or t == S - 1]
        if ends:
            iou = env.iou().cpu().numpy()
            mem = env.environment_memory().cpu().numpy().reshape(n, -1)
            for i in ends:
                e = starts[i][t + 1] - 1 if (t + 1) in starts[i] else len(recs[i]["ep_start"]) - 1
                assert np.float64(iou[i]).tobytes() == np.float64(recs[i]["ep_iou"][e]).tobytes(), (names[i], e)
                assert np.array_equal(mem[i], recs[i]["ep_final_grid"][e].astype(np.float64)), (names[i], e)
    obs_all, rew_all, done_all = obs_all.cpu().numpy(), rew_all.cpu().numpy(), done_all.cpu().numpy()
    for i, r in enumerate(recs):
        want = np.concatenate([r["win"].astype(np.float64), r["sc"]], axis=1)
        assert obs_all[:, i].tobytes() == want.tobytes(), names[i]
        assert np.array_equal(rew_all[:, i], r["reward"].astype(np.float32)), names[i]
        assert np.array_equal(done_all[:, i].astype(np.uint8), r["done"]), names[i]


def test_sharding_is_invisible():
    """Two shards with env_id_base 0 / n produce the rows of one 2n batch (counter RNG keyed by global id)."""
    n, T = 128, 300
    whole, _ = _make(2, True, 2 * n, seed=5)
    lo, _ = _make(2, True, n, seed=5, base=0)
    hi, _ = _make(2, True, n, seed=5, base=n)
    for e in (whole, lo, hi):
        e.reset()
    ow, rw, dw = whole.rollout(T)
    ol, rl, dl = lo.rollout(T)
    oh, rh, dh = hi.rollout(T)
    assert _same_bits(ow[:, :n].cpu().numpy(), ol.cpu().numpy()) and _same_bits(ow[:, n:].cpu().numpy(), oh.cpu().numpy())
    a, b, c = whole.episodic_stats(), lo.episodic_stats(), hi.episodic_stats()
    assert all(a[k] == b[k] + c[k] for k in a)


@pytest.mark.parametrize("dyn", [False, True], ids=["sta", "dyn"])
def test_3d_plan_table_paths(dyn):
    """3D with plan tables that are not the reference's {0, 6} pattern: arbitrary heights, and another common height."""
    rng = np.random.default_rng(5)
    general = np.zeros((37, 26, 26), np.int32)
    general[:, 3:23, 3:23] = rng.integers(0, 4, size=(37, 20, 20)) * rng.integers(0, 2, size=(37, 20, 20))   # heights 0..3
    general[:, 5, 5] = 3
    binary9 = np.zeros((21, 26, 26), np.int32)
    binary9[:, 3:23, 3:23] = 9 * (rng.random((21, 20, 20)) < 0.4)
    binary9[:, 4, 4] = 9
    for full in (general, binary9):
        table = full.reshape(len(full), -1)
        env, orc = _make(3, dyn, 150, seed=12, table=table)
        assert _same_bits(env.reset().cpu().numpy(), orc.reset())
        t0 = 0
        for T in (3, 400, 5, 120):
            og, rg, dg = env.rollout(T)
            oc, rc, dc = orc.rollout(T, t0=t0)
            assert _same_bits(og.cpu().numpy(), oc) and _same_bits(rg.cpu().numpy(), rc)
            assert np.array_equal(dg.cpu().numpy().astype(np.uint8), dc)
            t0 += T
        _check_state(env, orc)


@pytest.mark.parametrize("kind", KINDS, ids=_ids)
@pytest.mark.parametrize("n", [1, 63, 64, 200, 16400, 65600])
def test_tiled_trajectory_layout_holds_the_same_rows(kind, n):
    """rollout(obs="tiled"): [ceil(N / 64), T, 64, D], row (t, env) at [env // 64, t, env % 64] -- the same observations as
    obs="all", only laid out tile-major (every tile size of the kernels, ragged last tiles, explicit inputs too)."""
    import torch
    from snac_amd import BatchedDMPEnv

    dim, dyn = kind
    if dim == 3 and n > 20000:
        n = 16400 + 8
    full = _full(dim, _table(dim, dyn))
    a = BatchedDMPEnv(dim, dyn, n, plans=full, seed=8)
    b = BatchedDMPEnv(dim, dyn, n, plans=full, seed=8)
    a.reset()
    b.reset()
    for T, explicit in ((23, False), (9, True)):
        acts = ks = None
        if explicit:
            g = torch.Generator().manual_seed(n)
            acts = torch.randint(0, helpers.DIMS[dim]["A"], (T, n), generator=g).to(torch.int8)
            ks = torch.randint(1, 4, (T, n), generator=g).to(torch.int8)
        oa, ra, da = a.rollout(T, actions=acts, step_size=ks)
        ot, rt, dt = b.rollout(T, actions=acts, step_size=ks, obs="tiled")
        assert tuple(ot.shape) == ((n + 63) // 64, T, 64, a.obs_dim)
        assert torch.equal(b.untile(ot), oa) and torch.equal(ra, rt) and torch.equal(da, dt)
