"""GPU: the observation-layout variants of the reference's env copies as flags of the HIP kernels:
  Env/2D/DMP_Env_2D_static_Lnet.py:61-76            frame cells 2, normalised scalars with a static plan   layout="lnet2d"
  Env/1D/DMP_Env_1D_static_Lnet.py:83               position appended (8 values)                          layout="lnet1d"
  script/PPO/*/DMP_*.py                             raw counters, dataset classes append the plan           layout="ppo"
  and the build's own 8-value record tail (reward, done, position, counters) that the single-env classes read back.
Checked (a) batched, N = 65 536, against the oracle configured the same way (reset, fused rollout, per-tick step, tree edges),
(b) by replaying the goldens recorded from the reference (traj_lnet.npz, traj_ppo.npz) through BatchedDMPEnv itself."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

TAIL = {"position": 1, "plan": 2, "record": 4}


def _mk(kind, dyn, n, seed=5, **kw):
    import torch

    from snac_amd import BatchedDMPEnv

    tag = "dense_train" if kind != 1 else "sin_train"
    table = helpers.plan_table(kind, dyn, tag if dyn else "p0")
    full = table.reshape((len(table), 30) if kind == 1 else (len(table), 26, 26))
    env = BatchedDMPEnv(kind, dyn, n, plans=full, seed=seed, **kw)
    orc = helpers.oracle().OracleBatch(kind, dyn, n, table, seed=seed)
    norm = {None: dyn, "raw": False, "norm": True}[env.obs_scalars]
    orc.configure(obs_norm=norm, frame=env.frame_value, tail=env.obs_tail)
    assert orc.obs_dim == env.obs_dim
    return env, orc, torch


CASES = [
    # kind, dynamic, N, T, kwargs, expected obs_dim
    (2, False, 65536, 6, dict(layout="lnet2d"), 51),
    (1, False, 65536, 9, dict(layout="lnet1d"), 8),
    (2, True, 65536, 2, dict(layout="ppo"), 451),
    (3, True, 16384, 3, dict(layout="ppo"), 451),
    (1, True, 65536, 5, dict(layout="ppo"), 37),
    (2, False, 65536, 4, dict(layout="ppo"), 51),
    (2, True, 65536, 5, dict(obs_tail=("record",)), 59),
    (3, True, 16384, 7, dict(obs_tail=("position", "record"), obs_scalars="raw"), 61),
    (1, False, 4099, 11, dict(obs_tail=("position", "plan", "record"), frame_value=2), 46),
    (2, True, 1000, 7, dict(obs_tail=("position", "plan", "record"), frame_value=2, obs_scalars="raw"), 461),
    (3, False, 777, 9, dict(obs_tail=("plan",), obs_scalars="norm"), 451),
]


@pytest.mark.parametrize("kind,dyn,n,T,kw,dim", CASES)
def test_layout_flags_batched_vs_oracle(kind, dyn, n, T, kw, dim):
    env, orc, torch = _mk(kind, dyn, n, **kw)
    assert env.obs_dim == dim
    assert env.reset().cpu().numpy().tobytes() == orc.reset().tobytes()
    # a few ticks first so that the rollout starts from a used state
    for t in range(2):
        og, rg, dg = env.step(auto_reset=True)
        oc, rc, dc = orc.step(t, auto_reset=True)
        assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
        assert np.array_equal(dg.cpu().numpy().astype(np.uint8), dc)
    og, rg, dg = env.rollout(T)
    oc, rc, dc = orc.rollout(T, t0=2, nthreads=8)
    assert tuple(og.shape) == (T, n, dim)
    assert og.cpu().numpy().tobytes() == oc.tobytes()
    assert rg.cpu().numpy().tobytes() == rc.tobytes() and np.array_equal(dg.cpu().numpy().astype(np.uint8), dc)
    want = oc[-1].copy()
    if env.obs_tail & 4:
        want[:, dim - 8] = 0.0                                   # a row written outside a step: record reward 0, done = pending reset
    assert env.observe().cpu().numpy().tobytes() == want.tobytes()
    # float32 observations of the same layout = (float) of the float64 row
    e32, o2, _ = _mk(kind, dyn, min(n, 2048), obs_dtype=torch.float32, **kw)
    e32.reset(), o2.reset()
    o32 = e32.rollout(3)[0].cpu().numpy()
    assert o32.tobytes() == o2.rollout(3)[0].astype(np.float32).tobytes()


@pytest.mark.parametrize("kind,dyn,kw", [(2, True, dict(layout="ppo")), (1, False, dict(layout="lnet1d")), (2, False, dict(layout="lnet2d")),
                                         (3, True, dict(obs_tail=("record", "position")))])
def test_layout_flags_on_tree_edges_and_explicit_inputs(kind, dyn, kw):
    n, m = 512, 2000
    env, orc, torch = _mk(kind, dyn, n, seed=3, **kw)
    env.reset(), orc.reset()
    rng = np.random.default_rng(4)
    A = env.num_actions
    a = rng.integers(0, A, (5, n)).astype(np.int8)
    k = rng.integers(1, 4, (5, n)).astype(np.int8)
    og, rg, dg = env.rollout(5, actions=a, step_size=k)
    oc, rc, dc = orc.rollout(5, actions=a, step_size=k)
    assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
    src = rng.integers(0, n // 2, m).astype(np.int32)
    dst = (n // 2 + rng.permutation(n // 2)[: min(m, n // 2)]).astype(np.int32)
    src, m = src[: len(dst)], len(dst)
    act = rng.integers(0, A, m).astype(np.int8)
    ks = rng.integers(1, 4, m).astype(np.int8)
    og, rg, dg = env.transition(act, ks, src=src, dst=dst)
    oc, rc, dc = orc.transition(act, ks, src=src, dst=dst)
    assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(dg.cpu().numpy().astype(np.uint8), dc)
    eq = env.obs_equal(og, og, idx_a=np.arange(m), idx_b=np.roll(np.arange(m), 1)).cpu().numpy()
    want = np.array([np.array_equal(oc[i], oc[i - 1]) for i in range(m)])
    assert np.array_equal(eq, want)


def test_scalar_step_and_reset_equal_the_array_forms():
    """snac_step_scalar / snac_reset_scalar: one action, step size and plan row by value for every env."""
    env, orc, torch = _mk(2, True, 300, obs_tail=("record",))
    o = env.reset_scalar(17).cpu().numpy()
    assert o.tobytes() == orc.reset(plan_idx=np.full(300, 17)).tobytes()
    for t, (a, k) in enumerate([(4, 1), (1, 3), (4, 2), (2, 2), (4, 1), (9, 2), (-3, 1), (0, 7)]):
        og = env.step_scalar(a, k).cpu().numpy()
        if 0 <= a < 5:
            oc, rc, dc = orc.step(t, np.full(300, a), np.full(300, min(max(k, 1), 3)))
            assert og.tobytes() == oc.tobytes()
            assert np.array_equal(og[:, 51], rc) and np.array_equal(og[:, 52], dc)
        else:                                                    # out of range: only count_step advances (the facade raises)
            assert np.array_equal(og[:, 56], np.full(300, t + 1))
            for e in range(300):
                orc.b.contents.envs[e].cs += 1
    with pytest.raises(Exception):
        env.reset_scalar(400)


@pytest.mark.parametrize("kind", [1, 2, 3])
def test_rows_written_into_page_locked_host_memory(kind):
    """new_host_obs(): the kernels store their rows straight into page-locked host memory (what the single-env classes read
    after one wait); the same steps into a device row must give the same bytes.  Unpinned or mis-shaped host tensors are refused."""
    import torch
    from snac_amd import BatchedDMPEnv

    n = 5
    a = BatchedDMPEnv(kind, True, n, seed=4, obs_tail=("record",))
    b = BatchedDMPEnv(kind, True, n, seed=4, obs_tail=("record",))
    host = a.new_host_obs()
    assert host.is_pinned() and host.device.type == "cpu"
    a.reset_scalar(11, out=host)
    a.sync()
    assert host.numpy().tobytes() == b.reset_scalar(11).cpu().numpy().tobytes()
    A = helpers.DIMS[kind]["A"]
    rng = np.random.default_rng(kind)
    for t in range(300):
        act, k = int(rng.integers(0, A)), int(rng.integers(1, 4))
        a.step_scalar(act, k, auto_reset=True, out=host)
        a.sync()
        assert host.numpy().tobytes() == b.step_scalar(act, k, auto_reset=True).cpu().numpy().tobytes(), t
    with pytest.raises(ValueError):
        a.step_scalar(0, 1, out=torch.empty((n, a.obs_dim), dtype=torch.float64))            # not page-locked
    with pytest.raises(ValueError):
        a.step_scalar(0, 1, out=torch.empty((n + 1, a.obs_dim), dtype=torch.float64, pin_memory=True))


# ---- the reference's own recordings through the BATCHED path --------------------------------------------------------------------
def _npz(name):
    return np.load(os.path.join(helpers.GOLDEN, name))


def _rec(z, case):
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(case + "/")}


def _replay_batched(env, rec, dim, reset_idx, tail_check=None):
    """One golden case on a 1-env batch: explicit actions / step sizes / plan rows, every observation row compared."""
    import torch

    starts = dict((int(s), e) for e, s in enumerate(rec["ep_start"]))
    S = len(rec["actions"])
    nw = rec["win"].shape[1] + 2
    for t in range(S):
        if t in starts:
            e = starts[t]
            o = env.reset(plan_idx=[reset_idx(e)]).cpu().numpy()[0]
            want = np.concatenate([rec["ep_reset_win"][e].astype(np.float64), rec["ep_reset_sc"][e]])
            assert o[:nw].tobytes() == want.tobytes()
            assert int(env.total_brick[0]) == rec["ep_total_brick"][e]
            cur = e
        o, r, d = env.step(torch.tensor([int(rec["actions"][t])], dtype=torch.int8), torch.tensor([int(rec["step_size"][t])], dtype=torch.int8))
        o = o.cpu().numpy()[0]
        want = np.concatenate([rec["win"][t].astype(np.float64), rec["sc"][t]])
        assert o[:nw].tobytes() == want.tobytes(), t
        assert float(r[0]) == rec["reward"][t] and bool(d[0]) == bool(rec["done"][t]), t
        if tail_check:
            tail_check(o[nw:], t, cur)
        if (t + 1) in starts or t == S - 1:
            e = starts[t + 1] - 1 if (t + 1) in starts else len(rec["ep_start"]) - 1
            assert np.array_equal(env.environment_memory()[0].cpu().numpy().reshape(-1), rec["ep_final_grid"][e].astype(np.float64))


def _lnet_cases():
    return _npz("traj_lnet.npz")["cases"].tolist()


@pytest.mark.parametrize("case", _lnet_cases())
def test_lnet_goldens_through_the_batched_path(case):
    from snac_amd import BatchedDMPEnv

    rec = _rec(_npz("traj_lnet.npz"), case)
    dim, pc = int(case[0]), int(case.split(".")[1][1:])
    if dim == 1:
        env = BatchedDMPEnv(1, False, 1, plan_choose=pc, layout="lnet1d")

        def check(tail, t, e):
            assert np.array_equal(tail, [rec["pos"][t][0]])      # Env/1D/DMP_Env_1D_static_Lnet.py:83: the position column
    elif dim == 2:
        env = BatchedDMPEnv(2, False, 1, plan_choose=pc, layout="lnet2d")
        check = None
    else:   # 3D L-Net: the dynamic class's rules on a static plan, total_step 1300, canonical layout
        from snac_amd import plans

        env = BatchedDMPEnv(3, True, 1, plans=plans.static_plan(3, pc)[None], total_step=1300)
        check = None
    _replay_batched(env, rec, dim, lambda e: 0, check)
    if dim == 2:
        assert (rec["win"] == 2).any() and not (rec["win"] == -1).any()      # the recording really holds the frame value 2


def _ppo_cases():
    return _npz("traj_ppo.npz")["cases"].tolist()


@pytest.mark.parametrize("case", _ppo_cases())
def test_ppo_goldens_through_the_batched_path(case):
    from snac_amd import BatchedDMPEnv

    rules = {"1d_static": (0, 0), "1d_dynamic": (1, 0), "2d_static": (1, 0), "2d_dynamic": (0, 0), "3d_static": (1, 1), "3d_dynamic": (0, 0)}
    rec = _rec(_npz("traj_ppo.npz"), case)
    fork, plan = case.split(".")[0], case.split(".")[1]
    dim, dyn = int(fork[0]), fork.endswith("dynamic")
    bg, tg = rules[fork]
    if dyn:
        dens, split = plan.split("-")
        env = BatchedDMPEnv(dim, True, 1, density=dens, split=split, layout="ppo", brick_gt=bool(bg), time_gt=bool(tg))
    else:
        env = BatchedDMPEnv(dim, False, 1, plan_choose=int(plan), layout="ppo", brick_gt=bool(bg), time_gt=bool(tg))
    assert env.obs_dim == {1: 7, 2: 51, 3: 51}[dim] + ((30 if dim == 1 else 400) if dyn else 0)

    def check(tail, t, e):
        if dyn:
            p = rec["ep_plan"][e].astype(np.float64)
            p = p if dim == 1 else p.reshape(26, 26)[3:23, 3:23].reshape(-1)
            assert np.array_equal(tail, p)

    _replay_batched(env, rec, dim, (lambda e: int(rec["ep_plan_idx"][e])) if dyn else (lambda e: 0), check)
