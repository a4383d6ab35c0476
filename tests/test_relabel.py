"""Hindsight relabelling as the reference's DRQN_hindsight scripts do it (goldens: tests/golden/make_golden_relabel.py):
replay a finished episode against its own final grid as the plan, with the recorded actions and step sizes.
CPU: the oracle reproduces the relabelled rewards.  GPU: (a) the drop-in hindsight classes, driven exactly like the
scripts drive the reference (reset, overwrite .plan, step(action, step_size)); (b) snac_amd.hindsight.relabel_rewards
relabels all episodes of a dimension in one launch."""
import importlib
import os
import sys

import numpy as np
import pytest

import helpers

_Z = None


def _file():
    global _Z
    if _Z is None:
        _Z = np.load(os.path.join(helpers.GOLDEN, "relabel_static.npz"))
    return _Z


def _names():
    return _file()["cases"].tolist()


def _rec(name):
    z = _file()
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def _new_plan(dim, final_grid, orig_plan):
    g = final_grid.astype(np.float64)
    if dim == 1:
        return g[2:32].copy()
    p = np.array(orig_plan, np.float64).reshape(26, 26)
    p[3:23, 3:23] = g.reshape(26, 26)[3:23, 3:23]
    return p


@pytest.mark.parametrize("name", _names())
def test_oracle_relabels_like_the_reference(name):
    orc = helpers.oracle()
    rec = _rec(name)
    dim, pc = int(name[0]), int(name.split(".")[1][1:])
    orig = orc.static_plan(dim, pc)
    env = orc.OracleEnv(dim, False)
    env.reset(orig)
    assert env.e.tb == rec["total_brick"]
    new = _new_plan(dim, rec["final_grid"], orig).reshape(-1).astype(np.int32)
    for i, v in enumerate(new):                      # plan swapped AFTER reset(): total_brick stays the original one
        env.e.plan[i] = int(v)
    for t, (a, k) in enumerate(zip(rec["actions"], rec["step_size"])):
        _, r, d = env.step(int(a), int(k))
        assert r == rec["hindsight_reward"][t] and d == bool(rec["hindsight_done"][t]), (name, t)


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names())
def test_hindsight_class_relabel_flow_on_hip(name):
    mods = {1: ("1D", "DMP_Env_1D_static_hindsight_replay"), 2: ("2D", "DMP_Env_2D_static_hindsight_replay"),
            3: ("3D", "DMP_simulator_3d_static_circle_hindsight_replay")}
    rec = _rec(name)
    dim, pc = int(name[0]), int(name.split(".")[1][1:])
    sub, mod = mods[dim]
    path = os.path.join(helpers.ROOT, "snac_amd", "Env", sub)
    if path not in sys.path:
        sys.path.append(path)
    env = getattr(importlib.import_module(mod), "deep_mobile_printing_%dd1r_hindsight" % dim)(plan_choose=pc)
    for rep in range(2):                             # twice: the second reset() must restore the original plan first
        env.reset()
        assert env.total_brick == rec["total_brick"]
        g = rec["final_grid"].astype(np.float64)
        if dim == 1:
            env.plan = g[2:32]                                              # DRQN_hindsight_1D_static.py:243 (rebinding)
        else:
            env.plan[3:23, 3:23] = g.reshape(26, 26)[3:23, 3:23]           # DRQN_hindsight_2D_static.py:246 (in place)
            env.input_plan = env.plan[3:23, 3:23]
        for t, (a, k) in enumerate(zip(rec["actions"], rec["step_size"])):
            _, r, d = env.step(int(a), int(k))
            assert r == rec["hindsight_reward"][t] and d == bool(rec["hindsight_done"][t]), (name, rep, t)
    env.reset()                                      # and an un-relabelled replay gives the original rewards again
    for t, (a, k) in enumerate(zip(rec["actions"], rec["step_size"])):
        _, r, _ = env.step(int(a), int(k))
        assert r == rec["reward"][t], (name, t)


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 2, 3])
def test_batched_relabel_in_one_launch(dim):
    import torch
    from snac_amd.hindsight import relabel_rewards

    names = [n for n in _names() if n.startswith("%dd." % dim)]
    recs = [_rec(n) for n in names]
    T = max(len(r["actions"]) for r in recs)
    N = len(recs)
    A = np.zeros((T, N), np.int8)
    K = np.ones((T, N), np.int8)
    for i, r in enumerate(recs):
        A[:len(r["actions"]), i] = r["actions"]
        K[:len(r["actions"]), i] = r["step_size"]
    shape = (N, 34) if dim == 1 else (N, 26, 26)
    grids = np.stack([r["final_grid"].astype(np.float64) for r in recs]).reshape(shape)
    rew, done = relabel_rewards(dim, grids, [int(r["total_brick"]) for r in recs], torch.from_numpy(A), torch.from_numpy(K))
    rew, done = rew.cpu().numpy(), done.cpu().numpy()
    for i, r in enumerate(recs):
        L = len(r["actions"])
        assert np.array_equal(rew[:L, i], r["hindsight_reward"].astype(np.float32)), names[i]
        assert np.array_equal(done[:L, i].astype(np.uint8), r["hindsight_done"]), names[i]


def test_hindsight_module_has_no_host_round_trip():
    """the relabel path keeps grids, plans and rewards on the GPU -- no .cpu(), no numpy."""
    src = open(os.path.join(helpers.ROOT, "snac_amd", "hindsight.py")).read()
    code = src.split('"""', 2)[2]                     # the module body behind its docstring
    for word in (".cpu(", "numpy", "np.", ".item(", ".tolist("):
        assert word not in code, word


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 2, 3])
def test_relabel_of_a_device_batch_without_synchronising(dim):
    """relabel_batch: the finished episodes live in a BatchedDMPEnv (their final grids were imported into it here); the plan rows are
    written from its packed records by snac_plans_from_grids and the recorded actions rolled out -- with torch's sync debug mode
    set to "error", so any device-to-host copy or .item() in the path would raise.  Rewards / done equal the reference's."""
    import torch
    from snac_amd import BatchedDMPEnv
    from snac_amd.hindsight import relabel_batch

    names = [n for n in _names() if n.startswith("%dd." % dim)]
    recs = [_rec(n) for n in names]
    T = max(len(r["actions"]) for r in recs)
    N = len(recs)
    A = np.zeros((T, N), np.int8)
    K = np.ones((T, N), np.int8)
    for i, r in enumerate(recs):
        A[:len(r["actions"]), i] = r["actions"]
        K[:len(r["actions"]), i] = r["step_size"]
    shape = (N, 1, 34) if dim == 1 else (N, 26, 26)
    grids = np.stack([r["final_grid"].astype(np.float64) for r in recs]).reshape(shape)
    holder = BatchedDMPEnv(dim, False, N + 3, plan_choose=0)       # the batch the episodes "ended" in (+ rows that are not picked)
    holder.reset()
    pos = np.full((N, 2), 3) if dim != 1 else np.full((N,), 2)
    rows = np.arange(N)[::-1].copy() + 2                              # scattered over the holder, in reverse order
    holder.import_states(pos, np.zeros(N, np.int32), np.zeros(N, np.int32), grids, total_brick=np.array([int(r["total_brick"]) for r in recs]),
                         dst=rows)
    a_dev, k_dev = torch.from_numpy(A).cuda(), torch.from_numpy(K).cuda()
    rows_dev = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    mode = None
    try:
        mode = torch.cuda.get_sync_debug_mode()
        torch.cuda.set_sync_debug_mode("error")
    except Exception:
        mode = None
    try:
        rew, done = relabel_batch(holder, a_dev, k_dev, rows=rows_dev, dynamic_rules=False)
    finally:
        if mode is not None:
            torch.cuda.set_sync_debug_mode(mode)
    rew, done = rew.cpu().numpy(), done.cpu().numpy()
    for i, r in enumerate(recs):
        L = len(r["actions"])
        assert np.array_equal(rew[:L, i], r["hindsight_reward"].astype(np.float32)), names[i]
        assert np.array_equal(done[:L, i].astype(np.uint8), r["hindsight_done"]), names[i]
    # the same rows from environment_memory in the reference's format, as a device tensor
    from snac_amd.hindsight import relabel_rewards

    rew2, done2 = relabel_rewards(dim, torch.from_numpy(grids).cuda(), [int(r["total_brick"]) for r in recs], a_dev, k_dev)
    assert torch.equal(rew2.cpu(), torch.from_numpy(rew)) and np.array_equal(done2.cpu().numpy(), done)
