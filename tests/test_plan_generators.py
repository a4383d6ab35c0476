"""Plan generators.

The reference draws random-triangle plans with cv2 (Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py:37-59) and ships
2000 of them as datasets (2D / 3D x dense / sparse x 400 + 50 + 50).  cv2 is absent here, its output is not: every dataset
plan must come out of the rasteriser bit for bit from the three vertices recovered in tests/golden/dataset_triangles.npz
(tests/golden/make_dataset_triangles.py) -- for the oracle (CPU) and for the HIP kernel (GPU).  Generated plans: HIP == oracle
bit for bit, the reference's acceptance rules hold, and rollouts on generated plan tables match the oracle."""
import os

import numpy as np
import pytest

import helpers

SETS = [(dim, dens, split) for dim in (2, 3) for dens in ("dense", "sparse") for split in ("train", "val", "test")]


def _tri():
    return np.load(os.path.join(helpers.GOLDEN, "dataset_triangles.npz"))


@pytest.mark.parametrize("dim,dens,split", SETS)
def test_oracle_rasteriser_reproduces_the_cv2_drawn_datasets(dim, dens, split):
    orc = helpers.oracle()
    plans = helpers.plans_npz()["%dd_%s_%s" % (dim, dens, split)]
    v = _tri()["%dd_%s_%s" % (dim, dens, split)]
    assert len(v) == len(plans) and (v >= 0).all() and (v < 20).all()      # a vertex triple was found for EVERY plan
    for p, q in zip(plans, v):
        img, area = orc.raster_triangle(q[:3], q[3:], dens == "sparse")
        want = (np.asarray(p).reshape(26, 26)[3:23, 3:23] != 0).astype(np.int32)
        assert np.array_equal(img, want) and area == want.sum()


@pytest.mark.parametrize("dim,sparse", [(1, 0), (2, 0), (2, 1), (3, 0), (3, 1)])
def test_oracle_generator_obeys_the_reference_rules(dim, sparse):
    orc = helpers.oracle()
    table, tb = orc.make_plans(dim, sparse, 5, 1000, 300)
    if dim == 1:
        # k1 in [3, 12) around 20: heights within 20 +- 12, total = sum
        assert table.min() >= 8 and table.max() <= 32 and np.array_equal(tb, table.sum(1))
        assert len({tuple(r) for r in table}) > 200                     # small amplitudes round to the same curve now and then
        return
    t = table.reshape(-1, 26, 26)
    cells = (t[:, 3:23, 3:23] != 0).sum((1, 2))
    assert (cells > (20 if sparse else 50)).all() and t.sum() == t[:, 3:23, 3:23].sum()       # area rule; frame stays zero
    assert set(np.unique(t)) <= ({0, 1} if dim == 2 else {0, 6})
    if dim == 3:
        assert (cells < 110).all() and np.array_equal(tb, 6 * cells)       # script/HumanPlayerGUI/env/Env3D.py:360-364
    else:
        assert np.array_equal(tb, np.maximum(cells, 30))


# ---- GPU ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dim,dens,split", SETS)
def test_hip_rasteriser_reproduces_the_cv2_drawn_datasets(dim, dens, split):
    from snac_amd import BatchedDMPEnv

    plans = helpers.plans_npz()["%dd_%s_%s" % (dim, dens, split)]
    v = _tri()["%dd_%s_%s" % (dim, dens, split)]
    P = len(plans)
    env = BatchedDMPEnv(dim, True, 4, plans=np.zeros((P, 26, 26)))
    area = env.generate_plans(0, P, sparse=dens == "sparse", vertices=v[:, [0, 3, 1, 4, 2, 5]]).cpu().numpy()
    env._sync_plans_full()
    assert np.array_equal(env.plans_full, np.asarray(plans, np.float64).reshape(P, 26, 26))
    assert np.array_equal(area, (np.asarray(plans).reshape(P, 26, 26) != 0).sum((1, 2)))
    tb = (np.asarray(plans).reshape(P, -1).sum(1)).astype(np.int64)
    assert np.array_equal(env._plan_tb.cpu().numpy().astype(np.int64), np.maximum(tb, 30) if dim == 2 else tb)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,sparse,P", [(1, 0, 5000), (2, 0, 20000), (2, 1, 20000), (3, 0, 20000), (3, 1, 20000)])
def test_generated_plans_hip_vs_oracle_and_rollouts_on_them(dim, sparse, P):
    import torch

    from snac_amd import BatchedDMPEnv

    orc = helpers.oracle()
    first, seed, base = 7, 99, 12345
    env = BatchedDMPEnv(dim, True, 512, plans=np.zeros((P, 30) if dim == 1 else (P, 26, 26)) + (20 if dim == 1 else 0), seed=3)
    area = env.generate_plans(first, P - first, sparse=bool(sparse), seed=seed, id_base=base).cpu().numpy()
    env._sync_plans_full()
    table, tb = orc.make_plans(dim, sparse, seed, base + first, P - first)
    got = env.plans_full[first:].reshape(P - first, -1)
    assert np.array_equal(got, table.astype(np.float64))
    assert np.array_equal(env._plan_tb.cpu().numpy()[first:].astype(np.int64), tb.astype(np.int64))
    cells = (table != 0).sum(1) if dim != 1 else table.sum(1)
    assert np.array_equal(area, cells)
    # the generated table is a plan table like any other: a rollout on it equals the oracle on the same table
    full = env.plans_full
    ob = orc.OracleBatch(dim, True, 512, full.reshape(P, -1).astype(np.int32), seed=3)
    # rows below `first` keep what the constructor packed
    assert env.reset().cpu().numpy().tobytes() == ob.reset().tobytes()
    og, rg, dg = env.rollout(300)
    oc, rc, dc = ob.rollout(300, nthreads=8)
    assert og.cpu().numpy().tobytes() == oc.tobytes() and rg.cpu().numpy().tobytes() == rc.tobytes()
    assert np.array_equal(dg.cpu().numpy().astype(np.uint8), dc) and env.iou().cpu().numpy().tobytes() == ob.iou().tobytes()
    assert torch.equal(env.total_brick.cpu(), torch.from_numpy(ob.state()["tb"].astype(np.int64)))


@pytest.mark.gpu
@pytest.mark.parametrize("dim,sparse", [(2, 0), (2, 1), (3, 0), (3, 1), (1, 0)])
def test_plans_from_numpys_stream_equal_the_reference_draw_order(dim, sparse):
    """generate_plans_numpy(): same np.random seed -> the plans the reference's create_plan() would draw, row after row (two
    randint(0, 20, size=3) per attempt, redrawn while the area fails the bounds; 1D: uniform, randint, uniform).  A RandomState twin
    that consumes the stream that way -- with the oracle's rasteriser deciding the redraws -- lands on the same rows, and on the same
    next word of the global stream."""
    from snac_amd import BatchedDMPEnv

    orc = helpers.oracle()
    P, seed = 24, 123 + 10 * dim + sparse
    env = BatchedDMPEnv(dim, True, 8, plans=np.zeros((P, 30) if dim == 1 else (P, 26, 26)) + (20 if dim == 1 else 0), seed=3)
    np.random.seed(seed)
    areas = env.generate_plans_numpy(2, P - 2, sparse=bool(sparse))
    after = int(np.random.randint(0, 1 << 30))
    env._sync_plans_full()
    st = np.random.RandomState(seed)
    for r in range(2, P):
        if dim == 1:
            k1 = st.uniform(3, 12); k2 = st.randint(1, 4); ph = st.uniform(-1, 1) * np.pi
            y = np.round(k1 * np.sin(2 * np.pi / 30 * (k2 * np.arange(30) + ph)) + 20)
            assert np.array_equal(env.plans_full[r], y) and areas[r - 2] == int(y.sum())
            continue
        while True:
            x, y = st.randint(0, 20, size=3), st.randint(0, 20, size=3)
            img, a = orc.raster_triangle(x, y, sparse)
            if (20 if sparse else 50) < a < (110 if dim == 3 else 401):
                break
        assert areas[r - 2] == a
        assert np.array_equal(env.plans_full[r][3:23, 3:23], img * (6 if dim == 3 else 1)), r
    assert int(st.randint(0, 1 << 30)) == after
    tb = env._plan_tb.cpu().numpy()[2:].astype(np.int64)
    want = np.asarray(areas) if dim == 1 else (np.maximum(np.asarray(areas), 30) if dim == 2 else 6 * np.asarray(areas))
    assert np.array_equal(tb, want)
