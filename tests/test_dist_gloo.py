"""N > 1 path on CPU: two gloo ranks, each holding a shard of the global env ids (snac_amd.dist.shard), reduce
their episodic sums with snac_amd.dist.all_reduce_stats.  The per-shard numbers come from the CPU oracle (this is
a test), the sharding / reduction logic is the product's.  The GPU twin (env_id_base on the HIP path) is
tests/test_gpu_parity.py::test_sharding_is_invisible."""
import os
import socket

import numpy as np
import pytest

import helpers

TOTAL, T, SEED = 42, 400, 9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from snac_amd import dist as sdist

    r, w, _ = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    n, base = sdist.shard(TOTAL, rank, world)
    table = helpers.plan_table(2, True, "dense_train")
    orc = helpers.oracle().OracleBatch(2, True, n, table, seed=SEED, env_id_base=base)
    orc.reset()
    obs, rew, done = orc.rollout(T, obs="last")
    s = orc.stats()
    local = torch.tensor([int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum())], dtype=torch.int64)
    red = sdist.all_reduce_stats(local.clone())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), obs=obs, rew=rew, local=local.numpy(), red=red.numpy(), base=base, n=n)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_stats_reduce_to_the_unsharded_result(world, tmp_path):
    import torch.multiprocessing as mp

    from snac_amd import dist as sdist

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    table = helpers.plan_table(2, True, "dense_train")
    whole = helpers.oracle().OracleBatch(2, True, TOTAL, table, seed=SEED)
    whole.reset()
    obs, rew, _ = whole.rollout(T, obs="last")
    s = whole.stats()
    want = np.array([int(s["episodes"].sum()), int(s["ret"].sum()), int(s["iou_fx"].sum())])
    assert want[0] > 0
    acc = np.zeros(3, np.int64)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(z["red"], want)                       # every rank holds the global sums
        acc += z["local"]
        b, n = int(z["base"]), int(z["n"])
        assert z["obs"].tobytes() == obs[b:b + n].tobytes()         # shard rows == rows of the unsharded batch
        assert z["rew"].tobytes() == np.ascontiguousarray(rew[:, b:b + n]).tobytes()
    assert np.array_equal(acc, want)
    import torch

    m = sdist.episodic_means(torch.from_numpy(want))
    assert m["episodes"] == want[0] and abs(m["mean_iou"] - want[2] / 2.0 ** 40 / want[0]) < 1e-15


def test_all_reduce_is_identity_without_a_process_group():
    import torch

    from snac_amd import dist as sdist

    t = torch.tensor([3, -7, 11], dtype=torch.int64)
    assert torch.equal(sdist.all_reduce_stats(t.clone()), t)
    assert sdist.episodic_means(torch.zeros(3, dtype=torch.int64))["mean_iou"] is None
