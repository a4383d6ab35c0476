"""Multi-GPU: envs are independent, so a node runs one process per GPU, each with a contiguous shard of the global
env ids (weak scaling, no data-path collective).  The only exchange is one all-reduce of three int64 episodic sums
per rollout -- RCCL over xGMI on GPUs (backend "nccl" is RCCL on ROCm), gloo in the CPU tests.  The sums are
integers (returns are integer, IoU is accumulated as llrint(iou * 2^40)), so the reduced result is exact and
independent of the number of GPUs.
"""
import os

FX = float(2 ** 40)


def shard(total_envs, rank, world):
    """Contiguous shard of global env ids [0, total_envs): returns (num_local, env_id_base)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(total_envs), world)
    n = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return n, start


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun). Returns (rank, world, local)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def all_reduce_stats(stats, group=None):
    """stats: int64 tensor [3] = [episodes, return_sum, iou_fx_sum] (BatchedDMPEnv.stats_tensor()).  Summed over the
    ranks in place when a process group is initialised; returned unchanged otherwise."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def episodic_means(stats):
    """-> dict(episodes, mean_return, mean_iou) from the (reduced) sums."""
    e, r, i = (int(x) for x in stats.tolist())
    if e == 0:
        return dict(episodes=0, mean_return=None, mean_iou=None)
    return dict(episodes=e, mean_return=r / e, mean_iou=i / FX / e)


def make_sharded_env(kind, dynamic, total_envs, rank=None, world=None, **kw):
    """BatchedDMPEnv holding this rank's shard of `total_envs` global envs."""
    from .batched import BatchedDMPEnv

    if rank is None:
        rank = int(os.environ.get("RANK", "0"))
    if world is None:
        world = int(os.environ.get("WORLD_SIZE", "1"))
    n, base = shard(total_envs, rank, world)
    return BatchedDMPEnv(kind, dynamic, n, env_id_base=base, **kw)
