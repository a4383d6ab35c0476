"""Batched hindsight relabelling, entirely on the device.

The reference's DRQN_hindsight scripts replay a finished episode on a second env whose plan has been overwritten with
the episode's own final grid, feeding back the recorded actions and step sizes, and keep the new rewards
(script/DRQN_hindsight/2d/DRQN_hindsight_2D_dynamic.py:270-282; env side: step(action, step_size) of
Env/*/DMP_*_hindsight_replay*.py).  total_brick of that second env is the one reset() computed from the ORIGINAL plan,
because the plan is swapped after reset().  Here N episodes are relabelled by one kernel + one rollout launch: the plan
table of a scratch batch receives one row per episode (its final grid, snac_plans_from_grids) with the episode's
total_brick, env i is reset onto row i, and the recorded actions are rolled out without observations.  No grid, plan or
reward crosses the bus: inputs that already live on the GPU (a BatchedDMPEnv whose episodes just ended, device tensors)
are read where they are; host arrays are uploaded once.
"""
import torch

from .batched import BatchedDMPEnv

_MAX_ROWS = 32767                                                   # plan rows are int16: episodes per scratch batch


def _relabel_chunk(kind, dynamic_rules, device, total_step, m, actions, step_size, fill):
    env = BatchedDMPEnv(kind, bool(dynamic_rules), m, empty_plans=m, device=device, total_step=total_step)
    fill(env)
    env.reset(plan_idx=torch.arange(m, device=env.device, dtype=torch.int16), want_obs=False, check=False)
    _, reward, done = env.rollout(int(actions.shape[0]), actions=actions, step_size=step_size, obs=None)
    return reward, done


def relabel_rewards(kind, final_grids, total_brick, actions, step_size, device="cuda", total_step=None, dynamic_rules=False):
    """final_grids: [N, 34] / [N, 26, 26] environment_memory of the N finished episodes (frame values are ignored), a device
    tensor or a host array; total_brick: [N] total_brick of the episodes' original plans; actions / step_size: int [T, N] as
    recorded (entries after an episode's end are ignored by the caller).  Only the rewards are returned, float32 [T, N] on the
    device; step t of episode i is meaningful up to and including its done step.
    dynamic_rules: the 3D *_usedata hindsight env (post-build boxed-in test, -100, total_step 1000); no effect in 1D / 2D.
    Returns (reward [T, N], done [T, N] bool)."""
    kind = int(kind)
    dev = torch.device(device)
    mem = torch.as_tensor(final_grids).to(dev, torch.float64)
    N = int(mem.shape[0])
    tb = torch.as_tensor(total_brick).to(dev, torch.int32).reshape(N)
    a = torch.as_tensor(actions).to(dev, torch.int8)
    k = torch.as_tensor(step_size).to(dev, torch.int8)
    rs, ds = [], []
    for lo in range(0, N, _MAX_ROWS):
        hi = min(N, lo + _MAX_ROWS)
        r, d = _relabel_chunk(kind, dynamic_rules, dev, total_step, hi - lo, a[:, lo:hi].contiguous(), k[:, lo:hi].contiguous(),
                              lambda env: env.plans_from_grids(environment_memory=mem[lo:hi], total_brick=tb[lo:hi]))
        rs.append(r), ds.append(d)
    return (rs[0], ds[0]) if len(rs) == 1 else (torch.cat(rs, dim=1), torch.cat(ds, dim=1))


def relabel_batch(env, actions, step_size, rows=None, total_brick=None, dynamic_rules=None):
    """The same for episodes that just ended in a BatchedDMPEnv: env's CURRENT grids are the final grids (rows int[m] picks the
    envs, None: all), each episode's total_brick is the one in its header unless given.  actions / step_size: int8 [T, m] device
    tensors as recorded by rollout(record=...).  Returns (reward [T, m] float32, done [T, m] bool) on the device."""
    m_all = env.num_envs if rows is None else int(torch.as_tensor(rows).numel())
    ri = None if rows is None else torch.as_tensor(rows, device=env.device).to(torch.int32)
    dyn = env.dynamic if dynamic_rules is None else dynamic_rules
    rs, ds = [], []
    for lo in range(0, m_all, _MAX_ROWS):
        hi = min(m_all, lo + _MAX_ROWS)
        sel = ri[lo:hi] if ri is not None else (None if (lo, hi) == (0, env.num_envs) else torch.arange(lo, hi, device=env.device, dtype=torch.int32))
        tbc = None if total_brick is None else torch.as_tensor(total_brick, device=env.device)[lo:hi]
        r, d = _relabel_chunk(env.kind, dyn, env.device, env.total_step, hi - lo, actions[:, lo:hi].contiguous(), step_size[:, lo:hi].contiguous(),
                              lambda e: e.plans_from_grids(src=env, rows=sel, total_brick=tbc))
        rs.append(r), ds.append(d)
    return (rs[0], ds[0]) if len(rs) == 1 else (torch.cat(rs, dim=1), torch.cat(ds, dim=1))
