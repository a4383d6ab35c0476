"""Batched hindsight relabelling.

The reference's DRQN_hindsight scripts replay a finished episode on a second env whose plan has been overwritten with
the episode's own final grid, feeding back the recorded actions and step sizes, and keep the new rewards
(script/DRQN_hindsight/2d/DRQN_hindsight_2D_dynamic.py:270-282; env side: step(action, step_size) of
Env/*/DMP_*_hindsight_replay*.py).  total_brick of that second env is the one reset() computed from the ORIGINAL plan,
because the plan is swapped after reset().  Here N episodes are relabelled in one rollout launch: the plan table holds
one row per episode (its final grid) with the original total_brick.
"""
import numpy as np
import torch

from .batched import BatchedDMPEnv


def relabel_rewards(kind, final_grids, total_brick, actions, step_size, device="cuda", total_step=None, dynamic_rules=False):
    """final_grids: [N, 34] / [N, 26, 26] environment_memory of the N finished episodes (frame values are ignored);
    total_brick: [N] total_brick of the episodes' original plans; actions / step_size: int [T, N] as recorded (entries
    after an episode's end are ignored by the caller).  Static-class observation scalars are irrelevant here: only the
    rewards are returned, float32 [T, N]; step t of episode i is meaningful up to and including its done step.
    Returns (reward [T, N], done [T, N] bool)."""
    kind = int(kind)
    g = np.asarray(final_grids.cpu().numpy() if torch.is_tensor(final_grids) else final_grids, np.float64)
    N = len(g)
    if N > 32767:
        raise ValueError("at most 32767 episodes per call (one plan row each)")
    if kind == 1:
        plans = np.clip(g.reshape(N, -1)[:, 2:32], 0, None)
    else:
        plans = np.zeros((N, 26, 26))
        plans[:, 3:23, 3:23] = np.clip(g.reshape(N, 26, 26)[:, 3:23, 3:23], 0, None)
    # dynamic_rules: the 3D *_usedata hindsight env (post-build boxed-in test, -100, total_step 1000); no effect in 1D / 2D
    env = BatchedDMPEnv(kind, bool(dynamic_rules), N, plans=plans, plan_tb=np.asarray(total_brick), device=device,
                        total_step=total_step)
    env.reset(plan_idx=np.arange(N))
    T = int(actions.shape[0])
    _, reward, done = env.rollout(T, actions=actions, step_size=step_size, obs=None)
    return reward, done
