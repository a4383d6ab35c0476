"""A vectorised DQN loop on the batched API -- everything stays on the GPU: N envs step per tick (snac_step), transitions land
in the device replay ring (snac_rollout_rec via ReplayRing.collect is the random-policy prefill; the epsilon-greedy ticks
are appended with ReplayRing.append), minibatches come from snac_replay_gather.  The network and the update rule follow the
reference's 2D dynamic DQN (script/DQN/2d/DQN_2d_dynamic.py: observation 51 + plan 400 -> 5 Q values, target network,
replace every `replace` steps); this file is an illustration of the API, not part of the parity surface.

    python examples/dqn_batched.py --envs 4096 --ticks 200
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from snac_amd import BatchedDMPEnv, ReplayRing  # noqa: E402


class QNet(nn.Module):
    def __init__(self, obs_dim, plan_cells, actions, hidden=256):
        super().__init__()
        self.body = nn.Sequential(nn.Linear(obs_dim + plan_cells, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU(),
                                  nn.Linear(hidden, actions))

    def forward(self, obs, plan):
        return self.body(torch.cat([obs, plan.flatten(1)], dim=1))


def run(envs=4096, ticks=200, batch=2048, gamma=0.95, lr=1e-4, replace=50, prefill=64, seed=1, log=print):
    torch.manual_seed(seed)
    env = BatchedDMPEnv(2, True, envs, seed=seed, obs_dtype=torch.float32)
    dev = env.device
    obs = env.reset()
    ring = ReplayRing(env, capacity_ticks=max(2 * prefill, 128))
    ring.collect(prefill)                                          # random policy, one fused launch
    obs = env.observe()
    plan = env.input_plan().float()
    net, target = QNet(env.obs_dim, 400, env.num_actions).to(dev), QNet(env.obs_dim, 400, env.num_actions).to(dev)
    target.load_state_dict(net.state_dict())
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    t0, losses, returns = time.time(), [], []
    for tick in range(ticks):
        eps = max(0.05, 1.0 - tick / (0.6 * ticks))
        with torch.no_grad():
            greedy = net(obs.float(), plan).argmax(dim=1)
        explore = torch.rand(envs, device=dev) < eps
        actions = torch.where(explore, torch.randint(0, env.num_actions, (envs,), device=dev), greedy).to(torch.int8)
        ring.append(actions)                                       # one env tick, recorded in the ring (auto-reset)
        obs = env.observe()
        plan = env.input_plan().float()                            # resets change plans
        mb = ring.sample(batch)
        with torch.no_grad():
            q_next = target(mb["s_next"], mb["plan"]).max(dim=1).values
            y = mb["reward"] + gamma * q_next * (~mb["done"]).float()
        q = net(mb["s"], mb["plan"]).gather(1, mb["action"][:, None]).squeeze(1)
        loss = nn.functional.mse_loss(q, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if (tick + 1) % replace == 0:
            target.load_state_dict(net.state_dict())
            st = env.episodic_stats()
            returns.append(st["return_sum"] / max(st["episodes"], 1))
            log("tick %4d  eps %.2f  loss %.4f  episodes %d  mean return %.2f  mean IoU %.4f  (%.0f env-steps/s incl. learning)" % (
                tick + 1, eps, losses[-1], st["episodes"], returns[-1], st["iou_fx_sum"] / 2.0**40 / max(st["episodes"], 1),
                envs * (tick + 1) / (time.time() - t0)))
    return losses, returns


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--batch", type=int, default=2048)
    a = ap.parse_args()
    run(a.envs, a.ticks, a.batch)
