"""Counterpart of the reference driver `python multiprocess.py --env 2DDynamic --plan_type 0 --num_envs N`
(multiprocess.py:34-97) on the HIP path:

    python -m snac_amd.multiprocess --env 2DDynamic --plan_type 0 --num_envs 65536 [--fused]

Default: the reference's loop shape -- T = total_step vector steps, one VectorizedEnvWrapper.step per tick, uniform
random actions from np.random (over the env's own action_dim; --reference-actions reproduces the reference's
`np.random.randint(3, size=N)`), then the three shapes are printed.  --fused runs the same T ticks as one
snac_rollout launch (counter RNG, auto-reset) and prints env-steps/s.
"""
import argparse
import time

import numpy as np

ENVS = {"1DStatic": (1, False), "1DDynamic": (1, True), "2DStatic": (2, False), "2DDynamic": (2, True),
        "3DStatic": (3, False), "3DDynamic": (3, True)}


def make_plans(name, plan_type):
    from . import plans

    kind, dynamic = ENVS[name]
    if not dynamic:
        return kind, dynamic, plans.static_plan(kind, plan_type)[None]
    if kind == 1:
        return kind, dynamic, plans.dataset(1, "sin", "train")
    # the reference loads the 2-D dataset for 3DDynamic (multiprocess.py:76); the 3-D set is the intended one
    return kind, dynamic, plans.dataset(kind, ["dense", "sparse"][plan_type], "train")


def main(args):
    import torch

    from .vector import VectorizedEnvWrapper

    if args.env is None or args.env not in ENVS:
        print("please choose an environment in the list: {1DStatic, 1DDynamic,2DStatic, 2DDynamic, 3DStatic, 3DDynamic} ")
        return
    if args.plan_type is None and args.env != "1DDynamic":
        print("please choose a shape from list: {0: sin, 1:Gaussian, 2: Step}" if args.env == "1DStatic"
              else "please choose a shape from list: {0: Dense, 1: Sparse}")
        return
    env = VectorizedEnvWrapper(make_plans(args.env, args.plan_type or 0), num_envs=args.num_envs)
    T = env.envs[0].total_step
    observations = env.reset()
    if args.fused:
        b = env.batched
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        obs, rewards, dones = b.rollout(T, obs="last")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(tuple(obs.shape), tuple(rewards.shape), tuple(dones.shape))
        print("%.3e env-steps/s (%d envs x %d steps in %.3f ms)" % (args.num_envs * T / dt, args.num_envs, T, dt * 1e3))
        print(b.episodic_stats())
        return
    A = 3 if args.reference_actions else env.action_dim
    for t in range(T):
        actions = np.random.randint(A, size=args.num_envs)
        observations, rewards, dones = env.step(actions)
    print(observations.shape)
    print(rewards.shape)
    print(dones.shape)


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--env', type=str, default=None,
                        help='Environment Name: {1DStatic, 1DDynamic,2DStatic, 2DDynamic, 3DStatic, 3DDynamic}')
    parser.add_argument('--plan_type', type=int, default=None, help='type of shapes')
    parser.add_argument('--num_envs', type=int, default=3, help='Number of environments')
    parser.add_argument('--fused', action='store_true', help='one fused rollout launch instead of T step() calls')
    parser.add_argument('--reference-actions', action='store_true', help="actions = np.random.randint(3, size=N) as in the reference")
    main(parser.parse_args())
