"""VectorizedEnvWrapper: the batching loop of the reference (multiprocess.py:15-32) on the HIP path.

The reference wrapper aliases ONE env object N times (multiprocess.py:19), so its "vector step" is N sequential
steps of a single env; its dynamic modes also fail on numpy >= 1.24 (ragged np.asarray).  This wrapper implements
the intended semantics -- N independent envs -- with the same call surface and return shapes:

    reset()            -> observations  (N, 1, D) float64
    reset_at(i)        -> observation of env i (1, D)
    step(actions)      -> (observations (N, 1, D), rewards (N,), dones (N,))      no auto-reset, like the reference

Host randomness is consumed as N sequential reference envs would consume it: one np.random.randint(1, 4) per env
per step (drawn as one size-N array: numpy fills arrays in element order from the same stream) and one
np.random.randint(0, len(dataset)) per env on a dynamic reset.  `wrapper.batched` is the underlying
BatchedDMPEnv for callers that want device tensors, the counter RNG or fused rollouts.
"""
import numpy as np

from .batched import BatchedDMPEnv
from .envs import _draw_step_size


class VectorizedEnvWrapper:
    HOST_OBS_MAX = 1024

    def __init__(self, env_, num_envs=1, device="cuda", obs_dtype=None):
        """env_: one of the snac_amd.envs facades (its kind, plan set and plan mode are copied), or a
        (kind, dynamic, plans) tuple."""
        import torch

        self.env = env_
        self.num_envs = int(num_envs)
        if isinstance(env_, tuple):
            kind, dynamic, plans = env_
            self._random = True
        else:
            kind, dynamic, plans = env_._dim, env_._dynamic, env_._table
            self._random = getattr(env_, "random_choose_paln", True)
        self._seq = 0
        self.batched = BatchedDMPEnv(kind, dynamic, self.num_envs, plans=plans, device=device,
                                     obs_dtype=obs_dtype or torch.float64)
        self.envs = [self.batched] * self.num_envs        # len(wrapper.envs) and envs[0].total_step keep working
        # numpy in, numpy out without copy commands: actions, step sizes, rewards and done flags live in page-locked host
        # memory that the kernel reads / writes over the bus; so do the observation rows of a small batch (up to HOST_OBS_MAX
        # envs the kernel's own stores beat a DMA copy, tools/hostrow_time.py), a large batch copies them from the device
        n = self.num_envs
        pin = dict(pin_memory=True)
        self._a, self._k = torch.empty(n, dtype=torch.int8, **pin), torch.empty(n, dtype=torch.int8, **pin)
        self._r, self._d = torch.empty(n, dtype=torch.float32, **pin), torch.empty(n, dtype=torch.uint8, **pin)
        self._o = self.batched.new_host_obs() if n <= self.HOST_OBS_MAX else self.batched._new_obs()
        self._a_np, self._k_np, self._r_np, self._d_np = self._a.numpy(), self._k.numpy(), self._r.numpy(), self._d.numpy()
        self._out = (self._o, self._r, self._d)                  # ONE tuple object: step()'s fast path recognises the call before by identity
        # Up to 256 envs -- the reference driver's default is --num_envs 3 (multiprocess.py:96) -- step through the batch's MAILBOX: up to
        # four resident wavefronts (an env per lane, 64 envs per wave) poll a doorbell in host memory, so a vector step is a store and a
        # spin instead of a launch and a stream wait (snac_mailbox_step_n; 24 -> 9 us per vector step at 3 envs).  SNAC_MAILBOX=0 keeps
        # the launch path, SNAC_MAILBOX_MAX_ENVS moves the limit (round 5: 64, one wave).
        import os

        self._mrows = None
        if n <= min(256, int(os.environ.get("SNAC_MAILBOX_MAX_ENVS", "256"))) and os.environ.get("SNAC_MAILBOX", "1") != "0":
            try:
                self._mrows = self.batched.mailbox_open().numpy()
                self._mr, self._md = self.batched.mailbox_outputs()
            except Exception:
                self._mrows = None
        self._o_np = self._o.numpy() if self._o.device.type == "cpu" else None
        self.action_dim = self.batched.num_actions
        self.total_step = self.batched.total_step

    def _plan_indices(self, n):
        b = self.batched
        if not b.dynamic:
            return np.zeros(n, np.int16)
        if self._random:
            return np.random.randint(0, b.num_plans, size=n).astype(np.int16)
        idx = (self._seq + np.arange(n)) % b.num_plans
        self._seq = int((self._seq + n) % b.num_plans)
        return idx.astype(np.int16)

    def reset(self):
        obs = self.batched.reset(plan_idx=self._plan_indices(self.num_envs))
        return obs.cpu().numpy().reshape(self.num_envs, 1, -1)

    def reset_at(self, env_index):
        mask = np.zeros(self.num_envs, np.uint8)
        mask[env_index] = 1
        pidx = np.zeros(self.num_envs, np.int16)
        pidx[env_index] = self._plan_indices(1)[0]
        obs = self.batched.reset(mask=mask, plan_idx=pidx)
        return obs[env_index].cpu().numpy().reshape(1, -1)

    def step(self, actions):
        actions = np.asarray(actions)
        n = self.num_envs
        if actions.shape != (n,):
            raise ValueError("actions must have shape (%d,)" % n)
        if n <= 8:                                               # a handful of envs: python scalars are cheaper than numpy calls
            al = actions.tolist()
            if min(al) < 0 or max(al) >= self.action_dim:
                raise ValueError("action outside [0, %d)" % self.action_dim)
            k = self._k_np
            for i in range(n):                                   # np.random.randint(1, 4, size=n): the same words of the global stream, in element order
                k[i] = _draw_step_size()
        else:
            if actions.dtype == np.int64 and actions.flags.c_contiguous:
                bad = int(actions.view(np.uint64).max()) >= self.action_dim   # one reduction: a negative value is a huge unsigned one
            else:
                bad = actions.min() < 0 or actions.max() >= self.action_dim
            if bad:
                raise ValueError("action outside [0, %d)" % self.action_dim)
            self._k_np[:] = np.random.randint(1, 4, size=n)
        self._a_np[:] = actions
        b = self.batched
        if self._mrows is not None:                              # doorbell + acknowledgement of the batch's resident wave
            b.mailbox_step_n(self._a_np, self._k_np)
            return self._mrows.reshape(self.num_envs, 1, -1).copy(), self._mr.astype(np.float64), self._md.astype(bool)
        b.step(self._a, self._k, out=self._out)
        if self._o.device.type == "cpu":                         # one launch, one wait
            b.sync()
            obs = self._o_np.copy()
        else:                                                    # the copy is ordered behind the kernel on the same stream
            obs = self._o.cpu().numpy()
            b.sync()
        return obs.reshape(self.num_envs, 1, -1), self._r_np.astype(np.float64), self._d_np.astype(bool)
