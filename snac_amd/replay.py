"""ReplayRing: the replay memory of the reference's DQN / DRQN scripts, kept on the GPU.

The reference appends one python tuple (s, a, r, s', env_plan) per env-step to a deque and rebuilds float32 minibatches
on the host (script/DQN/2d/DQN_2d_dynamic.py:122-124 store_memory, :184-199 prefill loop, :145-166 minibatch assembly).
Here the fused rollout writes its outputs straight into a ring over ticks,

    obs[cap, N, D]   reward[cap, N]   done[cap, N]   action[cap, N]   step_size[cap, N]   plan_idx[cap, N]   first[cap, N]

and a transition is addressed by (slot, env): s' = obs[slot, env]; s = obs[slot - 1, env], or the reset observation
when first[slot, env] (the reference takes prev_state from env.reset() there); plan = the plan row in effect.  Nothing
is stored twice and nothing crosses PCIe; sample() gathers float32 minibatches with snac_replay_gather.
"""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class ReplayRing:
    def __init__(self, env, capacity_ticks, layout="ticks", place_candidates=0, memory="malloc"):
        """env: BatchedDMPEnv (already reset); capacity_ticks: ring length in vector steps (>= 2).
        layout "ticks": obs[cap, N, D]; "tiled": obs[ceil(N / 64), cap, 64, D] -- a tile of 64 envs streams through its own
        contiguous region of the ring (SNAC_OBS_TILED: the faster layout to collect into); row(slot, env) and
        obs_at(slot) read either.  memory: where the observation ring lives -- "vmm": a snac_traj_alloc block (one virtual range
        over chunks from two 32 GiB slices of physical memory: MI355X writes it 15-20 % faster; building it
        holds up to half of the free device memory for about a second and freeing it waits for the device, so it is opt-in),
        "malloc" (default): torch.empty, "auto": "vmm" from 1 GiB up.  place_candidates > 1: the ring is placed by env.alloc_trajectory (that many
        candidate blocks timed with the rollout itself, the fastest kept: where the driver puts a block's runs still matters)."""
        if capacity_ticks < 2:
            raise ValueError("capacity_ticks must be >= 2")
        if layout not in ("ticks", "tiled"):
            raise ValueError("layout must be 'ticks' or 'tiled'")
        self.env = env
        self.cap = int(capacity_ticks)
        self.tiled = layout == "tiled"
        N, D, dev = env.num_envs, env.obs_dim, env.device
        self.placement = None
        if memory not in ("auto", "vmm", "malloc"):
            raise ValueError("memory must be 'auto', 'vmm' or 'malloc'")
        shape = ((N + 63) // 64, self.cap, 64, D) if self.tiled else (self.cap, N, D)
        nbytes = shape[0] * shape[1] * shape[2] * (shape[3] if self.tiled else 1) * torch.empty((), dtype=env.obs_dtype).element_size()
        if memory == "auto":
            memory = "vmm" if nbytes >= (1 << 30) else "malloc"
        if place_candidates and place_candidates > 1:
            self.obs, self.placement = env.alloc_trajectory(self.cap, candidates=int(place_candidates), layout=layout, memory=memory)
        elif memory == "vmm":
            from . import trajmem

            self.obs = trajmem.traj_empty(shape, env.obs_dtype, dev)
        else:
            self.obs = torch.empty(shape, dtype=env.obs_dtype, device=dev)
        self.memory = memory
        self.reward = torch.zeros((self.cap, N), dtype=torch.float32, device=dev)
        self.done = torch.zeros((self.cap, N), dtype=torch.uint8, device=dev)
        self.action = torch.zeros((self.cap, N), dtype=torch.int8, device=dev)
        self.step_size = torch.zeros((self.cap, N), dtype=torch.int8, device=dev)
        self.plan_idx = torch.zeros((self.cap, N), dtype=torch.int16, device=dev)
        self.first = torch.zeros((self.cap, N), dtype=torch.uint8, device=dev)
        self.head = 0          # next slot to write
        self.ticks = 0         # ticks collected so far
        self.plan_cells = 30 if env.kind == 1 else 400
        # the predecessor of slot 0 is slot cap - 1: seed it with the envs' CURRENT observation, so that the very first
        # transitions have their `s` also when the ring is attached to envs in mid-episode (first[0, i] == 0)
        self._put(self.cap - 1, env.observe())
        self._env_t = env.t    # the env may only advance through the ring: s' / s are paired by slot

    def _put(self, slot, rows):
        """rows [N, D] -> the ring's slot (either layout)"""
        if not self.tiled:
            self.obs[slot].copy_(rows)
            return
        N, D = rows.shape
        G = self.obs.shape[0]
        if N != G * 64:                                            # ragged last tile: pad (the padding rows are never read)
            rows = torch.cat([rows, rows.new_zeros((G * 64 - N, D))])
        self.obs[:, slot].copy_(rows.view(G, 64, D))

    def obs_at(self, slot):
        """The observations of one ring slot as [N, D] (a view for layout "ticks", a copy for "tiled")."""
        if not self.tiled:
            return self.obs[slot]
        G, _, E, D = self.obs.shape
        return self.obs[:, slot].reshape(G * E, D)[:self.env.num_envs]

    def row(self, slot, env_index):
        """obs rows of (slot, env) pairs -> [B, D] in the ring's dtype (either layout)."""
        slot = torch.as_tensor(slot, device=self.env.device).long()
        env_index = torch.as_tensor(env_index, device=self.env.device).long()
        if not self.tiled:
            return self.obs[slot, env_index]
        return self.obs[env_index >> 6, slot, env_index & 63]

    def __len__(self):
        """Number of addressable transitions."""
        return self.valid_ticks() * self.env.num_envs

    def valid_ticks(self):
        """Slots whose predecessor is still in the ring (the oldest slot is kept only as a predecessor once wrapped)."""
        return self.ticks if self.ticks < self.cap else self.cap - 1

    def collect(self, T, actions=None, step_size=None):
        """Run T vector steps (auto-reset) and append them: at most two launches (ring wrap).  actions / step_size as in
        BatchedDMPEnv.rollout ([T, N] or None for the counter RNG)."""
        T = int(T)
        if T > self.cap:
            raise ValueError("T exceeds the ring capacity")
        if self.env.t != self._env_t:
            raise _lib.SnacError("the envs were stepped outside the ring since the last collect(): the predecessor rows no longer "
                                 "match (attach a new ReplayRing)")
        done_ticks = 0
        while done_ticks < T:
            n = min(T - done_ticks, self.cap - self.head)
            sl = slice(self.head, self.head + n)
            a = None if actions is None else actions[done_ticks:done_ticks + n]
            k = None if step_size is None else step_size[done_ticks:done_ticks + n]
            rec = dict(actions=self.action[sl], step_size=self.step_size[sl], plan_idx=self.plan_idx[sl], first=self.first[sl])
            if self.tiled:
                self.env.rollout(n, actions=a, step_size=k, obs="tiled", out=self.obs, ring=(self.cap, self.head),
                                 reward_out=self.reward[sl], done_out=self.done[sl], record=rec)
            else:
                self.env.rollout(n, actions=a, step_size=k, obs="all", out=self.obs[sl], reward_out=self.reward[sl],
                                 done_out=self.done[sl], record=rec)
            self.head = (self.head + n) % self.cap
            self.ticks += n
            done_ticks += n
        self._env_t = self.env.t

    def append(self, actions, step_size=None):
        """One vector step with the caller's actions ([N], e.g. an epsilon-greedy policy's), appended to the ring."""
        actions = torch.as_tensor(actions, device=self.env.device)
        self.collect(1, actions=actions.reshape(1, -1), step_size=None if step_size is None else torch.as_tensor(step_size).reshape(1, -1))

    def slots(self):
        """Ring slots that hold addressable transitions, oldest first (int64 tensor on the device)."""
        v = self.valid_ticks()
        start = (self.head - v) % self.cap
        return (torch.arange(v, device=self.env.device) + start) % self.cap

    def gather(self, slot, env_index, with_plan=True):
        """Minibatch for explicit (slot, env) pairs -> dict of float32 / int tensors on the device:
        s [B, D], s_next [B, D], action [B], reward [B], done [B] (bool), plan [B, 20, 20] (1D: [B, 30])."""
        e = self.env
        slot = torch.as_tensor(slot, device=e.device).to(torch.int32).contiguous()
        env_index = torch.as_tensor(env_index, device=e.device).to(torch.int32).contiguous()
        B = int(slot.numel())
        s = torch.empty((B, e.obs_dim), dtype=torch.float32, device=e.device)
        s_next = torch.empty_like(s)
        plan = torch.empty((B, self.plan_cells), dtype=torch.float32, device=e.device) if with_plan else None
        with torch.cuda.device(e.device):
            fn = e._lib.snac_replay_gather_tiled if self.tiled else e._lib.snac_replay_gather
            _lib.check(fn(C.byref(e._desc), C.byref(e._state), self.cap, _ptr(self.obs), _ptr(self.first),
                          _ptr(self.plan_idx), _ptr(slot), _ptr(env_index), B, _ptr(s), _ptr(s_next), _ptr(plan), e._stream()))
        # action / reward / done of the samples: ONE flat index and three 1-D gathers (two-index advanced indexing was five index
        # kernels that took as long as the gather kernel itself, tools/gather_time.py)
        flat = slot.long() * e.num_envs + env_index.long()
        out = dict(s=s, s_next=s_next, action=self.action.view(-1)[flat].long(), reward=self.reward.view(-1)[flat],
                   done=self.done.view(-1)[flat].bool())
        if with_plan:
            out["plan"] = plan if e.kind == 1 else plan.view(B, 20, 20)
        return out

    def sample(self, batch, generator=None, with_plan=True):
        """Uniform minibatch over the addressable transitions (random.sample in the reference, :144)."""
        v = self.valid_ticks()
        if v == 0:
            raise ValueError("the ring is empty")
        dev = self.env.device
        age = torch.randint(0, v, (batch,), device=dev, generator=generator)
        slot = (self.head - 1 - age) % self.cap
        env_index = torch.randint(0, self.env.num_envs, (batch,), device=dev, generator=generator)
        return self.gather(slot, env_index, with_plan=with_plan)

    def sample_sequences(self, batch, time_step, generator=None, with_plan=True, oversample=4):
        """DRQN-style minibatch (Memory.get_batch of script/DRQN/2d/DRQN_2D_dynamic_training.py:131-143): `batch` windows of
        `time_step` consecutive transitions of one env that lie inside ONE episode -> dict of tensors
        s / s_next [B, L, D] float32, action / reward / done [B, L], plan [B, 20, 20] (1D: [B, 30]).
        Windows are drawn uniformly over all valid windows in the ring (the reference draws an episode first, then a
        window inside it, so it favours short episodes; neither is a parity surface -- the reference uses python's
        `random`).  Candidates are drawn `oversample` x batch at a time and filtered on the device."""
        L, B = int(time_step), int(batch)
        v = self.valid_ticks()
        if L < 1 or v < L:
            raise ValueError("time_step must be in [1, valid_ticks()]")
        dev, N = self.env.device, self.env.num_envs
        steps = torch.arange(L, device=dev)
        got_slot, got_env, have = [], [], 0
        for _ in range(64):
            n = max(B * oversample, 64)
            age = torch.randint(L - 1, v, (n,), device=dev, generator=generator)        # age of the window's FIRST transition
            env_index = torch.randint(0, N, (n,), device=dev, generator=generator)
            slots = (self.head - 1 - age[:, None] + steps[None, :]) % self.cap            # [n, L], oldest first
            inside = self.first[slots[:, 1:], env_index[:, None]].sum(dim=1) == 0 if L > 1 else torch.ones(n, dtype=torch.bool, device=dev)
            got_slot.append(slots[inside]); got_env.append(env_index[inside])
            have += int(inside.sum())
            if have >= B:
                break
        if have < B:
            raise ValueError("no episode in the ring holds %d consecutive steps" % L)
        slots, env_index = torch.cat(got_slot)[:B], torch.cat(got_env)[:B]
        flat = self.gather(slots.reshape(-1), env_index[:, None].expand(B, L).reshape(-1), with_plan=False)
        out = {k: t.view(B, L, *t.shape[1:]) for k, t in flat.items()}
        if with_plan:
            out["plan"] = self.gather(slots[:, 0], env_index, with_plan=True)["plan"]
        out["slot"], out["env"] = slots, env_index
        return out
