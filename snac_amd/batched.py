"""BatchedDMPEnv: N independent mobile-construction envs resident in MI355X HBM.

Host-side Python over PyTorch-ROCm tensors (device memory + streams only); every transition, observation,
reward and IoU is computed by the HIP kernels behind the C ABI of include/snac_hip.h.  The semantics per env
are those of the reference classes (file:line in include/snac_hip.h); the batched additions are documented
there too: explicit or counter-RNG step sizes / actions / plan indices, and auto-reset.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import plans as _plans

_KINDS = {1: 1, 2: 2, 3: 3, "1d": 1, "2d": 2, "3d": 3, "1D": 1, "2D": 2, "3D": 3}
_GRID_DTYPE = {1: torch.int16, 2: torch.int32, 3: torch.int16}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


# the per-tick fast path of step(): torch's raw accessors (no Stream / device objects are built)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_current_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _meta(*tensors):
    """(shape, stride, dtype) of every tensor (None stays None): what step()'s fast path compares besides the data pointers."""
    return tuple(None if t is None else (t.shape, t.stride(), t.dtype) for t in tensors)


class BatchedDMPEnv:
    """N envs of one kind.

    kind      1 | 2 | 3 (or "1d" ...)
    dynamic   False: the static-plan classes (obs scalars count_brick, count_step);
              True: the *_usedata_plan classes (count_brick/total_brick, count_step/total_step)
    plans     [P, 30] / [P, 26, 26] array of full plans as the reference stores them; default: the static plan
              `plan_choose` (static) or the converted training set of `density` (dynamic)
    seed      counter-RNG seed (include/snac_hip.h); env_id_base: global id of local env 0 (multi-GPU shards)
    brick_gt / time_gt   the strict termination tests of the env copies under script/PPO (SNAC_RULE_* in snac_hip.h):
              done when count_brick > total_brick / count_step > total_step instead of >=
    empty_plans   P: a device plan table of P empty rows (total_brick 1) that plans_from_grids() / generate_plans() fill on the
              device -- nothing is packed or uploaded by the host (the hindsight relabel path)
    layout    observation layout of the reference's env copies, produced by the same kernels (snac_env_desc.frame_value /
              obs_scalars / obs_tail): None = the canonical classes; "lnet1d" (Env/1D/DMP_Env_1D_static_Lnet.py: position
              appended, 8 values), "lnet2d" (Env/2D/DMP_Env_2D_static_Lnet.py: frame cells 2, normalised scalars), "ppo"
              (script/PPO/*: raw counters; the dataset classes append the plan: 37 / 451 values).  Or set the three fields
              directly: frame_value (-1 | 2), obs_scalars ("raw" | "norm" | None), obs_tail (iterable of "position", "plan",
              "record", or the bit set).
    """

    LAYOUTS = {None: {}, "lnet1d": dict(obs_tail=("position",)), "lnet2d": dict(frame_value=2, obs_scalars="norm"),
               "ppo": dict(obs_scalars="raw"), "ppo_plan": dict(obs_scalars="raw", obs_tail=("plan",))}
    _TAILS = {"position": _lib.TAIL_POSITION, "plan": _lib.TAIL_PLAN, "record": _lib.TAIL_RECORD}

    def __init__(self, kind, dynamic, num_envs, plans=None, plan_choose=0, density="dense", split="train",
                 device="cuda", seed=1, obs_dtype=torch.float64, env_id_base=0, total_step=None, plan_tb=None,
                 brick_gt=False, time_gt=False, layout=None, frame_value=None, obs_scalars=None, obs_tail=None, static_plan=0,
                 empty_plans=None):
        if not torch.cuda.is_available():
            raise _lib.SnacError("BatchedDMPEnv needs a ROCm GPU: there is no CPU fallback")
        self.kind = _KINDS[kind]
        self.dynamic = bool(dynamic)
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SnacError("device must be a cuda (ROCm) device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if obs_dtype not in (torch.float64, torch.float32):
            raise ValueError("obs_dtype must be torch.float64 or torch.float32")
        self.obs_dtype = obs_dtype
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)
        self._lib = _lib.lib()
        sz = _lib.env_sizes(self.kind, self.dynamic)
        self.sizes = sz
        self.num_actions = sz.num_actions
        if layout == "ppo" and dynamic:
            layout = "ppo_plan"                                  # the dataset copies under script/PPO append the plan
        lay = dict(self.LAYOUTS[layout])
        for key, val in (("frame_value", frame_value), ("obs_scalars", obs_scalars), ("obs_tail", obs_tail)):
            if val is not None:
                lay[key] = val
        self.frame_value = int(lay.get("frame_value", -1))
        self.obs_scalars = lay.get("obs_scalars")
        tail = lay.get("obs_tail", 0)
        if not isinstance(tail, int):
            tail = sum(self._TAILS[t] for t in set(tail))
        self.obs_tail = int(tail)
        self.total_step = int(total_step) if total_step else sz.total_step   # override: the 3D L-Net variant (1300)
        dev, N = self.device, self.num_envs
        if empty_plans is not None:
            P = int(empty_plans)
            if plans is not None or not 1 <= P <= 32767:
                raise ValueError("empty_plans: 1 .. 32767 rows, without `plans`")
            self.num_plans = P
            self.plans_full = None                                 # decoded from the device table when someone asks (_sync_plans_full)
            self._plans_stale = True
            self._plans = torch.zeros((P, sz.plan_elems), dtype=torch.int32 if self.kind == 2 else torch.int16, device=dev)
            self._plan_tb = torch.ones((P,), dtype=torch.int16, device=dev)
        else:
            if plans is None:
                if self.dynamic:
                    plans = _plans.dataset(self.kind, density, split)
                else:
                    plans = _plans.static_plan(self.kind, plan_choose)[None]
            self.plans_full = np.array(plans, np.float64)          # own copy: set_plan_row() edits it
            packed, tb = _plans.pack_plans(self.kind, self.plans_full)
            if plan_tb is not None:                                # caller-supplied total_brick per plan row (hindsight relabel)
                tb = np.asarray(plan_tb).astype(np.int16)
                if tb.shape != (len(packed),) or tb.min() < 1:
                    raise ValueError("plan_tb must hold one positive total_brick per plan")
            self.num_plans = len(packed)
            self._plans = torch.from_numpy(packed.view(np.int32) if self.kind == 2 else packed).to(dev)
            self._plan_tb = torch.from_numpy(tb).to(dev)
        self._hdr = torch.zeros((N, 4), dtype=torch.int32, device=dev)
        self._episode = torch.full((N,), -1, dtype=torch.int32, device=dev)
        self._grid = torch.zeros((N, sz.grid_elems), dtype=_GRID_DTYPE[self.kind], device=dev)
        self._stats = torch.zeros((3, N), dtype=torch.int64, device=dev)
        self._desc = _lib.EnvDesc(self.kind, int(self.dynamic), N, self.num_plans,
                                  _lib.OBS_F64 if obs_dtype == torch.float64 else _lib.OBS_F32, int(static_plan),
                                  self.seed & 0xFFFFFFFFFFFFFFFF, self.env_id_base, self.total_step,
                                  (_lib.RULE_BRICK_GT if brick_gt else 0) | (_lib.RULE_TIME_GT if time_gt else 0),
                                  self.frame_value,
                                  {None: _lib.SCALARS_DEFAULT, "raw": _lib.SCALARS_RAW, "norm": _lib.SCALARS_NORM}[self.obs_scalars],
                                  self.obs_tail, 0)
        self.obs_dim = self._lib.snac_obs_dim(C.byref(self._desc))   # values per observation row, tail included
        if self.obs_dim < 0:
            _lib.check(self.obs_dim)
        self.base_obs_dim = sz.obs_dim
        self.brick_gt, self.time_gt = bool(brick_gt), bool(time_gt)
        self.static_plan = int(static_plan)
        self._state = _lib.State(self._hdr.data_ptr(), self._episode.data_ptr(), self._grid.data_ptr(),
                                 self._plans.data_ptr(), self._plan_tb.data_ptr(), self._stats[0].data_ptr(),
                                 self._stats[1].data_ptr(), self._stats[2].data_ptr())
        self.t = 0  # tick: number of vector steps taken (keys the counter RNG)
        self._was_reset = False
        self._fast = None                                            # step(): the argument set validated by the previous call
        self._table_version, self._table_snapshot = 0, None          # state_dict(): one clone of the plan table per version of it
        self._dev_index = self.device.index
        self._desc_ref, self._state_ref = C.byref(self._desc), C.byref(self._state)
        self._snac_step = self._lib.snac_step
        self._snac_step_scalar, self._snac_sync = self._lib.snac_step_scalar, self._lib.snac_stream_sync
        self._ssf = None                                             # step_scalar_wait(): the host row validated by the call before
        self._host_ok = None                                         # the new_host_obs() row validated last
        self._mapped = {}                                            # page-locked host tensors seen by _is_mapped
        self._mb, self._mb_dirty, self._mb_final = None, False, None  # the resident single-env stepper (mailbox_open)

    # ---- helpers -------------------------------------------------------------------------------
    def _settle(self):
        """With a resident stepper: wait until the records in HBM hold its last acknowledged step (it writes them through behind the
        acknowledgement; a microsecond at most).  Every path that reads or changes the records passes here first."""
        if self._mb is not None:
            rc = self._lib.snac_mailbox_settle(self._mb)
            if rc:
                _lib.check(rc)

    def _stream(self):
        if self._mb is not None:
            self._settle()
            self._mb_dirty = True                                    # whoever asks for the stream launches: the resident stepper reloads the records
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the handle without building a Stream object (2 us)
        if raw is not None:
            return C.c_void_p(raw(self.device.index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _is_mapped(self, t):
        """A page-locked host tensor: mapped into the GPU's address space, kernels read / write it over the bus (inputs and
        outputs of a host-side caller without a copy command)."""
        if t.device.type != "cpu":
            return False
        if self._mapped.get(id(t)) is t:
            return True
        if not t.is_pinned():
            return False
        if len(self._mapped) >= 32:
            self._mapped.clear()
        self._mapped[id(t)] = t
        return True

    def _i8(self, x, shape, what):
        if x is None:
            return None
        if torch.is_tensor(x) and x.dtype == torch.int8 and x.is_contiguous() and tuple(x.shape) == tuple(shape) \
                and (x.device == self.device or self._is_mapped(x)):
            return x                                            # the per-tick case: nothing to convert
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x), device=self.device)
        if x.device != self.device:
            x = x.to(self.device)
        if tuple(x.shape) != tuple(shape):
            raise ValueError("%s must have shape %s, got %s" % (what, tuple(shape), tuple(x.shape)))
        return x.to(torch.int8).contiguous()

    def _new_obs(self, *lead):
        return torch.empty(tuple(lead) + (self.num_envs, self.obs_dim), dtype=self.obs_dtype, device=self.device)

    def new_host_obs(self):
        """A page-locked host tensor [N, obs_dim] that step_scalar() / reset_scalar() accept as `out`: page-locked memory is
        mapped into the GPU's address space, so the kernel stores its rows there itself and the host only waits for the stream
        (sync()) before reading -- no copy command.  For single-env callers (one launch + one wait per step: 15.6 instead of
        22.1 us, tools/facade_time.py); a large batch keeps its observations on the device."""
        return torch.empty((self.num_envs, self.obs_dim), dtype=self.obs_dtype, pin_memory=True)

    def sync(self):
        """Wait for everything enqueued on this env's current stream (what makes a new_host_obs() row readable)."""
        _lib.check(self._lib.snac_stream_sync(self._stream()))

    def _out_row(self, out):
        if out is None:
            return self._new_obs()
        if out is self._host_ok:                                     # checked before: the per-step call of a single-env class
            return out
        self._buf(out, (self.num_envs, self.obs_dim), self.obs_dtype, "out")
        if out.device.type == "cpu":                                 # new_host_obs(): written by the kernel over the bus
            self._host_ok = out
        return out

    # ---- API -----------------------------------------------------------------------------------
    def reset(self, mask=None, plan_idx=None, want_obs=True, check=True):
        """Reset all envs (or those with mask != 0).  plan_idx: per-env plan row; default counter RNG (dynamic)
        or the single static plan.  Returns the observation of every env [N, obs_dim] (want_obs=False: None).  check=False skips
        the range test of plan_idx -- a device-to-host round trip -- for indices the caller built itself."""
        N = self.num_envs
        m = p = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device)
            if tuple(m.shape) != (N,):
                raise ValueError("mask must have shape (N,)")
            m = (m != 0).to(torch.uint8).contiguous()
        if plan_idx is not None:
            p = torch.as_tensor(plan_idx, device=self.device)
            if tuple(p.shape) != (N,):
                raise ValueError("plan_idx must have shape (N,)")
            if check and (int(p.min()) < 0 or int(p.max()) >= self.num_plans):
                raise ValueError("plan_idx out of range")
            p = p.to(torch.int16).contiguous()
        obs = self._new_obs() if want_obs else None
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_reset(C.byref(self._desc), C.byref(self._state), _ptr(m), _ptr(p), _ptr(obs),
                                            self._stream()))
        self._was_reset = True
        return obs

    def step(self, actions=None, step_size=None, auto_reset=False, want_obs=True, out=None):
        """One vector step.  actions int[N] (None: counter RNG), step_size int[N] in {1,2,3} (None: counter RNG).
        out: optional preallocated (obs [N, obs_dim] obs_dtype, reward [N] float32, done [N] uint8) reused every tick -- a
        training loop that steps small batches is bound by host time, and three allocations are a third of it.  A call that
        passes the SAME tensor objects as the call before (out, actions, step_size -- the per-tick loop of a trainer) skips the
        argument checks: the validated pointers are kept and only compared with the tensors' current data_ptr().
        Inputs and outputs may also be page-locked HOST tensors (torch.empty(..., pin_memory=True); new_host_obs()): the kernel
        reads / writes them over the bus and the caller only waits (sync()) -- what VectorizedEnvWrapper does.
        Returns (obs [N, obs_dim], reward float32 [N], done bool [N])."""
        f = self._fast
        if f is not None and out is not None and f[0] is out and f[1] is actions and f[2] is step_size and f[3] == want_obs \
                and _current_device() == self._dev_index:
            o, r, d = out
            # the same objects may have been changed in place (resize_, set_, as_strided_, .data = ...): the pointers AND the
            # shapes, strides and dtypes the kernel was validated for must still hold, or the full checks run again
            if (o.data_ptr() if want_obs else 0) == f[4] and r.data_ptr() == f[5] and d.data_ptr() == f[6] \
                    and (actions is None or actions.data_ptr() == f[7]) and (step_size is None or step_size.data_ptr() == f[8]) \
                    and _meta(o if want_obs else None, r, d, actions, step_size) == f[10]:
                if self._mb is not None:
                    self._settle()
                    self._mb_dirty = True
                rc = self._snac_step(self._desc_ref, self._state_ref, self.t & 0xFFFFFFFF, f[7], f[8], 1 if auto_reset else 0,
                                     f[4] or None, f[5], f[6], _raw_stream(self._dev_index))
                if rc:
                    _lib.check(rc)
                self.t += 1
                return f[9]
        if not self._was_reset:
            raise _lib.SnacError("step() before reset()")
        N = self.num_envs
        a = self._i8(actions, (N,), "actions")
        k = self._i8(step_size, (N,), "step_size")
        if out is not None:
            obs, reward, done = out
            obs = self._buf(obs, (N, self.obs_dim), self.obs_dtype, "out[0]") if want_obs else None
            reward, done = self._buf(reward, (N,), torch.float32, "out[1]"), self._buf(done, (N,), torch.uint8, "out[2]")
        else:
            obs = self._new_obs() if want_obs else None
            reward = torch.empty((N,), dtype=torch.float32, device=self.device)
            done = torch.empty((N,), dtype=torch.uint8, device=self.device)
        args = (C.byref(self._desc), C.byref(self._state), self.t & 0xFFFFFFFF, _ptr(a), _ptr(k), int(bool(auto_reset)),
                _ptr(obs), _ptr(reward), _ptr(done))
        if torch.cuda.current_device() == self.device.index:
            _lib.check(self._lib.snac_step(*args, self._stream()))
        else:
            with torch.cuda.device(self.device):
                _lib.check(self._lib.snac_step(*args, self._stream()))
        self.t += 1
        ret = (obs, reward, done.view(torch.bool))
        # remember a call whose tensors were all taken as they are (nothing converted or copied): the next call with the same
        # objects goes straight to the launch
        if out is not None and a is actions and k is step_size and _raw_stream is not None:
            self._fast = (out, actions, step_size, bool(want_obs), obs.data_ptr() if obs is not None else 0, reward.data_ptr(), done.data_ptr(),
                          a.data_ptr() if a is not None else None, k.data_ptr() if k is not None else None, ret,
                          _meta(obs, reward, done, actions, step_size))
        else:
            self._fast = None
        return ret

    def step_scalar(self, action, step_size, auto_reset=False, out=None):
        """step() with ONE action and ONE step size for every env, passed by value (snac_step_scalar): no host-to-device copy
        precedes the launch.  With obs_tail "record" the row also carries reward, done and the header, so a single-env caller
        reads everything back with one copy.  out: preallocated obs [N, obs_dim] on the device, or a new_host_obs() row.  Returns obs."""
        if not self._was_reset:
            raise _lib.SnacError("step() before reset()")
        obs = self._out_row(out)
        args = (C.byref(self._desc), C.byref(self._state), self.t & 0xFFFFFFFF, int(action), int(step_size), int(bool(auto_reset)),
                _ptr(obs), None, None)
        if torch.cuda.current_device() == self.device.index:
            _lib.check(self._lib.snac_step_scalar(*args, self._stream()))
        else:
            with torch.cuda.device(self.device):
                _lib.check(self._lib.snac_step_scalar(*args, self._stream()))
        self.t += 1
        return obs

    def step_scalar_wait(self, action, step_size, out):
        """step_scalar(action, step_size, out=<a new_host_obs() row>) followed by sync(), as ONE call: what a single-env class does per
        step.  A call with the same row object as the call before goes straight to the two library calls (no argument checks, no
        descriptor / pointer objects built: 3 of the 18 us of such a step).  Returns `out`, readable."""
        f = self._ssf
        if f is not None and f[0] is out and out.data_ptr() == f[2] and _current_device() == self._dev_index:
            st = _raw_stream(self._dev_index)
            if self._mb is not None:
                self._settle()
                self._mb_dirty = True
            rc = self._snac_step_scalar(self._desc_ref, self._state_ref, self.t & 0xFFFFFFFF, action, step_size, 0, f[1], None, None, st)
            if rc:
                _lib.check(rc)
            self.t += 1
            rc = self._snac_sync(st)
            if rc:
                _lib.check(rc)
            return out
        obs = self.step_scalar(action, step_size, out=out)
        self.sync()
        if out is self._host_ok and _raw_stream is not None and isinstance(action, int) and isinstance(step_size, int):
            self._ssf = (out, C.c_void_p(out.data_ptr()), out.data_ptr())
        return obs

    def reset_scalar(self, plan_idx, out=None):
        """reset() of every env onto plan row `plan_idx`, passed by value (snac_reset_scalar).  Returns obs [N, obs_dim]."""
        obs = self._out_row(out)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_reset_scalar(C.byref(self._desc), C.byref(self._state), int(plan_idx), _ptr(obs), self._stream()))
        self._was_reset = True
        return obs

    def _buf(self, t, shape, dtype, what):
        if tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous() or not (t.device == self.device or self._is_mapped(t)):
            raise ValueError("%s must be a contiguous %s tensor of shape %s on %s (or in page-locked host memory)"
                             % (what, dtype, tuple(shape), self.device))
        return t

    def rollout(self, T, actions=None, step_size=None, obs="all", out=None, want_reward=True, want_done=True,
                reward_out=None, done_out=None, record=None, ring=None):
        """T vector steps with auto-reset in ONE launch (the loop of multiprocess.py:82-84).
        actions / step_size: int[T, N] or None (counter RNG).  obs: "all" -> [T, N, D], "last" -> [N, D], None;
        "tiled" -> [ceil(N / 64), T, 64, D], every observation in tile-major order (row (t, env) at [env // 64, t, env % 64];
        out[g] is the contiguous, reference-shaped [T, 64, D] trajectory of env group g):
        each tile of 64 envs streams through its own region -- the faster layout for trajectories that stay on the GPU
        (untile() gives the [T, N, D] view of it as a copy).  ring=(ring_ticks, first_tick) with obs="tiled": `out` is a tile-major
        RING [ceil(N / 64), ring_ticks, 64, D] and this call writes its steps first_tick .. first_tick + T - 1 (snac_rollout_tiled).
        out / reward_out / done_out: optional preallocated outputs (done_out uint8).  record: optional dict of
        preallocated [T, N] tensors {"actions": int8, "step_size": int8, "plan_idx": int16, "first": uint8} that receive
        the action taken, the step size used, the plan row in effect and the first-step-of-episode flag of every env-step.
        Returns (obs, reward [T, N] float32, done [T, N] bool).

        Memory: without `out=`, an observation tensor of 1 GiB or more comes from snac_amd.trajmem.cached_empty -- blocks of the HIP
        virtual-memory API OUTSIDE torch's caching allocator (invisible to torch.cuda.memory_*, not released by empty_cache()): the
        first request of a size builds a block (0.1-5 s; while it measures, its pool may hold up to SNAC_TRAJ_POOL_CAP_BYTES, default
        64 GiB and never more than half of what is free, beyond the block), later ones recycle it; up to SNAC_TRAJ_CACHE_BYTES
        (40 GiB per device) of free blocks are retained until trajmem.cache_trim() -- which this class calls by itself when torch
        runs out of memory while allocating an output.  SNAC_TRAJ_CACHE=0 switches the cache off (torch.empty: one physical run,
        5.7 instead of 7.0 TB/s for the headline's rows).  A recycled block is ordered behind its previous user's work on the stream
        it was ALLOCATED on; a tensor used on another stream must be waited for before it is dropped (torch's record_stream rule)."""
        if not self._was_reset:
            raise _lib.SnacError("rollout() before reset()")
        N, T = self.num_envs, int(T)
        a = self._i8(actions, (T, N), "actions")
        k = self._i8(step_size, (T, N), "step_size")
        mode = {"all": _lib.OBS_ALL, "last": _lib.OBS_LAST, "tiled": _lib.OBS_TILED, None: _lib.OBS_NONE}[obs]
        o = None
        if mode != _lib.OBS_NONE:
            if ring is not None and (mode != _lib.OBS_TILED or out is None):
                raise ValueError("ring=(ring_ticks, first_tick) goes with obs='tiled' and a preallocated `out`")
            shape = {_lib.OBS_ALL: (T, N, self.obs_dim), _lib.OBS_LAST: (N, self.obs_dim),
                     _lib.OBS_TILED: ((N + 63) // 64, T if ring is None else int(ring[0]), 64, self.obs_dim)}[mode]
            if out is not None:
                if tuple(out.shape) != shape or out.dtype != self.obs_dtype or out.device != self.device or not out.is_contiguous():
                    raise ValueError("out must be a contiguous %s tensor of shape %s on %s" % (self.obs_dtype, shape, self.device))
                o = out
            else:
                o = self._traj_out(shape)
        if reward_out is not None:
            reward = self._buf(reward_out, (T, N), torch.float32, "reward_out")
        else:
            reward = self._empty((T, N), torch.float32) if want_reward else None
        if done_out is not None:
            done = self._buf(done_out, (T, N), torch.uint8, "done_out")
        else:
            done = self._empty((T, N), torch.uint8) if want_done else None
        rec = None
        if record is not None:
            kinds = {"actions": torch.int8, "step_size": torch.int8, "plan_idx": torch.int16, "first": torch.uint8}
            ptrs = {}
            for name, dt in kinds.items():
                t = record.get(name)
                ptrs[name] = None if t is None else self._buf(t, (T, N), dt, "record[%r]" % name).data_ptr()
            rec = _lib.RolloutRecord(ptrs["actions"], ptrs["step_size"], ptrs["plan_idx"], ptrs["first"])
        with torch.cuda.device(self.device):
            if ring is not None:
                _lib.check(self._lib.snac_rollout_tiled(C.byref(self._desc), C.byref(self._state), T, self.t & 0xFFFFFFFF, _ptr(a),
                                                        _ptr(k), int(ring[0]), int(ring[1]), _ptr(o), _ptr(reward), _ptr(done),
                                                        C.byref(rec) if rec is not None else None, self._stream()))
            else:
                _lib.check(self._lib.snac_rollout_rec(C.byref(self._desc), C.byref(self._state), T, self.t & 0xFFFFFFFF, _ptr(a),
                                                      _ptr(k), mode, _ptr(o), _ptr(reward), _ptr(done),
                                                      C.byref(rec) if rec is not None else None, self._stream()))
        self.t += T
        return o, reward, (done.view(torch.bool) if done is not None else None)

    # outputs of a GiB and more that rollout() allocates itself come from the cache of measured trajectory blocks (snac_amd/trajmem.py
    # cached_empty: chunks of two slices of the physical address space taking turns -- 7.0 instead of 5.7-6.0 TB/s for the rows of a
    # headline pass; built on first use (0.1-5 s), recycled afterwards); anything smaller, or a box where that fails: torch.empty
    TRAJ_MIN_BYTES = 1 << 30

    def _traj_out(self, shape):
        numel = 1
        for d in shape:
            numel *= int(d)
        if numel * (8 if self.obs_dtype == torch.float64 else 4) >= self.TRAJ_MIN_BYTES and not getattr(self, "_traj_failed", False):
            try:
                from . import trajmem

                if trajmem._cache_limit() > 0:
                    return trajmem.cached_empty(shape, self.obs_dtype, self.device, pool_cap=trajmem.default_pool_cap())
            except (RuntimeError, OSError) as e:                    # SnacError is a RuntimeError: no such block on this box / out of ranges
                import warnings

                self._traj_failed = True
                warnings.warn("rollout(): no trajectory block (%s); its outputs come from torch.empty from now on" % (e,))
        return self._empty(shape, self.obs_dtype)

    def _empty(self, shape, dtype):
        """torch.empty on this batch's device; when torch is out of memory the free list of measured trajectory blocks (memory torch's
        allocator cannot see or reclaim) is given back and the allocation tried once more."""
        try:
            return torch.empty(shape, dtype=dtype, device=self.device)
        except torch.cuda.OutOfMemoryError:
            from . import trajmem

            if trajmem.cache_trim(self.device.index) == 0:
                raise
            torch.cuda.empty_cache()
            return torch.empty(shape, dtype=dtype, device=self.device)

    def alloc_trajectory(self, T, candidates=2, reps=3, layout="ticks", memory="vmm"):
        """The [T, N, obs_dim] output tensor of rollout(T, out=...), allocated where this batch's rollout writes fastest.  On
        MI355X write streams confined to one 32 GiB slice of the physical address space reach ~5.7 TB/s, spread over several
        ~7.1: memory="vmm" (default) takes the tensor from snac_traj_alloc (snac_amd/trajmem.py: one virtual
        range over chunks from two slices taking turns, found by measurement; 1.5-2.5 s per block), memory="malloc" from
        torch.empty (hipMalloc: one run).  `candidates` tensors are allocated, a copy of this batch rolls out into each and the
        fastest is kept (snac_amd/placement.py; blocks differ by 1-3 % now, hipMalloc tensors by 5-9 %).  The batch itself is not stepped.  layout "tiled": the tensor of
        rollout(obs="tiled").  Returns (tensor, report)."""
        from . import placement

        if not self._was_reset:
            raise _lib.SnacError("alloc_trajectory() before reset()")
        if memory not in ("vmm", "malloc"):
            raise ValueError("memory must be 'vmm' or 'malloc'")
        alloc = None
        if memory == "vmm":
            from . import trajmem

            alloc = trajmem.traj_empty
        scratch = self.fork(torch.arange(self.num_envs, device=self.device))
        T = int(T)
        if layout == "tiled":                                        # [ceil(N / 64), T, 64, D] for rollout(obs="tiled")
            t, rep = placement.fastest_tensor(((self.num_envs + 63) // 64, T, 64, self.obs_dim), self.obs_dtype, self.device,
                                              lambda t: scratch.rollout(T, obs="tiled", out=t, want_reward=False, want_done=False),
                                              candidates=candidates, reps=reps, alloc=alloc)
        else:
            t, rep = placement.fastest_tensor((T, self.num_envs, self.obs_dim), self.obs_dtype, self.device,
                                              lambda t: scratch.rollout(T, obs="all", out=t, want_reward=False, want_done=False),
                                              candidates=candidates, reps=reps, alloc=alloc)
        rep["memory"] = memory
        return t, rep

    def untile(self, tiled):
        """[ceil(N / 64), T, 64, D] (rollout(obs="tiled")) -> a [T, N, D] copy in the reference order."""
        G, T, E, D = tiled.shape
        return tiled.permute(1, 0, 2, 3).reshape(T, G * E, D)[:, :self.num_envs]

    def set_plan_row(self, index, full_plan, update_tb=False):
        """Replace row `index` of the device plan table by `full_plan` ([30] / [26, 26] as the reference stores it).
        update_tb=False keeps the row's total_brick: the hindsight scripts overwrite env.plan AFTER reset() has
        computed total_brick from the original plan (script/DRQN_hindsight/2d/DRQN_hindsight_2D_static.py:245-250)."""
        index = int(index)
        if not 0 <= index < self.num_plans:
            raise ValueError("plan index out of range")
        self._sync_plans_full()
        full = np.asarray(full_plan, np.float64)
        packed, tb = _plans.pack_plans(self.kind, full[None])
        row = torch.from_numpy(packed.view(np.int32) if self.kind == 2 else packed)[0]
        self._settle()
        self._plans[index].copy_(row.to(self.device))
        self._mb_dirty = True
        self._table_version += 1
        self.plans_full[index] = full
        if update_tb:
            self._plan_tb[index] = int(tb[0])


    def plans_from_grids(self, src=None, rows=None, environment_memory=None, total_brick=None, first=0):
        """Plan rows [first, first + m) of THIS batch's device table <- the grids of finished episodes, on the device
        (snac_plans_from_grids): what the DRQN_hindsight scripts do to the hindsight env's plan before they replay an episode
        (script/DRQN_hindsight/2d/DRQN_hindsight_2D_dynamic.py:270-282).  Either src (a BatchedDMPEnv of the same kind: its current
        grids; rows int[m] picks envs, None: all of them in order) or environment_memory (float64 [m, ...] in the reference's format,
        a device tensor).  total_brick int[m]: the rows' total_brick (None with src: each source env's own, from its header).
        Envs keep stepping on their current row: reset them to pick the new plans up."""
        tb = None
        if total_brick is not None:
            tb = torch.as_tensor(total_brick, device=self.device).to(torch.int32).contiguous()
        ri = None
        if src is not None:
            if src.kind != self.kind or src.device != self.device:
                raise ValueError("src must be a batch of the same kind on the same device")
            if rows is not None:
                ri = torch.as_tensor(rows, device=self.device).to(torch.int32).contiguous()
            m = int(ri.numel()) if ri is not None else src.num_envs
            mem, sstate, sn = None, C.byref(src._state), src.num_envs
            src._settle()                                            # src's resident stepper writes its records through BEHIND the acknowledgement
            # (src's launches must be on the current stream of this device, or waited for: the read below is ordered on that stream only)
        else:
            sz = self.sizes
            mem = torch.as_tensor(environment_memory, device=self.device).to(torch.float64).reshape(-1, sz.env_height, sz.env_width).contiguous()
            m, sstate, sn = int(mem.shape[0]), None, 0
            if tb is None:
                raise ValueError("environment_memory needs total_brick")
        if tb is not None and tuple(tb.shape) != (m,):
            raise ValueError("total_brick must have shape (%d,)" % m)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_plans_from_grids(C.byref(self._desc), C.byref(self._state), m, int(first), sstate, sn, _ptr(ri),
                                                       _ptr(mem), _ptr(tb), self._stream()))
        self._plans_stale = True
        self._table_version += 1
        return m


    # ---- the resident single-env stepper (snac_mailbox_*: what the drop-in classes step through) ------------------------------
    def mailbox_open(self, idle_us=0):
        """N <= 256 (up to four wavefronts, an env per lane; the drop-in classes: N = 1).  Creates the mailbox of this batch (coherent page-locked
        host memory + a stream of its own) and returns its rows: a host tensor [N, obs_dim] of obs_dtype that mailbox_step() /
        mailbox_step_n() fill -- and that reset_scalar() / step_scalar() accept as `out` (it is page-locked: the launch path writes it
        over the bus too).  A wavefront becomes resident at the first mailbox_step()
        and leaves by itself after idle_us (0: 1000) microseconds without a step, at mailbox_close() and at interpreter exit."""
        import weakref

        if self._mb is not None:
            return self._mb_row
        mb = C.c_void_p()
        with torch.cuda.device(self.device):                         # the mailbox records the CURRENT device: its waves and streams live there,
            _lib.check(self._lib.snac_mailbox_create(C.byref(self._desc), int(idle_us), C.byref(mb)))   # whatever is current when a step arms one
        ptr = self._lib.snac_mailbox_row(mb)
        n = self.obs_dim * self.num_envs
        if self.obs_dtype == torch.float64:
            arr = np.ctypeslib.as_array((C.c_double * n).from_address(ptr))
        else:
            arr = np.ctypeslib.as_array((C.c_float * n).from_address(ptr))
        row = torch.from_numpy(arr.reshape(self.num_envs, self.obs_dim))
        N = self.num_envs
        self._mb_reward = np.ctypeslib.as_array((C.c_float * N).from_address(self._lib.snac_mailbox_reward(mb)))
        self._mb_done = np.ctypeslib.as_array((C.c_uint8 * N).from_address(self._lib.snac_mailbox_done(mb)))
        self._mb_step_n = self._lib.snac_mailbox_step_n
        self._mapped[id(row)] = row                                  # page-locked and mapped: a valid `out` of the launch path
        self._mb, self._mb_row, self._mb_dirty = mb, row, True
        self._mb_step = self._lib.snac_mailbox_step
        self._mb_final = weakref.finalize(self, self._lib.snac_mailbox_destroy, mb)    # gc and interpreter exit: the wave is told to leave
        return row

    def mailbox_step(self, action, step_size):
        """One step of the env through its resident wave (semantics of step_scalar(action, step_size) + sync(), no auto-reset);
        the row of mailbox_open() holds the result when this returns."""
        if self._mb_dirty:                                           # another entry point has touched the records: wait for it, then say so
            _lib.check(self._lib.snac_stream_sync(C.c_void_p(_raw_stream(self._dev_index) if _raw_stream is not None
                                                              else torch.cuda.current_stream(self.device).cuda_stream)))
            _lib.check(self._lib.snac_mailbox_touch(self._mb))
            self._mb_dirty = False
        rc = self._mb_step(self._mb, self._desc_ref, self._state_ref, action, step_size)
        if rc:
            _lib.check(rc)
        self.t += 1

    def mailbox_step_n(self, actions, step_size):
        """One vector step of the batch (N <= 256) through its resident waves: actions / step_size contiguous int8 numpy arrays [N]
        (host memory).  When this returns the rows of mailbox_open() hold the observations and mailbox_outputs() the rewards and
        done flags; no auto-reset (the reference's wrapper has none)."""
        if self._mb_dirty:
            _lib.check(self._lib.snac_stream_sync(C.c_void_p(_raw_stream(self._dev_index) if _raw_stream is not None
                                                              else torch.cuda.current_stream(self.device).cuda_stream)))
            _lib.check(self._lib.snac_mailbox_touch(self._mb))
            self._mb_dirty = False
        rc = self._mb_step_n(self._mb, self._desc_ref, self._state_ref, actions.ctypes.data, step_size.ctypes.data)
        if rc:
            _lib.check(rc)
        self.t += 1

    def mailbox_outputs(self):
        """(reward float32 [N], done uint8 [N]) numpy views of the mailbox, rewritten by every mailbox step."""
        return self._mb_reward, self._mb_done

    def mailbox_stats(self):
        """dict(launches, steps_served, alive, idle_us, last_step_us: the wave's own timing of its last step) of this env's mailbox (None without one)."""
        if self._mb is None:
            return None
        out = (C.c_uint32 * 8)()
        _lib.check(self._lib.snac_mailbox_stats(self._mb, C.byref(out)))
        return dict(launches=int(out[0]), steps_served=int(out[1]), alive=bool(out[2]), idle_us=int(out[3]),
                    last_step_us=dict(transition=out[4] / 100.0, row_stores=out[5] / 100.0, fence=out[6] / 100.0, write_through=out[7] / 100.0))

    def mailbox_close(self):
        """Ends the resident wave and frees the mailbox (its row tensor must not be used afterwards)."""
        if self._mb is not None:
            self._mb_final()                                         # snac_mailbox_destroy, once
            self._mb, self._mb_row = None, None

    # ---- plan generators on the device ---------------------------------------------------------------
    def generate_plans(self, first=0, count=None, sparse=False, seed=None, id_base=0, vertices=None):
        """Rewrite rows [first, first + count) of the device plan table with freshly generated plans (snac_make_plans): the
        random triangles of the reference's 2D / 3D create_plan() (Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py:37-59)
        or, 1D, its random sine curves (Env/1D/DMP_Env_1D_dynamic_hindsight_replay.py:29-42) -- one wavefront per plan, so a
        65 536-env batch can draw from up to 32 767 plans (the table's limit: plan rows are int16) instead of the 400 stored ones.  Plan row r is keyed by
        (seed, id_base + r) on counter-RNG stream 2; vertices: optional int8 [count, 6] = x0 y0 x1 y1 x2 y2 rasterised as
        given (no redraw).  Envs keep stepping on their current row: reset them to pick the new plans up.
        Returns the number of cells set per plan (int32 [count]; 1D: total_brick)."""
        count = self.num_plans - first if count is None else int(count)
        v = None
        if vertices is not None:
            v = torch.as_tensor(np.asarray(vertices), device=self.device).to(torch.int8).contiguous()
            if tuple(v.shape) != (count, 6):
                raise ValueError("vertices must have shape (count, 6)")
        area = torch.empty((count,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_make_plans(C.byref(self._desc), C.byref(self._state), int(first), count, int(bool(sparse)),
                                                 int(self.seed if seed is None else seed) & 0xFFFFFFFFFFFFFFFF, int(id_base),
                                                 _ptr(v), _ptr(area), self._stream()))
        self._plans_stale = True                               # the host copy (plans_full) no longer mirrors the device table
        self._table_version += 1
        return area

    def generate_plans_numpy(self, first=0, count=None, sparse=False):
        """Rows [first, first + count) of the device plan table drawn THE WAY THE REFERENCE DRAWS THEM, from numpy's global stream, one
        row after the other -- so that a script that calls np.random.seed(s) gets the plans the reference's create_plan() would give it:
          2D / 3D  Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py:37-59: per attempt x = np.random.randint(0, 20, size=3), then
                   y likewise; the triangle is rasterised on the device with explicit vertices (snac_make_plans: cv2's rules restated,
                   the 2000 dataset plans reproduced bit for bit) and redrawn while the area is <= 50 (dense) / 20 (sparse) -- 3D also
                   while it is >= 110 (script/HumanPlayerGUI/env/Env3D.py:360-364, how the 3D datasets were drawn);
          1D       Env/1D/DMP_Env_1D_dynamic_hindsight_replay.py:29-42 through plans.random_sin_plan() (numpy's own sine, on the host).
        generate_plans() is the batched form (counter RNG, one launch for all rows); this one is for seed-level parity and costs a
        launch and a wait per attempt.  Returns the areas (1D: total_brick) as a list of ints."""
        count = self.num_plans - first if count is None else int(count)
        if first < 0 or count < 0 or first + count > self.num_plans:
            raise ValueError("plan rows out of range")
        areas = []
        if self.kind == 1:
            for r in range(first, first + count):
                y, area, _ = _plans.random_sin_plan()
                self.set_plan_row(r, y, update_tb=True)
                areas.append(int(area))
            return areas
        lo, hi = (20 if sparse else 50), (110 if self.kind == 3 else 401)
        for r in range(first, first + count):
            while True:
                x = np.random.randint(0, 20, size=3)
                y = np.random.randint(0, 20, size=3)
                v = np.array([[x[0], y[0], x[1], y[1], x[2], y[2]]], np.int8)
                area = int(self.generate_plans(r, 1, sparse=sparse, vertices=v).item())
                if lo < area < hi:
                    break
            areas.append(area)
        return areas

    def _sync_plans_full(self):
        """Decode the device plan table back into the reference's host format after generate_plans()."""
        if not getattr(self, "_plans_stale", False):
            return
        t = self._plans.cpu().numpy()
        P = self.num_plans
        if self.kind == 1:
            self.plans_full = t[:, :30].astype(np.float64)
        else:
            full = np.zeros((P, 26, 26), np.float64)
            if self.kind == 2:
                bits = (t.view(np.uint32)[:, :, None] >> np.arange(20, dtype=np.uint32)[None, None, :]) & 1
                full[:, 3:23, 3:23] = bits
            else:
                full[:, 3:23, 3:23] = t.reshape(P, 20, 20)
            self.plans_full = full
        self._plans_stale = False

    # ---- snapshots (MCTS-style branching, checkpoints) ---------------------------------------------
    def state_dict(self):
        """Everything that defines the envs' future (tensors are cloned): the MCTS variants of the reference snapshot
        (position, grid, count_brick, count_step) per env (Env/2D/DMP_ENV_2D_dynamic_MCTS.py:117-175).  The device plan table and
        its total_brick column travel with the snapshot: generate_plans() / set_plan_row() change them, and a header's plan row
        means nothing without the table it indexes."""
        # the table is cloned once per VERSION of it, not once per snapshot (tree search snapshots per node; a generated 3D table is
        # 26 MB): every snapshot of an unchanged table shares one read-only clone
        self._settle()
        snap = self._table_snapshot
        if snap is None or snap[0] != self._table_version:
            snap = self._table_snapshot = (self._table_version, self._plans.clone(), self._plan_tb.clone())
        return dict(hdr=self._hdr.clone(), episode=self._episode.clone(), grid=self._grid.clone(), stats=self._stats.clone(),
                    plans=snap[1], plan_tb=snap[2],
                    t=self.t, kind=self.kind, dynamic=self.dynamic, num_envs=self.num_envs)

    def load_state_dict(self, sd):
        if (sd["kind"], sd["dynamic"], sd["num_envs"]) != (self.kind, self.dynamic, self.num_envs):
            raise ValueError("snapshot belongs to a different env batch")
        if "plans" in sd:
            if tuple(sd["plans"].shape) != tuple(self._plans.shape):
                raise ValueError("snapshot holds a plan table of another size")
            snap = self._table_snapshot
            if not (snap is not None and snap[0] == self._table_version and sd["plans"] is snap[1] and sd["plan_tb"] is snap[2]):
                self._plans.copy_(sd["plans"]); self._plan_tb.copy_(sd["plan_tb"])   # (a snapshot of the current table: nothing to copy)
                self._table_version += 1
                self._plans_stale = True                           # plans_full is re-decoded from the device table when it is needed
        self._settle()
        self._hdr.copy_(sd["hdr"]); self._episode.copy_(sd["episode"]); self._grid.copy_(sd["grid"]); self._stats.copy_(sd["stats"])
        self._mb_dirty = True
        self.t = int(sd["t"])
        self._was_reset = True

    def fork(self, index):
        """New batch whose env j is a copy of this batch's env index[j] (same plans, seed, tick): expand tree leaves into
        children and step each child with its own action -- the batched form of the MCTS variants' functional
        transition(state, action).  Episodic sums start at zero; counter-RNG streams are keyed by the NEW local index."""
        index = torch.as_tensor(index, device=self.device, dtype=torch.long)
        self._settle()
        self._sync_plans_full()
        child = BatchedDMPEnv(self.kind, self.dynamic, int(index.numel()), plans=self.plans_full, device=self.device, seed=self.seed,
                              obs_dtype=self.obs_dtype, env_id_base=self.env_id_base, total_step=self.total_step,
                              brick_gt=self.brick_gt, time_gt=self.time_gt, frame_value=self.frame_value,
                              obs_scalars=self.obs_scalars, obs_tail=self.obs_tail, static_plan=self.static_plan)
        child._plan_tb.copy_(self._plan_tb)                          # caller-supplied total_brick rows travel with the fork
        child._table_version += 1
        child._hdr.copy_(self._hdr[index]); child._episode.copy_(self._episode[index]); child._grid.copy_(self._grid[index])
        child.t = self.t
        child._was_reset = self._was_reset
        return child

    # ---- tree search: the batch as a node pool (MCTS variants of the reference) ----------------------
    def _index(self, x, m, what):
        if x is None:
            return None
        x = torch.as_tensor(x, device=self.device)
        if tuple(x.shape) != (m,):
            raise ValueError("%s must have shape (%d,)" % (what, m))
        if m and (int(x.min()) < 0 or int(x.max()) >= self.num_envs):
            raise ValueError("%s out of range" % what)
        return x.to(torch.int32).contiguous()

    def transition(self, actions, step_size=None, src=None, dst=None, t=0, want_obs=True):
        """m functional transitions (Env/2D/DMP_ENV_2D_dynamic_MCTS.py:117-175 `transition(state, action)`), the batch used
        as a pool of tree nodes: row dst[i] <- step(row src[i], actions[i], step_size[i]); src / dst None = row i.  No
        auto-reset, the episodic sums are untouched.  step_size None: counter RNG keyed by (env_id_base + i, t).  A dst row
        must not be the src row of another edge of the same call.  Returns (obs [m, D], reward [m], done [m])."""
        if not self._was_reset:
            raise _lib.SnacError("transition() before reset() / import_states()")
        if actions is None:
            raise ValueError("actions are required")
        a = torch.as_tensor(actions, device=self.device) if not torch.is_tensor(actions) else actions.to(self.device)
        m = int(a.numel())
        a = self._i8(a.reshape(-1), (m,), "actions")
        k = self._i8(step_size, (m,), "step_size")
        si, di = self._index(src, m, "src"), self._index(dst, m, "dst")
        if (si is None or di is None) and m > self.num_envs:
            raise ValueError("more transitions than pool rows")
        if si is not None or di is not None:
            s_ = si if si is not None else torch.arange(m, device=self.device, dtype=torch.int32)
            d_ = di if di is not None else torch.arange(m, device=self.device, dtype=torch.int32)
            clash = torch.isin(d_, s_[s_ != d_]) if m else torch.zeros(0, dtype=torch.bool)
            if m and (bool(clash.any()) or int(torch.unique(d_).numel()) != m):
                raise ValueError("dst rows must be distinct and must not be the src row of another edge")
        obs = torch.empty((m, self.obs_dim), dtype=self.obs_dtype, device=self.device) if want_obs else None
        reward = torch.empty((m,), dtype=torch.float32, device=self.device)
        done = torch.empty((m,), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_transition(C.byref(self._desc), C.byref(self._state), m, _ptr(si), _ptr(di), int(t) & 0xFFFFFFFF,
                                                 _ptr(a), _ptr(k), _ptr(obs), _ptr(reward), _ptr(done), self._stream()))
        return obs, reward, done.view(torch.bool)

    def evaluate(self, rows, horizon, gamma, first_reward=None):
        """Default-policy evaluation of tree leaves, the "Evaluation" block of the vanilla MCTS procedure
        (script/MCTS/utils/mcts.py:100-110): from each pool row in `rows`, up to `horizon` uniformly random steps that stop
        at the first `done`, and   estimate = first_reward + sum_t reward_t * gamma**t   accumulated in that order in
        float64 (gamma**t as python computes it), so that it equals the reference loop bit for bit given the same
        actions.  Actions and step sizes come from the counter RNG (tick t of leaf i is keyed by (env_id_base + i, t));
        a leaf whose last step returned done is not rolled out.  The pool rows are left untouched (the leaves are forked).
        Returns (estimate float64 [m], steps int64 [m]: the number of steps actually taken)."""
        rows = torch.as_tensor(rows, device=self.device, dtype=torch.long)
        m, H = int(rows.numel()), int(horizon)
        est = torch.zeros(m, dtype=torch.float64, device=self.device) if first_reward is None else \
            torch.as_tensor(first_reward, device=self.device).to(torch.float64).clone()
        if m == 0 or H <= 0:
            return est, torch.zeros(m, dtype=torch.int64, device=self.device)
        terminal = self.need_reset[rows]
        leaves = self.fork(rows)
        leaves.t = 0
        leaves._hdr.view(torch.int8)[:, 2] &= ~_lib.FLAG_NEED_RESET        # a terminal leaf is masked below, not reset
        _, reward, done = leaves.rollout(H, obs=None)
        # the sums on the device (snac_discounted_return: one leaf per lane, sequential in t -- the reference's summation order --, product and
        # sum each rounded to float64); gamma**t as python computes it, so the powers come from the host.  (Round 6: a python loop of H steps
        # of torch operations before -- 2400 launches for 600 ticks.)
        gpow = torch.tensor([float(gamma) ** t for t in range(H)], dtype=torch.float64).to(self.device)
        steps = torch.empty(m, dtype=torch.int64, device=self.device)
        est = est.contiguous()
        term = terminal.to(torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_discounted_return(H, m, _ptr(reward), _ptr(done.view(torch.uint8)), _ptr(term), _ptr(gpow), _ptr(est), _ptr(steps),
                                                        self._stream()))
        return est, steps

    def import_states(self, position, count_brick, count_step, environment_memory, plan_idx=None, total_brick=None, dst=None):
        """Load states in the reference's own format -- the (position, environment_memory, count_brick, count_step) tuples
        of the MCTS variants -- into pool rows dst (None: rows 0..m-1).  position [m, 2] (1D: [m]), environment_memory
        [m, H, W] float64 with its frame; plan_idx None keeps each row's plan; total_brick None = that of the plan row."""
        sz = self.sizes
        mem = torch.as_tensor(environment_memory, device=self.device).to(torch.float64)
        mem = mem.reshape(-1, sz.env_height, sz.env_width).contiguous()
        m = int(mem.shape[0])

        def i32(x, shape, what):
            if x is None:
                return None
            x = torch.as_tensor(x, device=self.device)
            if tuple(x.shape) != shape:
                raise ValueError("%s must have shape %s" % (what, shape))
            return x.to(torch.int32).contiguous()

        pos = torch.as_tensor(position, device=self.device).to(torch.int32)
        if self.kind == 1:
            pos = torch.stack([pos.reshape(m), torch.zeros(m, dtype=torch.int32, device=self.device)], dim=1)
        pos = i32(pos, (m, 2), "position")
        lo, hi = sz.half_window, sz.half_window + sz.plan_width - 1
        if m and (int(pos[:, 0].min()) < lo or int(pos[:, 0].max()) > hi or
                  (self.kind != 1 and (int(pos[:, 1].min()) < lo or int(pos[:, 1].max()) > hi))):
            raise ValueError("position outside the plan area")
        cb, cs = i32(count_brick, (m,), "count_brick"), i32(count_step, (m,), "count_step")
        p, tb, di = i32(plan_idx, (m,), "plan_idx"), i32(total_brick, (m,), "total_brick"), self._index(dst, m, "dst")
        if p is None and not self._was_reset:
            raise _lib.SnacError("import_states() without plan_idx before reset()")
        if p is not None and m and (int(p.min()) < 0 or int(p.max()) >= self.num_plans):
            raise ValueError("plan_idx out of range")
        if di is None and m > self.num_envs:
            raise ValueError("more states than pool rows")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_import_state(C.byref(self._desc), C.byref(self._state), m, _ptr(di), _ptr(pos), _ptr(cb),
                                                   _ptr(cs), _ptr(p), _ptr(tb), _ptr(mem), self._stream()))
        self._was_reset = True

    def obs_equal(self, obs_a, obs_b, idx_a=None, idx_b=None):
        """equality_operator (np.array_equal of two observations, Env/2D/DMP_ENV_2D_dynamic_MCTS.py:254-258) for m pairs of
        rows: bool [m] = all(obs_a[idx_a[i]] == obs_b[idx_b[i]]); index None = row i."""
        for o in (obs_a, obs_b):
            if o.dim() != 2 or o.shape[1] != self.obs_dim or o.dtype != self.obs_dtype or o.device != self.device or not o.is_contiguous():
                raise ValueError("observations must be contiguous [rows, %d] %s tensors on %s" % (self.obs_dim, self.obs_dtype, self.device))
        ra, rb = int(obs_a.shape[0]), int(obs_b.shape[0])
        m = len(idx_a) if idx_a is not None else (len(idx_b) if idx_b is not None else min(ra, rb))

        def idx(x, rows):
            if x is None:
                if m > rows:
                    raise ValueError("more pairs than rows")
                return None
            x = torch.as_tensor(x, device=self.device)
            if tuple(x.shape) != (m,) or (m and (int(x.min()) < 0 or int(x.max()) >= rows)):
                raise ValueError("bad row index")
            return x.to(torch.int32).contiguous()

        ia, ib = idx(idx_a, ra), idx(idx_b, rb)
        out = torch.empty((m,), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_obs_equal(C.byref(self._desc), _ptr(obs_a), _ptr(ia), ra, _ptr(obs_b), _ptr(ib), rb, m,
                                                _ptr(out), self._stream()))
        return out.view(torch.bool)

    def observe(self):
        if not self._was_reset:
            raise _lib.SnacError("observe() before reset()")
        obs = self._new_obs()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_observe(C.byref(self._desc), C.byref(self._state), _ptr(obs), self._stream()))
        return obs

    def iou(self):
        """float64 [N]: env.iou() (1D, 3D) / the caller-side boolean IoU of the 2D scripts."""
        out = torch.empty((self.num_envs,), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_iou(C.byref(self._desc), C.byref(self._state), _ptr(out), self._stream()))
        return out

    def environment_memory(self):
        """float64 [N, H, W] with the -1 frame, as the reference holds it."""
        sz = self.sizes
        out = torch.empty((self.num_envs, sz.env_height, sz.env_width), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.snac_export_grid(C.byref(self._desc), C.byref(self._state), _ptr(out), self._stream()))
        return out

    # ---- state views (decoded from the packed 16-byte header) -------------------------------------
    def _h8(self):
        self._settle()
        return self._hdr.view(torch.int8)

    def _h16(self):
        self._settle()
        return self._hdr.view(torch.int16)

    @property
    def position(self):
        """[N, 2] (row, col) in bordered coordinates; 1D: column 0 is the position."""
        return self._h8()[:, 0:2].to(torch.int64)

    @property
    def need_reset(self):
        return (self._h8()[:, 2] & _lib.FLAG_NEED_RESET) != 0

    @property
    def count_brick(self):
        return self._h16()[:, 2].to(torch.int64)

    @property
    def count_step(self):
        return self._h16()[:, 3].to(torch.int64)

    @property
    def total_brick(self):
        return self._h16()[:, 4].to(torch.int64)

    @property
    def plan_idx(self):
        return self._h16()[:, 5].to(torch.int64)

    @property
    def episode_return(self):
        return self._h16()[:, 6].to(torch.int64)

    @property
    def episode(self):
        self._settle()
        return self._episode.to(torch.int64)

    def plan(self):
        """float64 [N, ...]: the full plan of every env (reference attribute `plan`)."""
        self._sync_plans_full()
        table = torch.from_numpy(self.plans_full).to(self.device)
        return table[self.plan_idx]

    def input_plan(self):
        """float64 [N, 20, 20] (2D/3D): plan[3:23, 3:23], the reference's `input_plan`."""
        if self.kind == 1:
            return self.plan()
        return self.plan()[:, 3:23, 3:23]

    def episodic_stats(self):
        """Local sums over finished episodes: dict(episodes, return_sum, iou_fx_sum) of python ints (iou in 2^-40 units)."""
        self._settle()
        s = self._stats.sum(dim=1).tolist()
        return dict(episodes=int(s[0]), return_sum=int(s[1]), iou_fx_sum=int(s[2]))

    def stats_tensor(self, out=None):
        """int64 [3] on device: [episodes, return_sum, iou_fx_sum]; what snac_amd.dist all-reduces.  out: a preallocated int64 [3]
        tensor on this device that receives the sums (one kernel instead of sum + copy)."""
        self._settle()
        if out is None:
            return self._stats.sum(dim=1)
        return torch.sum(self._stats, dim=1, out=out)
