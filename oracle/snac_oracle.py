"""ctypes binding of oracle/libsnac_oracle.so.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (snac_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("SNAC_ORACLE_LIB") or os.path.join(HERE, "libsnac_oracle.so")   # override: the sanitizer build (oracle/Makefile)
MAX_CELLS = 676


def build(force=False):
    src = [os.path.join(HERE, f) for f in ("snac_oracle.c", "snac_oracle.h")]
    if os.environ.get("SNAC_ORACLE_LIB"):
        return LIB
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.check_call(["make", "-s", "-C", HERE, "libsnac_oracle.so"])
    return LIB


class _Env(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "dynamic", "hw", "H", "W", "total_step", "num_actions", "obs_dim")] + [
        ("grid", C.c_int32 * MAX_CELLS), ("plan", C.c_int32 * MAX_CELLS), ("pos", C.c_int32 * 2),
        ("cb", C.c_int32), ("cs", C.c_int32), ("tb", C.c_int32), ("step_size", C.c_int32), ("plan_idx", C.c_int32),
        ("obs_norm", C.c_int32), ("rules_dyn", C.c_int32), ("frame", C.c_int32), ("brick_gt", C.c_int32), ("time_gt", C.c_int32),
        ("tail", C.c_int32), ("last_reward", C.c_int32), ("last_done", C.c_int32)]


class _Batch(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "dynamic", "n", "num_plans", "cells", "obs_dim", "total_step", "num_actions")] + [
        ("seed", C.c_uint64), ("env_id_base", C.c_int64), ("plans", C.POINTER(C.c_int32)), ("envs", C.POINTER(_Env)),
        ("episode", C.POINTER(C.c_int32)), ("ep_return", C.POINTER(C.c_int32)), ("need_reset", C.POINTER(C.c_uint8)),
        ("stat_episodes", C.POINTER(C.c_int64)), ("stat_return", C.POINTER(C.c_int64)),
        ("stat_iou_fx", C.POINTER(C.c_int64)), ("stat_steps", C.POINTER(C.c_int64))]


class _MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.orc_init.argtypes = [C.POINTER(_Env), C.c_int, C.c_int]
        L.orc_configure.argtypes = [C.POINTER(_Env), C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_reset.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_int, C.c_void_p]
        L.orc_step.argtypes = [C.POINTER(_Env), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_observe.argtypes = [C.POINTER(_Env), C.c_void_p]
        L.orc_iou.argtypes = [C.POINTER(_Env)]
        L.orc_iou.restype = C.c_double
        L.orc_static_plan.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.orc_mt_seed.argtypes = [C.POINTER(_MT), C.c_uint32]
        L.orc_mt_next.argtypes = [C.POINTER(_MT)]
        L.orc_mt_next.restype = C.c_uint32
        L.orc_mt_randint.argtypes = [C.POINTER(_MT), C.c_int64, C.c_int64]
        L.orc_mt_randint.restype = C.c_int64
        L.orc_rng_word.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]
        L.orc_rng_word.restype = C.c_uint32
        L.orc_batch_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_int64]
        L.orc_batch_create.restype = C.POINTER(_Batch)
        L.orc_batch_destroy.argtypes = [C.POINTER(_Batch)]
        L.orc_batch_reset.argtypes = [C.POINTER(_Batch), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_batch_step.argtypes = [C.POINTER(_Batch), C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int]
        L.orc_batch_rollout.argtypes = [C.POINTER(_Batch), C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_int]
        L.orc_batch_iou.argtypes = [C.POINTER(_Batch), C.c_void_p]
        L.orc_set_rules.argtypes = [C.POINTER(_Env), C.c_int, C.c_int]
        L.orc_batch_set_rules.argtypes = [C.POINTER(_Batch), C.c_int, C.c_int]
        L.orc_set_tail.argtypes = [C.POINTER(_Env), C.c_int]
        L.orc_batch_configure.argtypes = [C.POINTER(_Batch), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_transition.argtypes = [C.POINTER(_Env), C.POINTER(_Env), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_set_state.argtypes = [C.POINTER(_Env), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_batch_transition.argtypes = [C.POINTER(_Batch), C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_raster_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_make_plan.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def static_plan(dim, plan_choose):
    out = np.zeros(MAX_CELLS, np.int32)
    n = lib().orc_static_plan(dim, plan_choose, _ptr(out))
    if n < 0:
        raise ValueError("no such static plan")
    return out[:n].copy()


def make_plans(dim, sparse, seed, first_id, count):
    """Plans first_id .. first_id + count - 1 of the generator specification -> (table [count, 30 | 676] int32, total_brick [count])."""
    cells = 30 if dim == 1 else 676
    table = np.zeros((count, cells), np.int32)
    tb = np.zeros(count, np.int32)
    for i in range(count):
        lib().orc_make_plan(dim, int(sparse), seed, first_id + i, _ptr(table[i]), C.c_void_p(tb[i:].ctypes.data))
    return table, tb


def raster_triangle(vx, vy, sparse):
    """One rasterisation of the specification's triangle rasteriser -> (mask [20, 20] int32, area)."""
    x = np.ascontiguousarray(vx, np.int32)
    y = np.ascontiguousarray(vy, np.int32)
    img = np.zeros(400, np.int32)
    area = lib().orc_raster_triangle(_ptr(x), _ptr(y), int(sparse), _ptr(img))
    return img.reshape(20, 20), int(area)


class MT19937:
    """np.random.seed(s) / np.random.randint(lo, hi) of numpy's legacy global stream."""

    def __init__(self, seed):
        self.m = _MT()
        lib().orc_mt_seed(C.byref(self.m), seed)

    def randint(self, lo, hi):
        return int(lib().orc_mt_randint(C.byref(self.m), lo, hi))


class OracleEnv:
    """One env; mirrors the reference class surface that the goldens exercise."""

    def __init__(self, dim, dynamic):
        self.e = _Env()
        if lib().orc_init(C.byref(self.e), dim, int(dynamic)):
            raise ValueError("bad dim")
        self.obs_dim = self.e.obs_dim

    def configure(self, obs_norm, rules_dyn, total_step=0, frame=-1):
        lib().orc_configure(C.byref(self.e), int(obs_norm), int(rules_dyn), int(total_step), int(frame))
        return self

    def set_rules(self, brick_gt=False, time_gt=False):
        lib().orc_set_rules(C.byref(self.e), int(brick_gt), int(time_gt))
        return self

    def set_tail(self, tail):
        """ORC_TAIL_* bits (1 position, 2 plan, 4 record) appended to every observation row."""
        self.obs_dim = lib().orc_set_tail(C.byref(self.e), int(tail))
        return self

    def reset(self, plan, plan_idx=0):
        plan = np.ascontiguousarray(np.asarray(plan).reshape(-1), np.int32)
        assert plan.size == (30 if self.e.dim == 1 else 676)
        obs = np.zeros(self.obs_dim, np.float64)
        lib().orc_reset(C.byref(self.e), _ptr(plan), plan_idx, _ptr(obs))
        return obs

    def step(self, action, k):
        obs = np.zeros(self.obs_dim, np.float64)
        r, d = C.c_double(0), C.c_int(0)
        rc = lib().orc_step(C.byref(self.e), int(action), int(k), _ptr(obs), C.byref(r), C.byref(d))
        if rc:
            raise ValueError("bad action %d (rc=%d)" % (action, rc))
        return obs, r.value, bool(d.value)

    def iou(self):
        return lib().orc_iou(C.byref(self.e))

    def set_state(self, grid, pos, cb, cs):
        """Load an explicit (position, environment_memory, count_brick, count_step) tuple (after reset(): the plan stays)."""
        g = np.ascontiguousarray(np.asarray(grid).reshape(-1), np.int32)
        assert g.size == self.e.H * self.e.W
        r, c = (int(pos), 0) if np.ndim(pos) == 0 else (int(pos[0]), int(pos[1]))
        if lib().orc_set_state(C.byref(self.e), _ptr(g), r, c, int(cb), int(cs)):
            raise ValueError("bad position")
        return self

    def transition(self, action, k, gate_cb=-1, inplace=False):
        """MCTS transition(state, action): -> (new OracleEnv, obs, reward, done); this env is left untouched unless inplace."""
        dst = self
        if not inplace:
            dst = OracleEnv.__new__(OracleEnv)
            dst.e, dst.obs_dim = _Env(), self.obs_dim
        obs = np.zeros(self.obs_dim, np.float64)
        r, d = C.c_double(0), C.c_int(0)
        if lib().orc_transition(C.byref(self.e), C.byref(dst.e), int(action), int(k), int(gate_cb), _ptr(obs), C.byref(r), C.byref(d)):
            raise ValueError("bad action %d" % action)
        return dst, obs, r.value, bool(d.value)

    @property
    def grid(self):
        n = self.e.H * self.e.W
        return np.array(self.e.grid[:n], np.int32)

    @property
    def pos(self):
        return (self.e.pos[0], self.e.pos[1])


class OracleBatch:
    """N independent envs with the batched semantics of include/snac_hip.h."""

    def __init__(self, dim, dynamic, n, plans, seed=1, env_id_base=0):
        self.plans = np.ascontiguousarray(np.asarray(plans).reshape(len(plans), -1), np.int32)
        assert self.plans.shape[1] == (30 if dim == 1 else 676)
        self.b = lib().orc_batch_create(dim, int(dynamic), n, _ptr(self.plans), len(self.plans), seed, env_id_base)
        if not self.b:
            raise MemoryError
        self.n, self.obs_dim, self.dim, self.dynamic = n, self.b.contents.obs_dim, dim, bool(dynamic)
        self.total_step, self.num_actions = self.b.contents.total_step, self.b.contents.num_actions

    def set_rules(self, brick_gt=False, time_gt=False):
        lib().orc_batch_set_rules(self.b, int(brick_gt), int(time_gt))
        return self

    def configure(self, obs_norm=None, rules_dyn=None, total_step=0, frame=-1, tail=0):
        """Layout / rule switches of every env: obs_norm (None: = dynamic), rules_dyn (None: = dynamic), total_step (0: keep),
        frame value, tail bits (1 position, 2 plan, 4 record)."""
        on = self.dynamic if obs_norm is None else obs_norm
        rd = self.dynamic if rules_dyn is None else rules_dyn
        lib().orc_batch_configure(self.b, int(on), int(rd), int(total_step), int(frame), int(tail))
        self.obs_dim, self.total_step = self.b.contents.obs_dim, self.b.contents.total_step
        return self

    def set_total_step(self, total_step):
        """The time limit of every env (snac_env_desc.total_step)."""
        self._words()[:, _Env.total_step.offset // 4] = int(total_step)
        self.b.contents.total_step = self.total_step = int(total_step)
        return self

    def _words(self):
        """The env structs as one int32 array [n, words per struct] (every field of orc_env is an int32): a view, not a copy."""
        return np.ctypeslib.as_array(C.cast(self.b.contents.envs, C.POINTER(C.c_int32)), shape=(self.n, C.sizeof(_Env) // 4))

    def __del__(self):
        if getattr(self, "b", None):
            lib().orc_batch_destroy(self.b)
            self.b = None

    def reset(self, mask=None, plan_idx=None):
        obs = np.zeros((self.n, self.obs_dim), np.float64)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p = None if plan_idx is None else np.ascontiguousarray(plan_idx, np.int32)
        if lib().orc_batch_reset(self.b, _ptr(m), _ptr(p), _ptr(obs)):
            raise ValueError("bad plan index")
        return obs

    def step(self, t, actions=None, step_size=None, auto_reset=False, nthreads=1, want_obs=True):
        obs = np.zeros((self.n, self.obs_dim), np.float64) if want_obs else None
        rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, np.uint8)
        a = None if actions is None else np.ascontiguousarray(actions, np.int8)
        k = None if step_size is None else np.ascontiguousarray(step_size, np.int8)
        if lib().orc_batch_step(self.b, t, _ptr(a), _ptr(k), int(auto_reset), _ptr(obs), _ptr(rew), _ptr(done), nthreads):
            raise ValueError("bad action")
        return obs, rew, done

    def rollout(self, T, t0=0, actions=None, step_size=None, obs="all", nthreads=1):
        """obs: 'all' -> [T,n,D]; 'last' -> [n,D]; None."""
        o = None
        if obs == "all":
            o = np.zeros((T, self.n, self.obs_dim), np.float64)
        elif obs == "last":
            o = np.zeros((self.n, self.obs_dim), np.float64)
        rew = np.zeros((T, self.n), np.float32)
        done = np.zeros((T, self.n), np.uint8)
        a = None if actions is None else np.ascontiguousarray(actions, np.int8)
        k = None if step_size is None else np.ascontiguousarray(step_size, np.int8)
        if lib().orc_batch_rollout(self.b, T, t0, _ptr(a), _ptr(k), _ptr(o), int(obs == "last"), _ptr(rew), _ptr(done), nthreads):
            raise ValueError("bad action")
        return o, rew, done

    def iou(self):
        out = np.zeros(self.n, np.float64)
        lib().orc_batch_iou(self.b, _ptr(out))
        return out

    def transition(self, actions, step_size=None, src=None, dst=None, t=0, want_obs=True):
        """m functional transitions with the batch as node pool: env[dst[i]] <- step(env[src[i]], actions[i], k_i)."""
        m = len(src) if actions is None else len(actions)
        a = None if actions is None else np.ascontiguousarray(actions, np.int8)
        k = None if step_size is None else np.ascontiguousarray(step_size, np.int8)
        si = None if src is None else np.ascontiguousarray(src, np.int32)
        di = None if dst is None else np.ascontiguousarray(dst, np.int32)
        obs = np.zeros((m, self.obs_dim), np.float64) if want_obs else None
        rew = np.zeros(m, np.float32)
        done = np.zeros(m, np.uint8)
        if lib().orc_batch_transition(self.b, m, _ptr(si), _ptr(di), t, _ptr(a), _ptr(k), _ptr(obs), _ptr(rew), _ptr(done)):
            raise ValueError("bad action or index")
        return obs, rew, done

    def set_state(self, i, grid, pos, cb, cs, plan_idx=None):
        """Load a reference-format state into env i (plan_idx: also switch the env to that plan row, as a reset would)."""
        e = self.b.contents.envs[i]
        if plan_idx is not None:
            lib().orc_reset(C.byref(e), _ptr(self.plans[plan_idx]), int(plan_idx), None)
            if self.b.contents.episode[i] < 0:
                self.b.contents.episode[i] = 0
        g = np.ascontiguousarray(np.asarray(grid).reshape(-1), np.int32)
        r, c = (int(pos), 0) if np.ndim(pos) == 0 else (int(pos[0]), int(pos[1]))
        if lib().orc_set_state(C.byref(e), _ptr(g), r, c, int(cb), int(cs)):
            raise ValueError("bad position")
        self.b.contents.ep_return[i] = 0
        self.b.contents.need_reset[i] = 0

    def _arr(self, name, dtype):
        return np.ctypeslib.as_array(getattr(self.b.contents, name), shape=(self.n,)).astype(dtype, copy=True)

    def stats(self):
        return dict(episodes=self._arr("stat_episodes", np.int64), ret=self._arr("stat_return", np.int64),
                    iou_fx=self._arr("stat_iou_fx", np.int64), steps=self._arr("stat_steps", np.int64))

    def state(self):
        """-> dict of numpy arrays: grid [n,H*W] int32 (bordered), pos [n,2], cb, cs, tb, plan_idx, episode."""
        w = self._words()
        col = lambda f: w[:, getattr(_Env, f).offset // 4].copy()
        cells = int(w[0, _Env.H.offset // 4]) * int(w[0, _Env.W.offset // 4])
        g0, p0 = _Env.grid.offset // 4, _Env.pos.offset // 4
        grid, pos = w[:, g0:g0 + cells].copy(), w[:, p0:p0 + 2].copy()
        cb, cs, tb, pi = col("cb"), col("cs"), col("tb"), col("plan_idx")
        return dict(grid=grid, pos=pos, cb=cb, cs=cs, tb=tb, plan_idx=pi, episode=self._arr("episode", np.int32),
                    need_reset=self._arr("need_reset", np.uint8), ep_return=self._arr("ep_return", np.int32))
