"""numpy statement of the build's counter-based generator (include/snac_hip.h, "Counter RNG").

Used by the tests and by tests/golden/make_golden.py to produce action streams that the C oracle
and the HIP kernels can regenerate from (seed, global env id, tick) alone.  Test infrastructure only.
"""
import numpy as np

M = np.uint64(0xFFFFFFFF)
GOLD = np.uint64(0x9E3779B9)
STREAM_STEP = 0   # one word per (env, tick): action from the high 16 bits, step size from the low 16
STREAM_PLAN = 1   # one word per (env, episode): plan index


def mix32(x):
    x = np.asarray(x, np.uint64) & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    return x


def stream_key(seed, stream):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    return mix32(lo ^ mix32((hi + GOLD * np.uint64(stream + 1)) & M))


def env_keys(key, env):
    env = np.asarray(env, np.uint64)
    elo, ehi = env & M, env >> np.uint64(32)
    e0 = mix32(key ^ mix32((elo + np.uint64(0x85EBCA6B) * ehi + np.uint64(0x1B873593)) & M))
    e1 = mix32(((key + np.uint64(0x27D4EB2F)) & M) ^ mix32(((elo ^ np.uint64(0x165667B1)) + np.uint64(0xC2B2AE35) * ehi) & M))
    return e0, e1


def words(seed, stream, env, t):
    """32-bit word for (seed, stream, env id, counter t); env and t broadcast."""
    e0, e1 = env_keys(stream_key(seed, stream), env)
    t = np.asarray(t, np.uint64) & M
    return mix32((mix32(e0 ^ ((GOLD * t) & M)) + e1) & M)


def action_of(w, num_actions):
    return (((w >> np.uint64(16)) * np.uint64(num_actions)) >> np.uint64(16)).astype(np.int8)


def step_size_of(w):
    return (np.uint64(1) + (((w & np.uint64(0xFFFF)) * np.uint64(3)) >> np.uint64(16))).astype(np.int8)


def plan_of(w, num_plans):
    return ((w * np.uint64(num_plans)) >> np.uint64(32)).astype(np.int32)


def counter_actions(seed, env, n_steps, num_actions, t0=0):
    """Actions of one env for ticks t0..t0+n_steps-1."""
    w = words(seed, STREAM_STEP, np.uint64(env), np.arange(t0, t0 + n_steps, dtype=np.uint64))
    return action_of(w, num_actions)


def counter_step_sizes(seed, env, n_steps, t0=0):
    w = words(seed, STREAM_STEP, np.uint64(env), np.arange(t0, t0 + n_steps, dtype=np.uint64))
    return step_size_of(w)
