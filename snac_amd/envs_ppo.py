"""The env copies under script/PPO of the reference (stable-baselines PPO2 wants gym spaces, flat observations and a
4-tuple step): same HIP path as snac_amd.envs; the flat layout (raw counters, the plan appended by the dataset classes) is
written by the kernels (snac_env_desc.obs_scalars / obs_tail, BatchedDMPEnv(layout="ppo")), the two `>` termination tests are
rule bits of the kernel (SNAC_RULE_BRICK_GT / SNAC_RULE_TIME_GT in include/snac_hip.h).

  script/PPO/1d_static/DMP_Env_1D_static.py                            obs (7,)
  script/PPO/1d_dynamic/DMP_Env_1D_dynamic_usedata_plan.py             obs (37,)  [window 5, count_brick, count_step, plan 30]; `>` brick test (:93)
  script/PPO/2d_static/DMP_Env_2D_static.py                            obs (51,); `>` brick test (:137)
  script/PPO/2d_dynamic/DMP_Env_2d_dynamic_usedata_plan.py             obs (451,) [window 49, count_brick, count_step, input_plan 400]
  script/PPO/3d_static/DMP_simulator_3d_static_circle.py               obs (51,); `>` brick test (:205) and `>` time test (:221)
  script/PPO/3d_dynamic/DMP_simulator_3d_dynamic_triangle_usedata.py   obs (451,)
Every class returns the raw counters and `info = {}`; randomness is consumed like the reference (np.random.randint per
step, per random-mode reset).  Import shims with the reference's module names: snac_amd/script/PPO/<variant>/.

The copies under script/SAC/environments/ (DMP_Env_1D_static.py, DMP_Env_1D_dynamic.py, DMP_Env_2D_static.py,
DMP_Env_2D_dynamic.py, DMP_simulator_3d_static_circle.py, DMP_simulator_3d_dynamic_triangle_usedata.py) are the same six
environments with a 3-tuple step() (no info) and gym spaces on the 3D pair only: the `*_sac_*` classes below, shims under
snac_amd/script/SAC/environments/.
"""
import numpy as np

from .envs import (deep_mobile_printing_1d1r_dynamic, deep_mobile_printing_1d1r_static, deep_mobile_printing_2d1r_dynamic,
                   deep_mobile_printing_2d1r_static, deep_mobile_printing_3d1r_dynamic, deep_mobile_printing_3d1r_static)
from .envs_mcts import Discrete

try:
    from gym.spaces import Box
except Exception:  # pragma: no cover - gym is optional

    class Box(object):
        """low / high / shape / dtype of gym.spaces.Box (what stable-baselines reads from observation_space)."""

        def __init__(self, low, high, dtype=None):
            self.low, self.high, self.dtype = np.asarray(low), np.asarray(high), dtype
            self.shape = self.low.shape

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class _PPO(object):
    """Mixin: flat observation [window, count_brick, count_step (, plan)], 4-tuple step."""
    _layout = dict(obs_scalars="raw")

    def _spaces(self, window, plan_cells=0, plan_high=1):
        self.action_space = Discrete(self.action_dim)
        low = [-1] * window + [0, 0] + [0] * plan_cells
        high = [99] * window + [self.total_step, self.total_step] + [plan_high] * plan_cells
        self.observation_space = Box(low=np.array(low), high=np.array(high), dtype=int)

    def _flat(self, obs):
        return np.asarray(obs, np.float64).reshape(-1)           # the row as the kernel wrote it


class deep_mobile_printing_1d1r_ppo_static(_PPO, deep_mobile_printing_1d1r_static):
    """script/PPO/1d_static/DMP_Env_1D_static.py :: deep_mobile_printing_1d1r(plan_choose=0)"""

    def __init__(self, plan_choose=0):
        deep_mobile_printing_1d1r_static.__init__(self, plan_choose)
        self._spaces(5)
        low, high = self.observation_space.low.copy(), self.observation_space.high
        low[2:5] = 0                                             # the reference's Box: [-1, -1, 0, 0, 0, 0, 0] (:30)
        self.observation_space = Box(low=low, high=high, dtype=int)

    def reset(self):
        return self._flat(deep_mobile_printing_1d1r_static.reset(self))

    def step(self, action):
        obs, reward, done = deep_mobile_printing_1d1r_static.step(self, action)
        return self._flat(obs), reward, done, {}


class deep_mobile_printing_1d1r_ppo_dynamic(_PPO, deep_mobile_printing_1d1r_dynamic):
    """script/PPO/1d_dynamic/DMP_Env_1D_dynamic_usedata_plan.py :: deep_mobile_printing_1d1r(data_path, random_choose_paln=True)"""
    _layout = dict(obs_scalars="raw", obs_tail=("plan",))
    _brick_gt = True

    def __init__(self, data_path, random_choose_paln=True):
        deep_mobile_printing_1d1r_dynamic.__init__(self, data_path, random_choose_paln)
        self._spaces(5, 30, 42)
        low = self.observation_space.low.copy()
        low[2:5] = 0
        self.observation_space = Box(low=low, high=self.observation_space.high, dtype=int)

    def reset(self):
        return self._flat(deep_mobile_printing_1d1r_dynamic.reset(self)[1])

    def step(self, action):
        obs, reward, done = deep_mobile_printing_1d1r_dynamic.step(self, action)
        return self._flat(obs[1]), reward, done, {}


class deep_mobile_printing_2d1r_ppo_static(_PPO, deep_mobile_printing_2d1r_static):
    """script/PPO/2d_static/DMP_Env_2D_static.py :: deep_mobile_printing_2d1r(plan_choose=0); spells count_brick `conut_brick`"""
    _brick_gt = True

    def __init__(self, plan_choose=0):
        deep_mobile_printing_2d1r_static.__init__(self, plan_choose)
        self._spaces(49)

    def _set_cb(self, cb):
        self.count_brick = cb
        self.conut_brick = cb

    def reset(self):
        return self._flat(deep_mobile_printing_2d1r_static.reset(self))

    def step(self, action):
        obs, reward, done = deep_mobile_printing_2d1r_static.step(self, action)
        return self._flat(obs), reward, done, {}


class deep_mobile_printing_2d1r_ppo_dynamic(_PPO, deep_mobile_printing_2d1r_dynamic):
    """script/PPO/2d_dynamic/DMP_Env_2d_dynamic_usedata_plan.py :: deep_mobile_printing_2d1r(data_path, random_choose_paln=True)"""
    _layout = dict(obs_scalars="raw", obs_tail=("plan",))

    def __init__(self, data_path, random_choose_paln=True):
        deep_mobile_printing_2d1r_dynamic.__init__(self, data_path, random_choose_paln)
        self._spaces(49, 400, 1)

    def reset(self):
        return self._flat(deep_mobile_printing_2d1r_dynamic.reset(self)[0])

    def step(self, action):
        obs, reward, done = deep_mobile_printing_2d1r_dynamic.step(self, action)
        return self._flat(obs[0]), reward, done, {}


class deep_mobile_printing_3d1r_ppo_static(_PPO, deep_mobile_printing_3d1r_static):
    """script/PPO/3d_static/DMP_simulator_3d_static_circle.py :: deep_mobile_printing_3d1r(plan_choose=1)"""
    _brick_gt = True
    _time_gt = True

    def __init__(self, plan_choose=1):
        deep_mobile_printing_3d1r_static.__init__(self, plan_choose)
        self._spaces(49)

    def reset(self):
        return self._flat(deep_mobile_printing_3d1r_static.reset(self))

    def step(self, action):
        obs, reward, done = deep_mobile_printing_3d1r_static.step(self, action)
        return self._flat(obs), reward, done, {}


class deep_mobile_printing_3d1r_ppo_dynamic(_PPO, deep_mobile_printing_3d1r_dynamic):
    """script/PPO/3d_dynamic/DMP_simulator_3d_dynamic_triangle_usedata.py :: deep_mobile_printing_3d1r(data_path, random_choose_paln=True)"""
    _layout = dict(obs_scalars="raw", obs_tail=("plan",))

    def __init__(self, data_path, random_choose_paln=True):
        deep_mobile_printing_3d1r_dynamic.__init__(self, data_path, random_choose_paln)
        self._spaces(49, 400, 1)

    def reset(self):
        return self._flat(deep_mobile_printing_3d1r_dynamic.reset(self)[0])

    def step(self, action):
        obs, reward, done = deep_mobile_printing_3d1r_dynamic.step(self, action)
        return self._flat(obs[0]), reward, done, {}


# ---- script/SAC/environments: the same six, step() -> (obs, reward, done) ---------------------------------------------------
def _sac(base, spaces):
    class _SAC(base):
        __doc__ = "script/SAC/environments copy of %s: 3-tuple step()%s" % (base.__name__, "" if spaces else ", no gym spaces")

        def __init__(self, *args, **kw):
            base.__init__(self, *args, **kw)
            if not spaces:
                del self.action_space, self.observation_space

        def step(self, action):
            return base.step(self, action)[:3]

    _SAC.__name__ = _SAC.__qualname__ = base.__name__.replace("_ppo_", "_sac_")
    return _SAC


deep_mobile_printing_1d1r_sac_static = _sac(deep_mobile_printing_1d1r_ppo_static, False)
deep_mobile_printing_1d1r_sac_dynamic = _sac(deep_mobile_printing_1d1r_ppo_dynamic, False)
deep_mobile_printing_2d1r_sac_static = _sac(deep_mobile_printing_2d1r_ppo_static, False)
deep_mobile_printing_2d1r_sac_dynamic = _sac(deep_mobile_printing_2d1r_ppo_dynamic, False)
deep_mobile_printing_3d1r_sac_static = _sac(deep_mobile_printing_3d1r_ppo_static, True)
deep_mobile_printing_3d1r_sac_dynamic = _sac(deep_mobile_printing_3d1r_ppo_dynamic, True)
