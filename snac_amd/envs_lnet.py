"""L-Net env variants of the reference (used by script/Representation_learning/*): static plans with a different
observation layout per dimension.  Same HIP path as snac_amd.envs; the layouts are flags of the kernels
(snac_env_desc.frame_value / obs_scalars / obs_tail, BatchedDMPEnv(layout="lnet1d" | "lnet2d")), nothing is rearranged on the host.

  Env/1D/DMP_Env_1D_static_Lnet.py                 position appended to the observation, shape (1, 8)
  Env/2D/DMP_Env_2D_static_Lnet.py                 frame cells hold 2 instead of -1, normalised scalars, [obs, position]
  Env/3D/DMP_simulator_3d_static_circle_Lnet.py    the dynamic class's rules with total_step 1300, [obs, position]
(the *_Lnet_test.py files differ from these in render() only).
"""
import numpy as np

from . import plans as _plans
from .envs import (_Env3D, _EnvGrid, _Facade, deep_mobile_printing_1d1r_static, deep_mobile_printing_2d1r_static,
                   deep_mobile_printing_3d1r_static)


class deep_mobile_printing_1d1r_lnet(deep_mobile_printing_1d1r_static):
    """Env/1D/DMP_Env_1D_static_Lnet.py :: deep_mobile_printing_1d1r (:83, :112, :127, :133): observation (1, 8)"""
    _layout = dict(obs_tail=("position",))

    def step(self, action):
        return deep_mobile_printing_1d1r_static.step(self, action)


class deep_mobile_printing_2d1r_lnet(_EnvGrid):
    """Env/2D/DMP_Env_2D_static_Lnet.py :: deep_mobile_printing_2d1r (:61-64 frame value 2, :75-76 return layout)"""
    _dim, _dynamic = 2, False
    _layout = dict(frame_value=2, obs_scalars="norm")

    def __init__(self, plan_choose=0):
        self._init_grid()
        self.total_step = 600
        self.action_dim = 5
        self.plan_choose = plan_choose
        self._err = None
        if plan_choose not in (0, 1):
            self._err = ValueError('0: Dense circle, 1: Sparse circle')
            return
        self._setup(_plans.static_plan(2, plan_choose)[None])

    create_plan = deep_mobile_printing_2d1r_static.create_plan

    def reset(self):
        self.create_plan()
        obs, pos = self._grid_reset(0)
        return [obs, pos]

    def step(self, action):
        obs, reward, done, pos = self._grid_step(action)
        return [obs, pos], reward, done


class deep_mobile_printing_3d1r_lnet(_Env3D):
    """Env/3D/DMP_simulator_3d_static_circle_Lnet.py :: deep_mobile_printing_3d1r (:28 total_step, :210-236 rules)"""
    _dynamic = True

    def __init__(self, plan_choose=1):
        self._init_3d()
        self.total_step = 1300
        self.plan_choose = plan_choose
        self._err = None
        if plan_choose not in (0, 1):
            self._err = ValueError('0: Dense circle, 1: Sparse circle')
            return
        self._setup(_plans.static_plan(3, plan_choose)[None], total_step=1300)

    create_plan = deep_mobile_printing_3d1r_static.create_plan

    def reset(self):
        self.create_plan()
        self.check = []
        self.step_size = 1
        obs, pos = self._grid_reset(0)
        return [obs, pos]

    def step(self, action):
        obs, reward, done, pos = self._grid_step(action)
        return [obs, pos], reward, done
