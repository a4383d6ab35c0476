"""MCTS env variants of the reference (used by script/MCTS/*): the canonical dynamics with
  * reset() -> (state, observation) and step() -> (state, observation, reward, done),
    state = (position, environment_memory copy, count_brick, count_step);
  * raw count_brick / count_step in the observation, also in the dataset ("dynamic") classes;
  * the functional transition(state, action, is_model_dynamic) that script/MCTS/utils/mcts_Qvalue*.py calls once per tree
    edge, equality_operator(o1, o2) and action_space.

  Env/1D/DMP_Env_1D_static_MCTS.py                       deep_mobile_printing_1d1r_MCTS            transition() edits the grid it is given
  Env/1D/DMP_Env_1D_static_MCTS_test.py                  deep_mobile_printing_1d1r_MCTS_obs_test   ... copies it (:96-97)
  Env/1D/DMP_Env_1D_dynamic_MCTS.py                      deep_mobile_printing_1d1r_MCTS_obs        ... copies it (:85-86)
  Env/2D/DMP_ENV_2D_static_MCTS{,_test}.py               deep_mobile_printing_2d1r_MCTS{,_test}    edits
  Env/2D/DMP_ENV_2D_dynamic_MCTS.py                      deep_mobile_printing_2d1r                 edits
  Env/3D/DMP_simulator_3d_static_circle_MCTS{,_test}.py  deep_mobile_printing_3d1r                 edits
  Env/3D/DMP_simulator_3d_dynamic_triangle_MCTS.py       deep_mobile_printing_3d1r                 edits; its brick-limit test reads the
                                                         ENV's count_brick, not the state's (:258) -- reproduced here
(the *_test.py files differ in render() only, and 1D in the copy).  Same HIP path as snac_amd.envs: step() is snac_step,
transition() is snac_import_state + snac_transition on a one-row node pool.  For tree search at scale use
BatchedDMPEnv.transition() directly (thousands of edges per launch).  Randomness: every step() and transition() draws
np.random.randint(1, 4) on the host, like the reference.
"""
import numpy as np

from .envs import (deep_mobile_printing_1d1r_dynamic, deep_mobile_printing_1d1r_static, deep_mobile_printing_2d1r_dynamic,
                   deep_mobile_printing_2d1r_static, deep_mobile_printing_3d1r_dynamic, deep_mobile_printing_3d1r_static)

try:  # spaces.Discrete(action_dim) when gym is installed (Env/2D/DMP_ENV_2D_dynamic_MCTS.py:29)
    from gym.spaces import Discrete
except Exception:  # pragma: no cover - gym is optional

    class Discrete(object):
        """What script/MCTS/utils/uct.py needs of gym.spaces.Discrete: n, sample(), contains()."""

        def __init__(self, n):
            self.n = int(n)

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return "Discrete(%d)" % self.n


class _MCTS(object):
    """Mixin over one of the canonical facades (which provides _do_reset / _do_step on self._env)."""
    _edits = True      # transition() writes the new grid into the array it was given
    _gated = False     # 3D dynamic: brick-limit test on the env's own count_brick

    def _mcts_init(self):
        from .batched import BatchedDMPEnv

        self.action_space = Discrete(self.action_dim)
        self.state = None
        self._leaf = BatchedDMPEnv(self._dim, self._dynamic, 1, plans=self._table, total_step=self.total_step)

    def _cb(self):
        return self.conut_brick if self._dim == 1 else self.count_brick

    def _raw(self, obs, cb, cs):
        obs = np.array(obs, np.float64).reshape(1, -1)
        obs[0, -2], obs[0, -1] = cb, cs                       # raw counts, also where the canonical class normalises
        return obs

    def _snapshot(self, pos):
        self.state = (pos, self.environment_memory.copy(), self._cb(), self.count_step)
        return self.state

    def _pos(self, r, c):
        return int(r) if self._dim == 1 else [int(r), int(c)]

    # ---- reference surface ---------------------------------------------------------------------------
    def _mcts_reset(self, plan_idx):
        obs, (r, c) = self._do_reset(plan_idx)
        self._plan_row = int(plan_idx)
        return self._snapshot(self._pos(r, c)), self._raw(obs, 0, 0), (r, c)

    def _mcts_step(self, action):
        obs, reward, done, (r, c) = self._do_step(action)
        return self._snapshot(self._pos(r, c)), self._raw(obs, self._cb(), self.count_step), reward, done, (r, c)

    def transition(self, state, action, is_model_dynamic=True):
        import torch

        step_size = int(np.random.randint(1, 4))
        position, memory, count_brick, count_step = state
        a = int(action)
        tb = int(self.total_brick)
        if self._gated:                                        # :258 `elif self.count_brick >= self.total_brick`
            tb = -32768 if self._cb() >= self.total_brick else 32767
        leaf = self._leaf
        leaf.import_states(np.asarray([position]).reshape(1, -1)[:, :2] if self._dim != 1 else [int(position)], [int(count_brick)],
                           [int(count_step)], np.asarray(memory, np.float64)[None], plan_idx=[self._plan_row], total_brick=[tb])
        obs, reward, done = leaf.transition(torch.tensor([a if -128 <= a <= 127 else 127], dtype=torch.int8),
                                            torch.tensor([step_size], dtype=torch.int8))
        new = leaf.environment_memory()[0].cpu().numpy().reshape(np.shape(memory))
        r, c = (int(v) for v in leaf.position[0].tolist())
        cb, cs = int(leaf.count_brick[0]), int(leaf.count_step[0])
        if self._edits:
            memory[...] = new                                  # the reference edits the caller's array (no copy)
            new = memory
        return (self._pos(r, c), new, cb, cs), self._raw(obs.cpu().numpy(), cb, cs), float(reward.item()), bool(done.item())

    def equality_operator(self, o1, o2):
        return bool(np.array_equal(o1, o2))


# ================================================================================================ 1D
class _Static1DMCTS(_MCTS, deep_mobile_printing_1d1r_static):
    def __init__(self, plan_choose=0):
        deep_mobile_printing_1d1r_static.__init__(self, plan_choose)
        if self._err is None:
            self._mcts_init()

    def reset(self):
        self.one_hot = None
        self.create_plan()
        state, obs, (r, _) = self._mcts_reset(0)
        self.total_brick = float(self.total_brick)
        self.brick_memory = [[-1, -1]]
        self.position_memory = [r]
        return state, obs

    def step(self, action):
        state, obs, reward, done, (r, _) = self._mcts_step(action)
        self._after_step(action, r, obs)
        return state, obs, reward, done


class deep_mobile_printing_1d1r_MCTS(_Static1DMCTS):
    """Env/1D/DMP_Env_1D_static_MCTS.py :: deep_mobile_printing_1d1r_MCTS(plan_choose=0)"""

    def iou_MCTS(self, environment_memory):
        """:250-263 -- iou() of an arbitrary grid against the env's plan"""
        self._leaf.import_states([2], [0], [0], np.asarray(environment_memory, np.float64)[None], plan_idx=[self._plan_row])
        return float(self._leaf.iou().item())


class deep_mobile_printing_1d1r_MCTS_obs_test(_Static1DMCTS):
    """Env/1D/DMP_Env_1D_static_MCTS_test.py :: deep_mobile_printing_1d1r_MCTS_obs_test(plan_choose=0); transition() works on
    a copy of the grid (:96-97) and there is no iou_MCTS"""
    _edits = False


class deep_mobile_printing_1d1r_MCTS_obs(_MCTS, deep_mobile_printing_1d1r_dynamic):
    """Env/1D/DMP_Env_1D_dynamic_MCTS.py :: deep_mobile_printing_1d1r_MCTS_obs(data_path, random_choose_paln=True)"""
    _edits = False

    def __init__(self, data_path, random_choose_paln=True):
        deep_mobile_printing_1d1r_dynamic.__init__(self, data_path, random_choose_paln)
        self._mcts_init()

    def reset(self):
        state, obs, (r, _) = self._mcts_reset(self._pick_plan())
        self.total_brick = float(self.total_brick)
        self.brick_memory = [[-1, -1]]
        self.position_memory = [r]
        return state, obs

    def step(self, action):
        state, obs, reward, done, (r, _) = self._mcts_step(action)
        self._after_step(action, r, obs)
        return state, obs, reward, done


# ================================================================================================ 2D / 3D
class _GridMCTS(_MCTS):
    def _grid_mcts_reset(self, plan_idx):
        state, obs, (r, c) = self._mcts_reset(plan_idx)
        h = self.HALF_WINDOW_SIZE
        self.input_plan = self.plan[h:h + self.plan_height, h:h + self.plan_width]
        self.position_memory = [[r, c]]
        return state, obs

    def step(self, action):
        state, obs, reward, done, (r, c) = self._mcts_step(action)
        self.position_memory.append([r, c])
        return state, obs, reward, done

    def observation_transition(self, environment_memory, position):
        h = self.HALF_WINDOW_SIZE
        return environment_memory[position[0] - h:position[0] + h + 1, position[1] - h:position[1] + h + 1].flatten().reshape(1, -1)


class deep_mobile_printing_2d1r_MCTS(_GridMCTS, deep_mobile_printing_2d1r_static):
    """Env/2D/DMP_ENV_2D_static_MCTS.py :: deep_mobile_printing_2d1r_MCTS(plan_choose=0)"""

    def __init__(self, plan_choose=0):
        deep_mobile_printing_2d1r_static.__init__(self, plan_choose)
        if self._err is None:
            self._mcts_init()

    def reset(self):
        self.create_plan()
        state, obs = self._grid_mcts_reset(0)
        self.total_brick = float(self.total_brick)
        return state, obs


class deep_mobile_printing_2d1r_MCTS_test(deep_mobile_printing_2d1r_MCTS):
    """Env/2D/DMP_ENV_2D_static_MCTS_test.py (render() differs)"""


class deep_mobile_printing_2d1r_MCTS_dynamic(_GridMCTS, deep_mobile_printing_2d1r_dynamic):
    """Env/2D/DMP_ENV_2D_dynamic_MCTS.py :: deep_mobile_printing_2d1r(data_path, random_choose_paln=True)"""

    def __init__(self, data_path, random_choose_paln=True):
        deep_mobile_printing_2d1r_dynamic.__init__(self, data_path, random_choose_paln)
        self._mcts_init()

    def reset(self):
        state, obs = self._grid_mcts_reset(self._pick_plan())
        self.total_brick = float(self.total_brick)
        return state, obs


class _Grid3DMCTS(_GridMCTS):
    def check_sur_trainsition(self, environment_memory, position):     # the reference's spelling (:95)
        g = environment_memory
        check = [0] * 8
        nb = [g[position[0], position[1] - 1], g[position[0], position[1] + 1], g[position[0] + 1, position[1]],
              g[position[0] - 1, position[1]]]
        for i, v in enumerate(nb):
            if v == -1:
                check[i] = 1
                check[i + 4] = 1
            elif v > 0:
                check[i] = 1
        return check

    def reward_check_transition(self, environment_memory, position):
        g = environment_memory
        if g[position[0], position[1]] > self.plan[position[0], position[1]]:
            return -1.0
        if g[position[0], position[1]] == self.plan[position[0], position[1]]:
            return 10.0
        return 1.0


class deep_mobile_printing_3d1r_MCTS(_Grid3DMCTS, deep_mobile_printing_3d1r_static):
    """Env/3D/DMP_simulator_3d_static_circle_MCTS{,_test}.py :: deep_mobile_printing_3d1r(plan_choose=1)"""

    def __init__(self, plan_choose=1):
        deep_mobile_printing_3d1r_static.__init__(self, plan_choose)
        if self._err is None:
            self._mcts_init()

    def reset(self):
        self.create_plan()
        self.check = []
        self.step_size = 1
        return self._grid_mcts_reset(0)


class deep_mobile_printing_3d1r_MCTS_dynamic(_Grid3DMCTS, deep_mobile_printing_3d1r_dynamic):
    """Env/3D/DMP_simulator_3d_dynamic_triangle_MCTS.py :: deep_mobile_printing_3d1r(data_path, random_choose_paln=True)"""
    _gated = True

    def __init__(self, data_path, random_choose_paln=True):
        deep_mobile_printing_3d1r_dynamic.__init__(self, data_path, random_choose_paln)
        self._mcts_init()

    def reset(self):
        idx = self._pick_plan()
        self.check = []
        self.step_size = 1
        return self._grid_mcts_reset(idx)
