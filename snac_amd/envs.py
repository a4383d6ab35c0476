"""Single-env facades with the reference's class surface, backed by the HIP path (N = 1).

Same constructor arguments, method names, return shapes / dtypes and public attributes as
  Env/1D/DMP_Env_1D_static.py, Env/1D/DMP_Env_1D_dynamic_usedata_plan.py,
  Env/2D/DMP_Env_2D_static.py, Env/2D/DMP_Env_2D_dynamic_usedata_plan.py,
  Env/3D/DMP_simulator_3d_static_circle.py, Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py
so that the DQN / DRQN style scripts run unchanged (snac_amd/Env/<dim>/ holds import shims under the
reference's module names).  Randomness is consumed exactly like the reference: every step() draws
`np.random.randint(1, 4)` and every random-mode dynamic reset() draws `np.random.randint(0, len(dataset))`
from numpy's global stream on the host, and hands the value to the kernel -- so a script that calls
`np.random.seed(s)` sees the same trajectory as with the reference.  There is no CPU fallback.
"""
import os

import numpy as np

from . import plans as _plans

try:  # the reference classes derive from gym.Env; keep that when gym is installed
    import gym as _gym

    _Base = _gym.Env
except Exception:  # pragma: no cover - gym is optional
    _Base = object

_KNOWN = {
    "data_1d_dynamic_sin_envplan_500_": (1, "sin"),
    "data_2d_dynamic_dense_envplan_500_": (2, "dense"), "data_2d_dynamic_sparse_envplan_500_": (2, "sparse"),
    "data_3d_dynamic_dense_envplan_500_": (3, "dense"), "data_3d_dynamic_sparse_envplan_500_": (3, "sparse"),
}


def _make_step_size_draw():
    """np.random.randint(1, 4) -- the draw every reference step() makes (Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:87) -- from numpy's
    GLOBAL stream, without randint's 1.8 us of argument handling: for a range of 3 numpy takes 32-bit words of the global MT19937
    and rejects (word & 3) == 3 (numpy/random/_bounded_integers: masked rejection), so the same words are taken here
    straight from the global RandomState's bit generator (0.26 us).  The generator object is the one np.random.seed() re-seeds in
    place.  Checked at import against randint itself on a scratch generator; any difference (another numpy) falls back to randint."""
    try:
        mt = np.random.mtrand
        state = [mt._rand, mt._rand._bit_generator, mt._rand._bit_generator.random_raw]
        if type(state[1]).__name__ != "MT19937":
            raise RuntimeError("the global generator is not an MT19937")

        def draw():
            # the global generator can be swapped after import (np.random.set_bit_generator, a rebound mtrand._rand): two identity tests
            # per draw (0.05 us) keep the fast path on whatever generator np.random.randint itself would use -- or leave it
            rs = mt._rand
            if rs is not state[0] or rs._bit_generator is not state[1]:
                bg = rs._bit_generator
                if type(bg).__name__ != "MT19937":
                    return int(np.random.randint(1, 4))
                state[0], state[1], state[2] = rs, bg, bg.random_raw
            raw = state[2]
            while True:
                v = raw() & 3
                if v != 3:
                    return 1 + v

        probe = np.random.RandomState(12345)
        want = [int(probe.randint(1, 4)) for _ in range(64)]
        probe = np.random.RandomState(12345)
        praw = probe._bit_generator.random_raw
        got = []
        while len(got) < 64:
            v = praw() & 3
            if v != 3:
                got.append(1 + v)
        if got != want or int(probe.randint(0, 400)) != _after(want):   # the same values, and the stream left where randint leaves it
            raise RuntimeError("stream mismatch")
        return draw
    except Exception:
        return lambda: int(np.random.randint(1, 4))


def _after(want):
    """The draw that follows 64 step-size draws on the scratch generator (the fast path must leave the stream where randint leaves it)."""
    probe = np.random.RandomState(12345)
    for _ in want:
        probe.randint(1, 4)
    return int(probe.randint(0, 400))


_draw_step_size = _make_step_size_draw()


def _load_dataset(dim, data_path):
    """joblib pickle as in the reference; when the file is absent but names one of the reference's 15 datasets,
    the converted copy shipped in snac_amd/data/plans.npz is used."""
    if os.path.exists(data_path):
        return _plans.load_plan_file(data_path)
    base = os.path.basename(data_path)
    for prefix, (d, dens) in _KNOWN.items():
        if base.startswith(prefix) and d == dim:
            split = base[len(prefix):].split(".")[0]
            return _plans.dataset(dim, dens, split)
    raise FileNotFoundError(data_path)


class _Facade(_Base):
    _dim = 0
    _dynamic = False

    _layout = {}                # extra BatchedDMPEnv layout arguments of a variant class (frame_value / obs_scalars / obs_tail)

    def _setup(self, plans_full, total_step=None):
        from .batched import BatchedDMPEnv  # imports torch; raises without a ROCm GPU

        lay = dict(self._layout)
        tail = tuple(lay.pop("obs_tail", ())) + ("record",)
        # obs_tail "record": every reset() / step() is ONE launch and ONE wait -- the row carries the
        # observation, reward, done, position and counters (SNAC_TAIL_RECORD); action / step size / plan index travel by value
        self._env = BatchedDMPEnv(self._dim, self._dynamic, 1, plans=plans_full, total_step=total_step,
                                  brick_gt=getattr(self, "_brick_gt", False), time_gt=getattr(self, "_time_gt", False),
                                  obs_tail=tail, **lay)
        self._table = np.asarray(plans_full, np.float64)
        # the row lives in page-locked host memory: the kernel writes it over the bus, the host waits for the stream -- a
        # step is one launch and one wait, no copy command (tools/facade_time.py: 22 -> 16 us per step)
        # ... and since round 5 not even a launch: the row is the env's MAILBOX (snac_mailbox_*: a resident wavefront polls a doorbell in
        # coherent page-locked memory, steps, writes the row back and acknowledges -- BatchedDMPEnv.mailbox_step).  SNAC_MAILBOX=0, or a
        # box where the mailbox cannot be had, keeps the launch path (step_scalar_wait).
        self._row = None
        if os.environ.get("SNAC_MAILBOX", "1") != "0":
            try:
                self._row = self._env.mailbox_open()
            except Exception:                                      # no coherent host memory / no second queue: the launch path it is
                self._row = None
        self._mbox = self._row is not None
        if self._row is None:
            self._row = self._env.new_host_obs()
        self._row_np = self._row.numpy()
        self._nobs = self._env.obs_dim - 8                     # the observation proper (with the variant's own tail)

    # ---- shared plumbing ---------------------------------------------------------------------------
    def _read(self, wait=True):
        """The one wait of a reset() / step() (step_scalar_wait() has waited already): -> (obs [1, n], reward, done, r, c, cb, cs, tb)."""
        if wait:
            self._env.sync()
        row = self._row_np.copy()                              # the buffer is rewritten by the next step
        rec = row[0, self._nobs:].tolist()                     # one conversion for the eight record values (integers, exact in float64)
        return row[:, :self._nobs], rec[0], rec[1] != 0.0, int(rec[2]), int(rec[3]), int(rec[4]), int(rec[5]), int(rec[6])

    def _do_reset(self, plan_idx):
        if getattr(self, "_plan_dirty", False):                # a hindsight relabel changed the device row: restore it
            self._env.set_plan_row(self._dirty_row, self._table[self._dirty_row])
            self._plan_dirty = False
        self._env.reset_scalar(plan_idx, out=self._row)
        obs, _, _, r, c, cb, cs, tb = self._read()
        self.plan = self._table[plan_idx].copy()               # a fresh array per reset, like create_plan()
        self._sent_plan = self.plan.copy()
        self._cur_plan_idx = int(plan_idx)
        self.total_brick = float(tb)
        self.count_step = 0
        self.observation = None
        self._set_cb(0)
        return obs, (r, c)

    def _do_step(self, action, step_size=None):
        if step_size is None:
            self.step_size = _draw_step_size()                 # np.random.randint(1, 4), drawn on EVERY step like the reference
        else:
            self.step_size = int(step_size)                    # hindsight variants: injected by the caller
        a = int(action)
        bad = not (0 <= a < self.action_dim) and self._dim != 3
        if self._mbox:
            self._env.mailbox_step(a if -2 ** 31 <= a < 2 ** 31 else -1, self.step_size)   # doorbell + the acknowledgement (raises without one)
        else:
            self._env.step_scalar_wait(a if -2 ** 31 <= a < 2 ** 31 else -1, self.step_size, self._row)   # launch + the step's one wait
        if bad:  # the reference leaves `position` unbound here, after count_step and the RNG have advanced
            self.count_step += 1
            raise UnboundLocalError("local variable 'position' referenced before assignment")
        obs, reward, done, r, c, cb, cs, tb = self._read(False)
        self.count_step = cs
        self._set_cb(cb)
        return obs, reward, done, (r, c)

    def _set_cb(self, cb):
        self.count_brick = cb

    def close(self):
        """gym.Env.close(): the env's resident wavefront (if one is up) is told to leave and the mailbox is freed; the object stays
        usable -- further steps take the launch path."""
        env = getattr(self, "_env", None)
        if env is not None and getattr(self, "_mbox", False):
            self._mbox = False
            row = env.new_host_obs()
            row.copy_(self._row)
            self._row, self._row_np = row, row.numpy()
            env.mailbox_close()

    def _sync_plan(self):
        """Hindsight relabelling (script/DRQN_hindsight/2d/DRQN_hindsight_2D_static.py:245-252): after reset() the
        caller overwrites `plan` (in place, or by rebinding it in 1D) with the final grid of the episode it replays.
        Push such a change to the device plan row; total_brick keeps the value computed by reset(), as in the reference."""
        cur = np.asarray(self.plan, np.float64)
        if not np.array_equal(cur, self._sent_plan):
            self._dirty_row = self._cur_plan_idx
            self._env.set_plan_row(self._dirty_row, cur)
            self._sent_plan = cur.copy()
            self._plan_dirty = True

    @property
    def environment_memory(self):
        return self._env.environment_memory()[0].cpu().numpy()

    def _pick_plan(self):
        """DMP_Env_2D_dynamic_usedata_plan.py:35-44: random index from the global stream, or sequential with wrap."""
        if self.random_choose_paln:
            self.index_random = int(np.random.randint(0, self.plan_dataset_len))
            return self.index_random
        idx = self.index_for_non_random
        self.index_for_non_random += 1
        if self.index_for_non_random == self.plan_dataset_len:
            self.index_for_non_random = 0
        return idx


# ================================================================================================ 1D
class _Env1D(_Facade):
    _dim = 1

    def _init_common(self):
        # Env/1D/DMP_Env_1D_static.py:9-28
        self.step_size = 1
        self.plan_width = 30
        self.plan_height = 20
        self.environment_height = 100
        self.conut_brick = None
        self.brick_memory = None
        self.HALF_WINDOW_SIZE = 2
        self.environment_width = self.plan_width + 2 * self.HALF_WINDOW_SIZE
        self.wall = np.ones((1, 2)) * (-1)
        self.position_memory = None
        self.observation = None
        self.count_step = 0
        self.total_step = 750
        self.plan = None
        self.total_brick = 0
        self.one_hot = None
        self.action_dim = 3
        self.state_dim = self.HALF_WINDOW_SIZE * 2 + 1 + 2

    def _set_cb(self, cb):
        self.conut_brick = cb       # the reference's spelling (DMP_Env_1D_static.py:14)
        self.count_brick = cb

    def clip_position(self, position):
        if position <= self.HALF_WINDOW_SIZE:
            return self.HALF_WINDOW_SIZE
        if position >= self.plan_width + self.HALF_WINDOW_SIZE - 1:
            return self.plan_width + self.HALF_WINDOW_SIZE - 1
        return position

    def iou(self):
        return float(self._env.iou().item())

    def _after_step(self, action, pos, obs):
        self.position_memory.append(pos)
        if action == 2:
            self.brick_memory.append([pos, float(obs[0, 2])])    # environment_memory[0, pos] = the window centre
        else:
            self.brick_memory.append([-1, -1])

    def render(self, axe, iou_min=None, iou_average=None, iter_times=1, **kw):
        axe.clear()
        axe.set_xlabel('X-axis')
        axe.set_xlim(-1, 30)
        axe.set_ylabel('Y-axis')
        axe.set_ylim(0, 50)
        x = np.arange(self.plan_width)
        iou = self.iou()
        env = self.environment_memory[0][self.HALF_WINDOW_SIZE:self.plan_width + self.HALF_WINDOW_SIZE]
        axe.title.set_text('step=%d, used_paint=%d, IOU=%.3f' % (self.count_step, self.conut_brick, iou))
        axe.plot(x, np.asarray(self.plan), color='b')
        axe.bar(x, env, color='r')
        axe.scatter(self.position_memory[-1] - self.HALF_WINDOW_SIZE, 0, color='g')


class deep_mobile_printing_1d1r_static(_Env1D):
    """Env/1D/DMP_Env_1D_static.py :: deep_mobile_printing_1d1r(plan_choose=0)"""
    _dynamic = False

    def __init__(self, plan_choose=0):
        self._init_common()
        self.plan_choose = plan_choose
        if plan_choose not in (0, 1, 2):
            self._err = ValueError('0: Sin, 1: Gaussian, 2: Step')   # the reference raises in create_plan(), i.e. at reset()
            return
        self._err = None
        self._setup(_plans.static_plan(1, plan_choose)[None])

    def create_plan(self):
        if self._err is not None:
            raise self._err
        self.one_hot = self.plan_choose
        y = _plans.static_plan(1, self.plan_choose)
        return y, sum(y)

    def reset(self):
        self.one_hot = None
        self.create_plan()
        self.one_hot = None
        obs, (r, _) = self._do_reset(0)
        self.total_brick = float(self.total_brick)
        self.brick_memory = [[-1, -1]]
        self.position_memory = [r]
        return obs

    def step(self, action, _step_size=None):
        obs, reward, done, (r, _) = self._do_step(action, _step_size)
        self._after_step(action, r, obs)
        return obs, reward, done


class deep_mobile_printing_1d1r_hindsight(deep_mobile_printing_1d1r_static):
    """Env/1D/DMP_Env_1D_static_hindsight_replay.py :: step(action, step_size) -- the caller injects the step size"""

    def step(self, action, step_size):
        self._sync_plan()
        return deep_mobile_printing_1d1r_static.step(self, action, step_size)


class deep_mobile_printing_1d1r_hindsight_dynamic(_Env1D):
    """Env/1D/DMP_Env_1D_dynamic_hindsight_replay.py :: deep_mobile_printing_1d1r_hindsight() -- a fresh random sin curve
    per reset() (create_plan :29-42, plans.random_sin_plan), raw counters, observation [obs, plan], step(action, step_size)"""
    _dynamic = False            # raw count_brick / count_step; one plan row, rewritten at every reset

    def __init__(self):
        self._init_common()
        self._setup(np.full((1, 30), 20.0))

    def create_plan(self):
        y, area, self.one_hot = _plans.random_sin_plan(self.plan_width, self.plan_height)
        return y, area

    def reset(self):
        self.one_hot = None
        y, area = self.create_plan()
        self._plan_dirty = False
        self._table[0] = y
        self._env.set_plan_row(0, y, update_tb=True)
        obs, (r, _) = self._do_reset(0)
        self.total_brick = area
        self.brick_memory = [[-1, -1]]
        self.position_memory = [r]
        return [obs, self.plan]

    def step(self, action, step_size):
        self._sync_plan()
        obs, reward, done, (r, _) = self._do_step(action, step_size)
        self._after_step(action, r, obs)
        return [obs, self.plan], reward, done


class deep_mobile_printing_1d1r_dynamic(_Env1D):
    """Env/1D/DMP_Env_1D_dynamic_usedata_plan.py :: deep_mobile_printing_1d1r(data_path, random_choose_paln=True)"""
    _dynamic = True

    def __init__(self, data_path, random_choose_paln=True):
        self._init_common()
        self.plan_dataset = list(_load_dataset(1, data_path))
        self.plan_dataset_len = len(self.plan_dataset)
        self.random_choose_paln = random_choose_paln
        self.index_for_non_random = 0
        self._setup(np.asarray(self.plan_dataset))

    def _lists(self, obs):
        norm = obs
        raw = norm.copy()
        raw[0, 5] = self.conut_brick
        raw[0, 6] = self.count_step
        return raw, norm

    def reset(self):
        idx = self._pick_plan()
        obs, (r, _) = self._do_reset(idx)
        self.total_brick = float(self.total_brick)
        self.plan_withborder = np.zeros((1, self.environment_width))
        self.plan_withborder[:, :self.HALF_WINDOW_SIZE] = -1
        self.plan_withborder[:, -self.HALF_WINDOW_SIZE:] = -1
        self.plan_withborder[:, self.HALF_WINDOW_SIZE:self.HALF_WINDOW_SIZE + self.plan_width] = self.plan
        self.brick_memory = [[-1, -1]]
        self.position_memory = [r]
        raw, norm = self._lists(obs)
        return [raw, norm, self.plan, r]                         # 4 elements on reset, 3 on step (:66-70 vs :92-96)

    def step(self, action):
        obs, reward, done, (r, _) = self._do_step(action)
        self._after_step(action, r, obs)
        raw, norm = self._lists(obs)
        return [raw, norm, self.plan], reward, done


# ================================================================================================ 2D / 3D
class _EnvGrid(_Facade):
    def _init_grid(self):
        self.step_size = 1
        self.plan_width = 20
        self.plan_height = 20
        self.count_brick = None
        self.brick_memory = None
        self.HALF_WINDOW_SIZE = 3
        self.environment_width = self.plan_width + 2 * self.HALF_WINDOW_SIZE
        self.environment_height = self.plan_height + 2 * self.HALF_WINDOW_SIZE
        self.position_memory = None
        self.observation = None
        self.count_step = 0
        self.plan = None
        self.input_plan = None
        self.total_brick = 0
        self.one_hot = None
        self.state_dim = (2 * self.HALF_WINDOW_SIZE + 1) ** 2 + 2

    def observation_(self, position):
        h = self.HALF_WINDOW_SIZE
        g = self.environment_memory
        return g[position[0] - h:position[0] + h + 1, position[1] - h:position[1] + h + 1].flatten().reshape(1, -1)

    def clip_position(self, position):
        lo, hi = self.HALF_WINDOW_SIZE, self.plan_width + self.HALF_WINDOW_SIZE - 1
        position[0] = min(max(position[0], lo), hi)
        position[1] = min(max(position[1], lo), hi)
        return position

    def _grid_reset(self, plan_idx):
        obs, (r, c) = self._do_reset(plan_idx)
        h = self.HALF_WINDOW_SIZE
        self.input_plan = self.plan[h:h + self.plan_height, h:h + self.plan_width]
        self.position_memory = [[r, c]]
        return obs, [r, c]

    def _grid_step(self, action, step_size=None):
        obs, reward, done, (r, c) = self._do_step(action, step_size)
        self.position_memory.append([r, c])
        return obs, reward, done, self.position_memory[-1]

    def iou(self):
        """3D: the class method of the reference; 2D: the caller-side boolean IoU (script/DQN/2d/DQN_2d_dynamic.py:63-71)."""
        return float(self._env.iou().item())

    def render(self, axe, *args, **kw):
        ax = axe
        ax.clear()
        h = self.HALF_WINDOW_SIZE
        g = self.environment_memory[h:h + self.plan_height, h:h + self.plan_width]
        p = self.plan[h:h + self.plan_height, h:h + self.plan_width]
        ax.set_xlim(0, self.plan_width)
        ax.set_ylim(0, self.plan_height)
        ax.title.set_text('step=%d, used_paint=%d, IOU=%.3f' % (self.count_step, self.count_brick, self.iou()))
        ax.imshow(np.where(g > 0, 2.0, 0.0) + np.where(p > 0, 1.0, 0.0), origin='lower', extent=(0, self.plan_width, 0, self.plan_height))
        ax.scatter(self.position_memory[-1][1] - h + 0.5, self.position_memory[-1][0] - h + 0.5, color='r')


class deep_mobile_printing_2d1r_static(_EnvGrid):
    """Env/2D/DMP_Env_2D_static.py :: deep_mobile_printing_2d1r(plan_choose=0)"""
    _dim, _dynamic = 2, False

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

    def create_plan(self):
        if self._err is not None:
            raise self._err
        plan = _plans.static_plan(2, self.plan_choose)
        return plan, plan.sum()

    def reset(self):
        self.create_plan()
        obs, _ = self._grid_reset(0)
        self.total_brick = float(self.total_brick)
        return obs

    def step(self, action, _step_size=None):
        obs, reward, done, _ = self._grid_step(action, _step_size)
        return obs, reward, done


class deep_mobile_printing_2d1r_hindsight(deep_mobile_printing_2d1r_static):
    """Env/2D/DMP_Env_2D_static_hindsight_replay.py :: step(action, step_size)"""

    def step(self, action, step_size):
        self._sync_plan()
        return deep_mobile_printing_2d1r_static.step(self, action, step_size)


class deep_mobile_printing_2d1r_dynamic(_EnvGrid):
    """Env/2D/DMP_Env_2D_dynamic_usedata_plan.py :: deep_mobile_printing_2d1r(data_path, random_choose_paln=True)"""
    _dim, _dynamic = 2, True

    def __init__(self, data_path, random_choose_paln=True):
        self._init_grid()
        self.total_step = 600
        self.action_dim = 5
        self.plan_dataset = list(_load_dataset(2, data_path))
        self.plan_dataset_len = len(self.plan_dataset)
        self.random_choose_paln = random_choose_paln
        self.index_for_non_random = 0
        self._setup(np.asarray(self.plan_dataset))

    def reset(self):
        obs, pos = self._grid_reset(self._pick_plan())
        self.total_brick = float(self.total_brick)
        return [obs, self.input_plan, pos]

    def step(self, action):
        obs, reward, done, pos = self._grid_step(action)
        return [obs, self.input_plan, pos], reward, done


class _Env3D(_EnvGrid):
    _dim = 3

    def _init_3d(self):
        self._init_grid()
        self.plan_length = 10
        self.z = 6
        self.environment_length = self.plan_length + 2 * self.HALF_WINDOW_SIZE
        self.start = None
        self.check = []
        self.action_dim = 8
        self.blank_size = 2

    def check_sur(self, position):
        """Env/3D/DMP_simulator_3d_static_circle.py:88-102 on the current height map (host copy)."""
        g = self.environment_memory
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

    def reward_check(self, position):
        g = self.environment_memory
        if g[position[0], position[1]] > self.plan[position[0], position[1]]:
            return -1.0
        if g[position[0], position[1]] == self.plan[position[0], position[1]]:
            return 10.0
        return 1.0


class deep_mobile_printing_3d1r_static(_Env3D):
    """Env/3D/DMP_simulator_3d_static_circle.py :: deep_mobile_printing_3d1r(plan_choose=1)"""
    _dynamic = False

    def __init__(self, plan_choose=1):
        self._init_3d()
        self.total_step = 1300
        self.plan_choose = plan_choose
        self._err = None
        if plan_choose not in (0, 1):
            self._err = ValueError('0: Dense circle, 1: Sparse circle')
            return
        self._setup(_plans.static_plan(3, plan_choose)[None])

    def create_plan(self):
        if self._err is not None:
            raise self._err
        plan = _plans.static_plan(3, self.plan_choose)
        return plan, plan.sum()

    def reset(self):
        self.create_plan()
        self.check = []
        self.step_size = 1
        obs, _ = self._grid_reset(0)
        return obs

    def step(self, action, _step_size=None):
        obs, reward, done, _ = self._grid_step(action, _step_size)
        return obs, reward, done


class deep_mobile_printing_3d1r_hindsight(deep_mobile_printing_3d1r_static):
    """Env/3D/DMP_simulator_3d_static_circle_hindsight_replay.py :: step(action, step_size)"""

    def step(self, action, step_size):
        self._sync_plan()
        return deep_mobile_printing_3d1r_static.step(self, action, step_size)


class deep_mobile_printing_3d1r_dynamic(_Env3D):
    """Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py :: deep_mobile_printing_3d1r(data_path, random_choose_paln=True)"""
    _dynamic = True

    def __init__(self, data_path, random_choose_paln=True):
        self._init_3d()
        self.total_step = 1000
        self.plan_dataset = list(_load_dataset(3, data_path))
        self.plan_dataset_len = len(self.plan_dataset)
        self.random_choose_paln = random_choose_paln
        self.index_for_non_random = 0
        self._setup(np.asarray(self.plan_dataset))

    def reset(self):
        idx = self._pick_plan()
        self.check = []
        self.step_size = 1
        obs, pos = self._grid_reset(idx)
        return [obs, self.input_plan, pos]

    def step(self, action):
        obs, reward, done, pos = self._grid_step(action)
        return [obs, self.input_plan, pos], reward, done


class deep_mobile_printing_3d1r_hindsight_dynamic(deep_mobile_printing_3d1r_dynamic):
    """Env/3D/DMP_simulator_3d_dynamic_triangle_hindsight_replay.py :: deep_mobile_printing_3d1r_hindsight(data_path,
    random_choose_paln=True) -- the dataset class with step(action, step_size); reset() returns [obs with the RAW counters,
    input_plan] (:71-73) while step() keeps the canonical normalised 3-list (:199-228)"""

    def reset(self):
        obs, plan, _ = deep_mobile_printing_3d1r_dynamic.reset(self)
        return [obs, plan]                                       # both counters are 0 at reset: raw == normalised

    def step(self, action, step_size):
        self._sync_plan()
        obs, reward, done, pos = self._grid_step(action, step_size)
        return [obs, self.input_plan, pos], reward, done


class deep_mobile_printing_2d1r_hindsight_dynamic(deep_mobile_printing_2d1r_dynamic):
    """Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py :: deep_mobile_printing_2d1r_hindsight(data_path,
    random_choose_paln=True) -- the dataset class with raw counters in every observation and step(action, step_size).

    reset() first draws a throw-away random triangle (create_plan, :37-59, :61-64) and then takes the dataset plan.  The
    triangle's vertices come from np.random exactly as in the reference (two randint(0, 20, size=3) per attempt, redrawn until
    the area exceeds 50 dense / 20 sparse); the rasterisation runs on the device (snac_make_plans with explicit vertices),
    whose rules reproduce every cv2-drawn plan of the reference's datasets bit for bit -- so a seeded script sees the
    reference's np.random stream.  (cv2 itself is absent where the goldens are recorded, hence no recorded trajectory of
    this class: the dynamics are the pinned 2D dataset dynamics with raw counters and caller-supplied step sizes.)"""

    _layout = dict(obs_scalars="raw")

    def __init__(self, data_path, random_choose_paln=True):
        deep_mobile_printing_2d1r_dynamic.__init__(self, data_path, random_choose_paln)
        self.plan_choose = 1 if "sparse" in data_path else 0     # :31-32
        self._gen = None

    def create_plan(self):
        if self.plan_choose not in (0, 1):
            raise ValueError(' 0: Dense triangle, 1: Sparse triangle')
        from .batched import BatchedDMPEnv

        if self._gen is None:                                    # a one-row scratch table for the rasteriser
            self._gen = BatchedDMPEnv(2, False, 1, plans=np.zeros((1, 26, 26)))
        area = [50, 20]
        total_area = 0
        while total_area <= area[self.plan_choose]:
            x = np.random.randint(0, self.plan_width, size=3)
            y = np.random.randint(0, self.plan_height, size=3)
            v = np.array([[x[0], y[0], x[1], y[1], x[2], y[2]]], np.int8)
            total_area = int(self._gen.generate_plans(0, 1, sparse=bool(self.plan_choose), vertices=v).item())
        self._gen._sync_plans_full()
        return self._gen.plans_full[0].copy(), float(total_area)

    def _raw(self, obs):
        return obs                                               # obs_scalars "raw": the kernel writes the counters

    def reset(self):
        self.plan, self.total_brick = self.create_plan()         # the reference's throw-away draw (:61)
        obs, plan, pos = deep_mobile_printing_2d1r_dynamic.reset(self)
        return [self._raw(obs), plan, pos]

    def step(self, action, step_size):
        self._sync_plan()
        obs, reward, done, pos = self._grid_step(action, step_size)
        return [self._raw(obs), self.input_plan, pos], reward, done
