"""2D tree-search node pools with ONE record per node (snac_node2d, include/snac_hip.h; snac_amd/csrc/k_nodes2d.hip).

`BatchedDMPEnv.transition()` uses the batch itself as the node pool: a tree edge then reads its parent's header, episode counter and
board from three arrays at a random row -- three lines of memory for 100 bytes.  A NodePool2D keeps the three in one 128-byte record,
so an edge reads one line and writes one (524 288 random-parent edges: see bench.py `transition_2d_nodes_524288_edges`).  The rules
are the batch's: Env/2D/DMP_ENV_2D_dynamic_MCTS.py:117-175 `transition(state, action)`, one call per tree edge in
script/MCTS/utils/mcts_Qvalue_dynamic.py:88,118 -- here m edges per launch.

    env = BatchedDMPEnv(2, True, 4096, seed=1); env.reset()            # the roots (or import_states(...))
    pool = NodePool2D(env, 1 << 20)
    pool.load(rows=root_rows, node_rows=root_nodes)                     # batch rows -> node records
    obs, reward, done = pool.transition(actions, step_size, src=parents, dst=children)
    pool.store(node_rows=leaves, rows=leaf_rows)                        # node records -> batch rows (evaluate(), observe(), ...)
"""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class NodePool2D:
    def __init__(self, env, rows):
        """env: a 2D BatchedDMPEnv -- its rules, plan table, observation dtype and device are the pool's; rows: node records."""
        if env.kind != 2:
            raise ValueError("node records exist for the 2D kinds")
        if env.obs_dim != 51:
            raise ValueError("a node pool writes the canonical observation rows")
        self.env, self.rows = env, int(rows)
        if self.rows < 1:
            raise ValueError("rows must be >= 1")
        # torch allocations are at least 512-byte aligned: one record = one 128-byte line
        self.records = torch.zeros((self.rows, 32), dtype=torch.int32, device=env.device)
        assert self.records.data_ptr() % 128 == 0
        self._lib = env._lib

    # ---- records <-> batch rows ---------------------------------------------------------------------------------
    def _idx(self, x, m, limit, what):
        if x is None:
            if m > limit:
                raise ValueError("%s: %d rows, %d given" % (what, limit, m))
            return None
        t = torch.as_tensor(x, device=self.env.device).reshape(-1)
        if int(t.numel()) != m:
            raise ValueError("%s must have %d entries" % (what, m))
        if m and (int(t.min()) < 0 or int(t.max()) >= limit):
            raise ValueError("%s out of range" % what)
        return t.to(torch.int32).contiguous()

    def _count(self, a, b, default):
        for x in (a, b):
            if x is not None:
                return int(torch.as_tensor(x).numel())
        return int(default)

    def load(self, rows=None, node_rows=None, env=None):
        """Node record node_rows[i] <- batch row rows[i] of `env` (default: the pool's own batch); None = row i."""
        env = env or self.env
        m = self._count(rows, node_rows, min(env.num_envs, self.rows))
        ri, ni = self._idx(rows, m, env.num_envs, "rows"), self._idx(node_rows, m, self.rows, "node_rows")
        with torch.cuda.device(env.device):
            _lib.check(self._lib.snac_nodes2d_pack(C.byref(env._desc), C.byref(env._state), _ptr(ri), m, _ptr(self.records), self.rows, _ptr(ni),
                                                   env._stream()))
        return m

    def store(self, node_rows=None, rows=None, env=None):
        """Batch row rows[i] of `env` <- node record node_rows[i]; None = row i."""
        env = env or self.env
        m = self._count(rows, node_rows, min(env.num_envs, self.rows))
        ri, ni = self._idx(rows, m, env.num_envs, "rows"), self._idx(node_rows, m, self.rows, "node_rows")
        with torch.cuda.device(env.device):
            _lib.check(self._lib.snac_nodes2d_unpack(C.byref(env._desc), _ptr(self.records), self.rows, _ptr(ni), m, C.byref(env._state), _ptr(ri),
                                                     env._stream()))
        env._was_reset = True
        return m

    # ---- the edges ------------------------------------------------------------------------------------------------
    def transition(self, actions, step_size=None, src=None, dst=None, t=0, want_obs=True, check=True):
        """m tree edges in one launch: record dst[i] <- step(record src[i], actions[i], step_size[i]); src / dst None = record i.
        Same rules and outputs as BatchedDMPEnv.transition(): no auto-reset, a dst record must not be the src record of another
        edge of the same call (check=False skips that test: a host round trip per wave).  Returns (obs [m, 51], reward [m], done [m])."""
        env = self.env
        a = torch.as_tensor(actions, device=env.device) if not torch.is_tensor(actions) else actions.to(env.device)
        m = int(a.numel())
        a = env._i8(a.reshape(-1), (m,), "actions")
        k = env._i8(step_size, (m,), "step_size")
        si, di = self._idx(src, m, self.rows, "src"), self._idx(dst, m, self.rows, "dst")
        if check and m and (si is not None or di is not None):
            s_ = si if si is not None else torch.arange(m, device=env.device, dtype=torch.int32)
            d_ = di if di is not None else torch.arange(m, device=env.device, dtype=torch.int32)
            if bool(torch.isin(d_, s_[s_ != d_]).any()) or int(torch.unique(d_).numel()) != m:
                raise ValueError("dst records must be distinct and must not be the src record of another edge")
        obs = torch.empty((m, env.obs_dim), dtype=env.obs_dtype, device=env.device) if want_obs else None
        reward = torch.empty((m,), dtype=torch.float32, device=env.device)
        done = torch.empty((m,), dtype=torch.uint8, device=env.device)
        with torch.cuda.device(env.device):
            _lib.check(self._lib.snac_transition_nodes2d(C.byref(env._desc), C.byref(env._state), _ptr(self.records), self.rows, m, _ptr(si), _ptr(di),
                                                         int(t) & 0xFFFFFFFF, _ptr(a), _ptr(k), _ptr(obs), _ptr(reward), _ptr(done), env._stream()))
        return obs, reward, done.view(torch.bool)

    # ---- what a search reads of its nodes (decoded from the records: snac_env_hdr) -----------------------------------
    def _hdr16(self):
        return self.records[:, :4].contiguous().view(torch.int16).view(self.rows, 8)

    @property
    def position(self):
        h = self.records[:, 0]
        r = ((h & 0xFF) << 24) >> 24
        c = (((h >> 8) & 0xFF) << 24) >> 24
        return torch.stack([r, c], dim=1)

    @property
    def need_reset(self):
        return ((self.records[:, 0] >> 16) & _lib.FLAG_NEED_RESET) != 0

    @property
    def count_brick(self):
        return self._hdr16()[:, 2].to(torch.int32)

    @property
    def count_step(self):
        return self._hdr16()[:, 3].to(torch.int32)

    @property
    def total_brick(self):
        return self._hdr16()[:, 4].to(torch.int32)

    @property
    def plan_idx(self):
        return self._hdr16()[:, 5].to(torch.int32)

    @property
    def boards(self):
        """[rows, 20] row words of the bit boards (bit j of word q = interior cell (q, j))."""
        return self.records[:, 8:28]
