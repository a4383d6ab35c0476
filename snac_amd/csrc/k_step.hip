// k_step.hip -- k_step2d / k_step3d: snac_step on identity rows
#include "snac_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// snac_step on the identity rows, round 3.  k_transition2d / k_transition3d spend their time in the texture addresser, not in
// HBM (profiles/r02j_step_summary.txt: 43 % / 26 % of the wave cycles stalled at ISSUE, half of the HBM rate): they issue one
// narrow memory instruction per edge -- a 408-byte row store, in 3D also a 49-lane gather of 2-byte cells -- and seven scattered
// 4-byte / 2-byte loads per lane.  Here a wave takes a tile of 64 consecutive envs and every memory instruction is wide:
//   2D  the tile's 64 records (5 120 contiguous bytes) arrive as five 16-byte-per-lane loads and lie in LDS; lane l steps env l
//       on its row word, builds the 7 window rows as two-bit codes (k_transition2d's encoding) and hands them to emit_tile;
//   3D  lane l loads the 7 window rows of ITS env as seven 16-byte loads (8 cells from a column clamped into the record, 2-byte
//       aligned: the hardware takes unaligned global accesses) into a scratch row in LDS with -1 on either side, so that frame
//       cells, the neighbour / path cells of K3D::step and the window cells are all ds_read_i16 at (row, 4 + column - first
//       column); the built cell is patched there; an env that moved reloads its rows round the new position (mostly L2 hits);
//   both    the 51 values of an env leave through emit_tile (LDS transposition, 1 KiB stores); the staging tile reuses the
//       record / scratch LDS, whose values are in registers by then.  Episodic sums by no-return atomics (nothing waits for them).
// Write-back: the header, the episode counter of an env that was reset, the ONE changed row word / cell (a reset env: its record).
// Identity rows only (snac_step, snac_step_scalar), N % 4 == 0 and a 16-byte aligned obs; the canonical layout, in 2D also the layout
// variants of large batches (k_step2d<.., VAR>: from 45 056 / 32 769 / 24 576 envs, half-filled tiles for 24 577 .. 32 768; k_step3d<.., VAR>:
// from 24 576); everything else -- tree edges with gathered rows, the other layout variants, N = 1 of
// the single-env classes -- stays on k_transition2d / 3d / k_transition.

// VAR: the layout variants of snac_env_desc (rows of a.ld values: the 451-value rows of the PPO copies are what a trainer that steps
// tens of thousands of envs per tick reads): the rows leave through emit_rows_var (k_rollout2d's row assembly), the plan tail from the
// lanes' plan rows in LDS.  25 KB of LDS per wave, one block of four waves per CU -- 65 536 envs are exactly one round.
// TE = 32: half-filled tiles (lanes 32 .. 63 idle) -- twice the waves for batches that do not fill the CUs with 64 rows of kilobytes per wave.
// NTL / NTS (canonical rows only): the records by non-temporal loads / the rows by non-temporal stores -- which pays depends on whether state
// and rows fit the Infinity Cache: the launch picks one of three forms by batch size (profiles/r06_step_loads.txt)
// AUX: not a step -- snac_reset with a mask (the masked envs start over, every env reports its observation) and snac_observe on the same loads
// and rows: no action, no rules, no reward / done; a header is written only for an env that was reset (k_aux: 80 us per masked reset of 524 288 envs).
template <bool DYN, typename OT, int WPB, bool VAR = false, int TE = 64, bool NTL = true, bool NTS = false, bool AUX = false>
__global__ __launch_bounds__(WPB * 64) void k_step2d(const KArgs a) {
    using K = K2D<DYN, 64>;
    constexpr int E = TE, GE = K::GE;
    static_assert(E * GE * 4 <= TILE_STG_BYTES, "the records fit the staging tile");
    constexpr int PLW = VAR ? GE * 65 : 0, CMPW = VAR ? 64 * VAR_CMP_WORDS : 0;           // the envs' plan rows [row][65], emit_rows_var's records
    constexpr int WAVE_WORDS = (VAR ? VAR_STG_BYTES : TILE_STG_BYTES) / 4 + PLW + CMPW;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB * WAVE_WORDS];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    uint32_t* const rec = lds_all + wv * WAVE_WORDS;                 // [64][20] row words, then the staging tile
    // ---- every load that does not depend on another: the tile's records (16 bytes per lane), header, episode counter
    uint4 rv[5];
    {
        const uint4* const g4 = (const uint4*)a.grid + (size_t)env0 * 5;
        // nontemporal, like k_step3d's window rows: read once per tick, and kept out of the way of the row stores' lines in L2
        // (46.1 against 48.4 us per tick at N = 524 288, three runs each; the tree-edge kernels, whose parents are shared by
        // their children, lose 15-25 % with it and keep plain loads)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int g = i * 64 + lane;
            rv[i] = make_uint4(0u, 0u, 0u, 0u);
            if (g < nenv * 5) { const u32x4 t = load_nt_if<NTL>((const u32x4*)(g4 + g)); rv[i] = make_uint4(t.x, t.y, t.z, t.w); }   // (NTL: outside SNAC_STEP2D_PLAIN_LO .. _HI envs)
        }
    }
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;
    int episode = 0;
    if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    int act = 0, k = 1;
    bool nr;
    if constexpr (AUX) {
        nr = active && a.aux_op == AUX_RESET && (a.mask ? a.mask[env] != 0 : true);
        if (nr) {                                                    // k_aux's reset: the plan row from the indices, the scalar or the counter RNG
            episode += 1;
            int pidx;
            if (a.plan_idx_in) pidx = a.plan_idx_in[env];
            else if (a.plan_scalar >= 0) pidx = a.plan_scalar;
            else pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, a.static_plan);
            K::reset(a, s, min(max(pidx, 0), a.num_plans - 1));
        }
    } else {
        const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
        act = (int)(((w >> 16) * (uint32_t)K::A) >> 16); k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
        if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
        if (a.actions && active) act = (int)a.actions[env];
        if (a.step_size && active) k = (int)a.step_size[env];
        k = min(max(k, 1), 3);
        nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
        if (nr) {
            const int old_pidx = s.pidx, old_tb = s.tb;
            episode += 1;
            const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
            K::reset(a, s, pidx == old_pidx ? -1 : pidx);
            if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
        }
    }
    const uint32_t* const prow = (const uint32_t*)a.plans + (size_t)s.pidx * GE;
    const int q0 = min(max(s.r - 3, 0), GE - 1), bit = min(max(s.c - 3, 0), 19);
    const uint32_t pword = AUX ? 0u : prow[q0];                      // the one dependent load: the plan row under the agent (L2)
#pragma unroll
    for (int i = 0; i < 5; ++i) ((uint4*)rec)[i * 64 + lane] = rv[i];
    uint32_t* const mine = rec + lane * GE;
    if (nr) {                                                        // a freshly reset board is empty
#pragma unroll
        for (int q = 0; q < GE; ++q) mine[q] = 0u;
    }
    // ---- K2D::step (DMP_Env_2D_dynamic_usedata_plan.py:85-147) on the agent's row word
    const uint32_t row0 = mine[q0];
    const bool was = ((row0 >> bit) & 1u) != 0u, planned = ((pword >> bit) & 1u) != 0u;
    const uint32_t newrow = row0 | (1u << bit);
    Rule2D u;
    u.drop = false; u.term = false; u.done = (s.flags & SNAC_FLAG_NEED_RESET) != 0; u.reward = 0;   // AUX: SNAC_TAIL_RECORD outside a step reports the pending-reset flag
    if constexpr (!AUX) u = rules2d(s, act, k, was, planned, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
    const bool drop = active && u.drop;
    if (drop) mine[q0] = newrow;
    const bool done = active && u.done;
    const int reward = u.reward;
    if constexpr (!AUX) {
        s.ep_ret = clamp16(s.ep_ret + reward);
        s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    }
    if (active) {
        if constexpr (!AUX) {
            if (a.reward) a.reward[env] = (float)reward;
            if (a.done) a.done[env] = done ? 1 : 0;
        }
        if (!AUX || nr) a.hdr[env] = s.pack();
        if (nr) {
            a.episode[env] = episode;
            uint32_t* const gw = (uint32_t*)a.grid + (size_t)env * GE;
#pragma unroll
            for (int q = 0; q < GE; ++q) gw[q] = mine[q];
        } else if (drop) {
            ((uint32_t*)a.grid)[(size_t)env * GE + q0] = newrow;
        }
    }
    if (!AUX && a.stats_on && __builtin_expect(__any(done), 0)) {    // snac_step: episodic sums; the boolean IoU needs board and plan
        if (done) {
            int inter = 0, uni = 0;
            for (int q = 0; q < GE; ++q) { const uint32_t g = mine[q], p = prow[q]; inter += __popc(g & p); uni += __popc(g | p); }
            const double v = (double)inter / (double)uni;
            stat_add(a.stat_episodes + env, 1);
            stat_add(a.stat_return + env, s.ep_ret);
            stat_add(a.stat_iou_fx + env, __double2ll_rn(v * FX40));
        }
    }
    if (!a.obs) return;
    // ---- the 7x7 window round the new position as two-bit codes (00 empty / 01 brick / 11 frame), 14 bits per row
    uint32_t wr[7];
    {
        const int sh = s.c - 3;                                      // first window column, bordered: 0..19
        constexpr uint32_t FRAME26 = 0x3800007u;                     // frame columns 0-2 and 23-25 of an interior row
        const uint32_t frm = spread16((FRAME26 >> sh) & 0x7Fu) * 3u;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = s.r - 6 + i;                               // board row of window row i
            const bool in = (unsigned)q < (unsigned)GE;
            const uint32_t g = mine[in ? q : 0];
            wr[i] = in ? (spread16(((g << 3) >> sh) & 0x7Fu) | frm) : 0x3FFFu;
        }
    }
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const bool norm = VAR ? (a.sc_norm != 0) : DYN;
    const double v0 = norm ? c0 / (double)s.tb : c0, v1 = norm ? c1 / (double)a.total_step : c1;
    if constexpr (VAR) {
        uint32_t* const pl = rec + VAR_STG_BYTES / 4;                // [20][65]: lane l's column holds its env's plan rows
        uint32_t* const cmp = pl + PLW;
        if (a.tail & SNAC_TAIL_PLAN) {
#pragma unroll
            for (int q = 0; q < GE; ++q) pl[q * 65 + lane] = prow[q];
        }
        const int recv[8] = {reward, done ? 1 : 0, s.r, s.c, s.cb, s.cs, s.tb, s.pidx};   // SNAC_TAIL_RECORD's values (record_value)
        emit_rows_var<OT>((char*)rec, cmp, (char*)a.obs + (size_t)env0 * (size_t)a.ld * sizeof(OT), lane, nenv, a.ld, a.tail, a.frame_val, wr, v0, v1,
                          recv, [&](int e, int row) { return pl[row * 65 + e]; });
    } else {
        emit_tile<OT, NTS>((char*)rec, (char*)a.obs + (size_t)env0 * K::D * sizeof(OT), lane, nenv,
                           [&](int el) { const int i = el / 7, j = el - 7 * i; return ((int)(wr[i] << (30 - 2 * j))) >> 30; }, v0, v1);
    }
}

typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));   // a 16-byte global access at a 2-byte aligned address

// VAR: the layout variants (rows of a.ld values: 451 with the plan tail of the PPO copies): the heads (window, scalar slots, position,
// record) leave in groups of 16 envs through the staging tile, lane = value; the plan tail of an env is its plan row itself -- 50
// lanes load it 16 bytes each, convert their 8 cells, and the 3200 (1600) bytes are turned through the staging tile into 16-byte
// pieces in row order; eight envs' loads are issued before the first of their stores (a load behind stores waits for them).
template <bool DYN, typename OT, int WPB, bool VAR = false>
__global__ __launch_bounds__(WPB * 64) void k_step3d(const KArgs a) {
    using K = K3D<DYN, 8>;
    constexpr int E = 64, GE = K::GE;
    // scratch per lane: 8 bytes of -1, 7 rows of 12 cells [2 x -1][8 loaded cells][2 x -1], 8 bytes of -1.  A window column may lie up
    // to 3 cells left or 4 right of the loaded block: what a row lacks in pads, its neighbour's pads (or the lane's own leading /
    // trailing 8 bytes) supply.  184 bytes per lane: 46 dwords, a 2-way bank pattern; the whole scratch is smaller than the staging tile.
    constexpr int LS = 184, RB = 24, R0 = 8;
    constexpr int WAVE_BYTES = E * LS > TILE_STG_BYTES ? E * LS : TILE_STG_BYTES;
    static_assert(WAVE_BYTES % 16 == 0, "16-byte aligned staging tiles");
    __shared__ __attribute__((aligned(16))) char lds_all[WPB * WAVE_BYTES];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    char* const scr = lds_all + wv * WAVE_BYTES;
    char* const mine = scr + lane * LS;
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;
    int episode = 0;
    if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
    *(uint64_t*)mine = ~0ull;                                        // the pads
    *(uint64_t*)(mine + R0 + 7 * RB) = ~0ull;
#pragma unroll
    for (int i = 0; i < 7; ++i) { *(uint32_t*)(mine + R0 + i * RB) = ~0u; *(uint32_t*)(mine + R0 + i * RB + 20) = ~0u; }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions && active) act = (int)a.actions[env];
    if (a.step_size && active) k = (int)a.step_size[env];
    k = min(max(k, 1), 3);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const int16_t* const src = (const int16_t*)a.grid + (size_t)env * GE;
    // the 7 window rows round (r, c) -> scratch; returns the cell index of window column 0 in a scratch row.  Interior column
    // of window column j: c - 6 + j; 8 cells are loaded from `start` (clamped so that they lie inside the row), to cells 2..9.
    auto load_window = [&](int r, int c) -> int {
        const int start = min(max(c - 6, 0), 12);
        uint4 v[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = r - 6 + i;                                 // interior row of window row i
            const bool in = (unsigned)q < 20u;
            v[i] = make_uint4(~0u, ~0u, ~0u, ~0u);                   // a frame row
            if (in) {
                v[i] = make_uint4(0u, 0u, 0u, 0u);                   // a freshly reset env is empty
                if (active && !nr) {
                    // nontemporal: the rows are streamed once per tick (100.2 against 103.8 us per tick at N = 524 288, three runs each)
                    const u32x4_a2 t = __builtin_nontemporal_load((const u32x4_a2*)(src + q * 20 + start));
                    v[i] = make_uint4(t.x, t.y, t.z, t.w);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            uint32_t* const d = (uint32_t*)(mine + R0 + i * RB + 4);
            d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
        return 2 + (c - 6) - start;
    };
    const int d = act & 3;
    const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
    const int tr = s.r + dr - 3, tc = s.c + dc - 3;                  // the build target in plan coordinates
    const bool inside = (unsigned)tr < 20u && (unsigned)tc < 20u;
    const int tcell = inside ? tr * 20 + tc : 0;
    const int pl = ((const int16_t*)a.plans)[(size_t)s.pidx * GE + tcell];
    const int h0 = load_window(s.r, s.c);
    constexpr int RC = RB / 2;                                       // cells per scratch row
    const int16_t* const cen = (const int16_t*)(mine + R0) + 3 * RC + h0 + 3;   // the agent's cell
    // ---- K3D::step by selects (the formulation of k_transition3d / Roll3D::tick)
    const int n0 = cen[-1], n1 = cen[1], n2 = cen[RC], n3 = cen[-RC];    // check_sur: left, right, "up" (row + 1), "down"
    const int dl = dr * RC + dc;
    const int c2 = cen[2 * dl], c3 = cen[3 * dl];
    const int old_r = s.r, old_c = s.c;
    const Rule3D u = rules3d<DYN>(s, act, k, n0, n1, n2, n3, c2, c3, active, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
    const bool built = u.built;
    const int newh = u.newh;
    s.cross += (built && newh <= pl) ? 1 : 0;
    bool done = u.done;
    const int reward = u.sel ? reward_check3d(newh, pl) : u.reward0;
    done = done && active;
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (built) ((int16_t*)(mine + R0))[(3 + dr) * RC + h0 + 3 + dc] = (int16_t)newh;   // the window shows the built cell
    if (active) {
        if (a.reward) a.reward[env] = (float)reward;
        if (a.done) a.done[env] = done ? 1 : 0;
        a.hdr[env] = s.pack();
        if (nr) a.episode[env] = episode;
        if (built && !nr) ((int16_t*)a.grid)[(size_t)env * GE + tcell] = (int16_t)newh;
        if (a.stats_on && done) {                                    // snac_step: episodic sums
            const double v = K::iou(nullptr, s, 0);
            stat_add(a.stat_episodes + env, 1);
            stat_add(a.stat_return + env, s.ep_ret);
            stat_add(a.stat_iou_fx + env, __double2ll_rn(v * FX40));
        }
    }
    for (unsigned long long mk = __ballot(nr); mk; mk &= mk - 1) {   // a reset env's record: empty, but for the cell it built
        const int e = __ffsll(mk) - 1;
        const int tp = __builtin_amdgcn_readlane(built ? tcell : -1, e), nh = __builtin_amdgcn_readlane(newh, e);
        if (lane < 50) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (tp >= 0 && (tp >> 3) == lane) {
                const int hw = tp & 7;
                const uint32_t put = ((uint32_t)nh & 0xFFFFu) << ((hw & 1) * 16);
                if ((hw >> 1) == 0) v.x = put; else if ((hw >> 1) == 1) v.y = put; else if ((hw >> 1) == 2) v.z = put; else v.w = put;
            }
            ((uint4*)a.grid)[(size_t)(env0 + e) * 50 + lane] = v;
        }
    }
    if (!a.obs) return;
    // ---- the window round the NEW position: an env that moved reloads its rows (the neighbours' lines are in L2 by now)
    int h1 = h0;
    if (s.r != old_r || s.c != old_c) h1 = load_window(s.r, s.c);
    int cellv[K::W];
    {
        const int16_t* const wp = (const int16_t*)(mine + R0) + h1;
#pragma unroll
        for (int el = 0; el < K::W; ++el) { const int i = el / 7, j = el - 7 * i; cellv[el] = wp[i * RC + j]; }
    }
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const bool norm = VAR ? (a.sc_norm != 0) : DYN;
    const double v0 = norm ? c0 / (double)s.tb : c0, v1 = norm ? c1 / (double)a.total_step : c1;
    if constexpr (!VAR) {
        emit_tile<OT, ROWS_NT_STEP>(scr, (char*)a.obs + (size_t)env0 * K::D * sizeof(OT), lane, nenv, [&](int el) { return cellv[el]; }, v0, v1);
    } else {
        typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte global store at a 4-byte aligned address
        const int LD = a.ld;
        const int pos_n = (a.tail & SNAC_TAIL_POSITION) ? 2 : 0, plan_n = (a.tail & SNAC_TAIL_PLAN) ? 400 : 0, rec_n = (a.tail & SNAC_TAIL_RECORD) ? 8 : 0;
        const int NE = K::D + pos_n + rec_n;                         // values of a row beside the plan tail (<= 61)
        OT* const orow = (OT*)a.obs + (size_t)env0 * LD;
        // ---- heads: 16 envs at a time, each lane of the group files its NE values, then one env per store, lane = value
        const int rv[8] = {reward, done ? 1 : 0, s.r, s.c, s.cb, s.cs, s.tb, s.pidx};   // SNAC_TAIL_RECORD's values (record_value)
        const int dst = lane < K::D + pos_n ? lane : lane + plan_n;  // the record lies behind the plan tail
        for (int g0 = 0; g0 < nenv; g0 += 16) {
            if (lane >= g0 && lane < g0 + 16) {
                OT* const S = (OT*)scr + (lane - g0) * NE;
#pragma unroll
                for (int el = 0; el < K::W; ++el) S[el] = (OT)(double)cellv[el];
                S[K::W] = (OT)v0; S[K::W + 1] = (OT)v1;
                OT* q = S + K::D;
                if (pos_n) { q[0] = (OT)(double)rv[2]; q[1] = (OT)(double)rv[3]; q += 2; }
                if (rec_n) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) q[j] = (OT)(double)rv[j];
                }
            }
            const int ge = min(16, nenv - g0);
            OT hv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = ((const OT*)scr)[r * NE + min(lane, NE - 1)];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r < ge && lane < NE) orow[(size_t)(g0 + r) * LD + dst] = hv[r];
        }
        // ---- plan tails: eight envs' rows loaded, then each turned through the staging tile into pieces in row order
        if (plan_n) {
            constexpr int CP = 16 / (int)sizeof(OT);                 // cells per 16-byte piece of the output: 2 / 4
            constexpr int NPC = 400 / CP;                            // pieces per tail: 200 / 100
            for (int b0 = 0; b0 < nenv; b0 += 8) {
                uint4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int pe = __builtin_amdgcn_readlane(s.pidx, min(b0 + u, nenv - 1));
                    t[u] = ((const uint4*)((const int16_t*)a.plans + (size_t)pe * 400))[min(lane, 49)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (b0 + u < nenv) {                             // wave-uniform
                        if (lane < 50) {
                            const uint32_t w4[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
                            OT* const S = (OT*)scr + lane * 8;
#pragma unroll
                            for (int c = 0; c < 8; ++c) S[c] = (OT)(double)(int)(int16_t)(w4[c >> 1] >> ((c & 1) * 16));
                        }
                        char* const gq = (char*)(orow + (size_t)(b0 + u) * LD + K::D + pos_n);
                        uint4 pv[(NPC + 63) / 64];
#pragma unroll
                        for (int k = 0; k < (NPC + 63) / 64; ++k) pv[k] = ((const uint4*)scr)[min(lane + 64 * k, NPC - 1)];
#pragma unroll
                        for (int k = 0; k < (NPC + 63) / 64; ++k)
                            if (lane + 64 * k < NPC) {
                                u32x4_a4 o; o.x = pv[k].x; o.y = pv[k].y; o.z = pv[k].z; o.w = pv[k].w;
                                *(u32x4_a4*)(gq + (size_t)(lane + 64 * k) * 16) = o;
                            }
                    }
                }
            }
        }
    }
}



// ------------------------------------------------------------------------------------------------
// k_step3ds (round 5): the canonical 3D step() with COOPERATIVE SPAN LOADS.  k_step3d reads the window as seven 16-byte loads per LANE,
// every lane in another line (448 scattered requests per wave), and again for every env that moved: what the step pays for is the number
// of those requests, not their bytes (profiles/r05_step_experiments.txt parts 1-3 and 5: more resident waves do not help, a byte plane buys
// 9 %, coalesced span loads 37 % in a timing build).  Here the rows a tick can need -- the 7 rows of the window plus the 3 a row move can
// reach, TEN whole rows = 400 contiguous bytes of the env's record -- are read by 26 neighbouring lanes, 16 bytes each (piece P = it * 64 +
// lane of the wave's 64 x 26 pieces belongs to env P / 26): coalesced runs, no reload after a move, the action is known before the load
// and picks the side the extra rows lie on; pieces none of whose cells the tick can touch (rows AND columns) are not fetched.  The pieces wait in registers; the wave's LDS (the staging tile: 13 312 bytes = 32 spans of
// 416 bytes exactly) takes them HALF a wave at a time -- iterations 0 .. 12 are envs 0 .. 31, 13 .. 25 envs 32 .. 63 --, and the lanes of
// that half do their whole tick on it: the six cells of the transition, K3D::step by selects (k_step3d's formulation), the state stores,
// the 49 window cells round the NEW position (rows and columns outside the map are the frame by their coordinates: no pads, no
// bounds to trust).  The rows then leave through emit_tile as in k_step3d.  Canonical layout, identity rows, N % 4 = 0, aligned obs.
template <bool DYN, typename OT, int WPB>
__global__ __launch_bounds__(WPB * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_step3ds(const KArgs a) {   // three waves per SIMD: what the LDS allows (<= 168 VGPRs)
    using K = K3D<DYN, 8>;
    constexpr int E = 64, GE = K::GE, NPC = 26, SPAN = NPC * 16;     // pieces and bytes per env
    static_assert(32 * SPAN == TILE_STG_BYTES && (E * NPC) % 64 == 0 && (32 * NPC) % 64 == 0, "half a wave's spans fill the staging tile; halves split on whole iterations");
    __shared__ __attribute__((aligned(16))) char lds_all[WPB * TILE_STG_BYTES];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    char* const scr = lds_all + wv * TILE_STG_BYTES;
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;
    int episode = 0;
    if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions && active) act = (int)a.actions[env];
    if (a.step_size && active) k = (int)a.step_size[env];
    k = min(max(k, 1), 3);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const int d = act & 3;
    const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
    const int tr = s.r + dr - 3, tc = s.c + dc - 3;                  // the build target in plan (interior) coordinates
    const bool inside = (unsigned)tr < 20u && (unsigned)tc < 20u;
    const int tcell = inside ? tr * 20 + tc : 0;
    const int pl = ((const int16_t*)a.plans)[(size_t)s.pidx * GE + tcell];
    // ---- the span: interior rows rlo .. rlo + 9.  The window round (r, c) is rows r - 6 .. r; a move "down" (act 3: row - m) can bring
    // rows down to r - 9 into the window, a move "up" rows up to r + 3
    const int rlo = min(max(s.r - 6 - ((act == 3) ? 3 : 0), 0), 10);
    const int boff = rlo * 40, ab = boff & ~15, mis = boff & 15;      // byte offset of the first row in the record, aligned down; 40 q mod 16 = 0 | 8
    // ... of which THIS tick can touch rows r - 6 - (3 if the action moves down) .. r + (3 if it moves up), inside the map: pieces outside
    // that range are not fetched (three quarters of the envs need 7 of the 10 rows)
    const int qlo = min(max(s.r - 6 - ((act == 3) ? 3 : 0), 0), 19), qhi = min(max(s.r + ((act == 2) ? 3 : 0), 0), 19);
    // ... and columns c - 6 - (3 if it moves left) .. c + (3 if it moves right): a piece (8 cells, possibly the end of one row and the start
    // of the next) none of whose cells lies in that rectangle is not fetched either (a row is 2.5 pieces, the rectangle's part of it 1-2)
    const int clo = min(max(s.c - 6 - ((act == 0) ? 3 : 0), 0), 19), chi = min(max(s.c + ((act == 1) ? 3 : 0), 0), 19);
    uint4 pc[NPC];
    {
        // what the lanes that fetch this env's pieces need to know, in one word: span start (10 bits), the needed byte range inside
        // it (10 + 10 bits), and whether the env reads at all (a freshly reset env is empty)
        const int word = (ab >> 4) | (clo << 5) | (chi << 10) | (qlo << 15) | (qhi << 20) | ((nr || !active) ? (1 << 30) : 0);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int it = 0; it < NPC; ++it) {
            const int P = it * 64 + lane, e = P / NPC, part = P - NPC * e;
            const int we = __builtin_amdgcn_ds_bpermute(e << 2, word);
            const int abe = (we & 31) << 4, cl = (we >> 5) & 31, ch = (we >> 10) & 31, ql = (we >> 15) & 31, qh = (we >> 20) & 31, sk = we >> 30;
            const int off = abe + part * 16;
            // the piece's 8 cells f .. f + 7 lie in row q1 from column f - 20 q1 on and, if they cross a row end, in row q1 + 1 from column 0
            const int f = off >> 1, q1 = f / 20, c1 = f - 20 * q1, e1 = min(c1 + 7, 19), e2 = c1 + 7 - 20;   // e2 >= 0: the last column of the part in row q1 + 1
            const bool hit = (q1 >= ql && q1 <= qh && c1 <= ch && e1 >= cl) || (e2 >= 0 && q1 + 1 >= ql && q1 + 1 <= qh && cl <= e2);
            pc[it] = make_uint4(0u, 0u, 0u, 0u);
            if (!sk && off + 16 <= GE * 2 && hit) {                         // nontemporal: streamed once per tick (k_step3d's measurement)
                const u32x4 t = __builtin_nontemporal_load((const u32x4*)((const char*)a.grid + (size_t)(env0 + e) * (GE * 2) + off));
                pc[it] = make_uint4(t.x, t.y, t.z, t.w);
            }
        }
    }
    // ---- half a wave at a time: the pieces of envs 32 h .. 32 h + 31 into LDS, then the whole tick of those lanes
    const char* const mine = scr + (lane & 31) * SPAN + mis - boff;  // + q * 40 + col * 2: the cell (q, col) of this lane's env, q in [rlo, rlo + 10)
    auto cell_at = [&](int q, int col) -> int {                      // interior coordinates; outside the map: the frame
        const bool in = (unsigned)q < 20u && (unsigned)col < 20u;
        const int qq = min(max(q, rlo), rlo + 9), cc = min(max(col, 0), 19);
        const int v = nr ? 0 : (int)*(const int16_t*)(mine + qq * 40 + cc * 2);
        return in ? v : -1;
    };
    int reward = 0, newh = 0;
    bool done = false, built = false;
    int cellv[K::W];
#pragma unroll
    for (int el = 0; el < K::W; ++el) cellv[el] = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int it = 0; it < NPC / 2; ++it) ((uint4*)scr)[it * 64 + lane] = pc[h * (NPC / 2) + it];
        if ((lane >> 5) == h) {
            const int qa = s.r - 3, ca = s.c - 3;                    // the agent's cell, interior coordinates
            // ---- K3D::step by selects (the formulation of k_step3d / k_transition3d / Roll3D::tick)
            const int n0 = cell_at(qa, ca - 1), n1 = cell_at(qa, ca + 1), n2 = cell_at(qa + 1, ca), n3 = cell_at(qa - 1, ca);   // check_sur: left, right, "up" (row + 1), "down"
            const int c2 = cell_at(qa + 2 * dr, ca + 2 * dc), c3 = cell_at(qa + 3 * dr, ca + 3 * dc);
            const Rule3D u = rules3d<DYN>(s, act, k, n0, n1, n2, n3, c2, c3, active, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
            built = u.built;
            newh = u.newh;
            s.cross += (built && newh <= pl) ? 1 : 0;
            done = u.done;
            reward = u.sel ? reward_check3d(newh, pl) : u.reward0;
            done = done && active;
            s.ep_ret = clamp16(s.ep_ret + reward);
            s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
            if (active) {
                if (a.reward) a.reward[env] = (float)reward;
                if (a.done) a.done[env] = done ? 1 : 0;
                a.hdr[env] = s.pack();
                if (nr) a.episode[env] = episode;
                if (built && !nr) ((int16_t*)a.grid)[(size_t)env * GE + tcell] = (int16_t)newh;
                if (a.stats_on && done) {                            // snac_step: episodic sums
                    const double v = K::iou(nullptr, s, 0);
                    stat_add(a.stat_episodes + env, 1);
                    stat_add(a.stat_return + env, s.ep_ret);
                    stat_add(a.stat_iou_fx + env, __double2ll_rn(v * FX40));
                }
            }
            if (a.obs) {                                             // the window round the NEW position; the built cell shows (a build does not move)
                const int q0 = s.r - 6, cl = s.c - 6;
#pragma unroll
                for (int el = 0; el < K::W; ++el) {
                    const int i = el / 7, j = el - 7 * i;
                    const int v = cell_at(q0 + i, cl + j);
                    cellv[el] = (built && q0 + i == tr && cl + j == tc) ? newh : v;
                }
            }
        }
    }
    for (unsigned long long mk = __ballot(nr); mk; mk &= mk - 1) {   // a reset env's record: empty, but for the cell it built
        const int e = __ffsll(mk) - 1;
        const int tp = __builtin_amdgcn_readlane(built ? tcell : -1, e), nh = __builtin_amdgcn_readlane(newh, e);
        if (lane < 50) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (tp >= 0 && (tp >> 3) == lane) {
                const int hw = tp & 7;
                const uint32_t put = ((uint32_t)nh & 0xFFFFu) << ((hw & 1) * 16);
                if ((hw >> 1) == 0) v.x = put; else if ((hw >> 1) == 1) v.y = put; else if ((hw >> 1) == 2) v.z = put; else v.w = put;
            }
            ((uint4*)a.grid)[(size_t)(env0 + e) * 50 + lane] = v;
        }
    }
    if (!a.obs) return;
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const double v0 = DYN ? c0 / (double)s.tb : c0, v1 = DYN ? c1 / (double)a.total_step : c1;
    emit_tile<OT, ROWS_NT_STEP>(scr, (char*)a.obs + (size_t)env0 * K::D * sizeof(OT), lane, nenv, [&](int el) { return cellv[el]; }, v0, v1);
}

}  // namespace

namespace snac_detail {

void launch_aux2d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {   // masked reset / observe: the canonical layout, plain loads and rows
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    const int tiles = (a.n + 63) / 64;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (dyn) { if (f32) hipLaunchKernelGGL((k_step2d<true, float, 4, false, 64, false, false, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<true, double, 4, false, 64, false, false, true>), grid, block, 0, s, a); }
    else { if (f32) hipLaunchKernelGGL((k_step2d<false, float, 4, false, 64, false, false, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<false, double, 4, false, 64, false, false, true>), grid, block, 0, s, a); }
}

void launch_step2d(const snac_env_desc* d, const KArgs& a, bool half, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    const int tiles = (a.n + 63) / 64;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (a.variant && half) {                                     // half-filled tiles: 32 envs per wave on twice the waves
        const dim3 grid2((unsigned)(((a.n + 31) / 32 + 3) / 4));
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step2d<true, float, 4, true, 32>), grid2, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<true, double, 4, true, 32>), grid2, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step2d<false, float, 4, true, 32>), grid2, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<false, double, 4, true, 32>), grid2, block, 0, s, a); }
    } else if (a.variant) {
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step2d<true, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<true, double, 4, true>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step2d<false, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<false, double, 4, true>), grid, block, 0, s, a); }
    } else {
        // canonical rows: four forms (bit 0: the records by non-temporal loads, bit 1: the rows by non-temporal stores), picked by what fits
        // the Infinity Cache at this batch size (profiles/r06_step_loads.txt); SNAC_STEP2D_FORM forces one
        int form = tune(TN_STEP2D_FORM);
        if (form < 0) {
            if (a.n < tune(TN_STEP2D_PLAIN_LO)) form = 1;            // small batches: round 5's form
            else if (a.n <= tune(TN_STEP2D_RES_HI)) form = 2;        // "resident": plain loads keep the state cached, non-temporal rows stay out of it
            else if (a.n <= tune(TN_STEP2D_PLAIN_HI)) form = 0;      // plain loads, plain rows
            else if (a.n < tune(TN_STEP2D_HUGE_MIN)) form = 1;       // the state streams
            else form = tune(TN_STEP2D_HUGE_FORM);                   // the rows of one tick alone overflow the cache
        }
#define SNAC_STEP2D_LAUNCH(NTL, NTS)                                                                                                                                            \
        do {                                                                                                                                                                   \
            if (dyn) { if (f32) hipLaunchKernelGGL((k_step2d<true, float, 4, false, 64, NTL, NTS>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<true, double, 4, false, 64, NTL, NTS>), grid, block, 0, s, a); } \
            else { if (f32) hipLaunchKernelGGL((k_step2d<false, float, 4, false, 64, NTL, NTS>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<false, double, 4, false, 64, NTL, NTS>), grid, block, 0, s, a); } \
        } while (0)
        switch (form & 3) {
            case 0: SNAC_STEP2D_LAUNCH(false, false); break;
            case 1: SNAC_STEP2D_LAUNCH(true, false); break;
            case 2: SNAC_STEP2D_LAUNCH(false, true); break;
            default: SNAC_STEP2D_LAUNCH(true, true); break;
        }
#undef SNAC_STEP2D_LAUNCH
    }
}

void launch_step3d(const snac_env_desc* d, const KArgs& a, bool span, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    const int tiles = (a.n + 63) / 64;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (a.variant) {
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step3d<true, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3d<true, double, 4, true>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step3d<false, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3d<false, double, 4, true>), grid, block, 0, s, a); }
    } else if (span) {                                              // the canonical rows of large batches: cooperative span loads (k_step3ds)
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step3ds<true, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3ds<true, double, 4>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step3ds<false, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3ds<false, double, 4>), grid, block, 0, s, a); }
    } else {
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step3d<true, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3d<true, double, 4>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step3d<false, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3d<false, double, 4>), grid, block, 0, s, a); }
    }
}

}  // namespace snac_detail
