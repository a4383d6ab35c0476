// k_trans.hip -- k_transition2d / k_transition3d / k_edges3d: single steps and tree edges with gathered rows
#include "snac_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// 3D single step / tree edge without the LDS image.  k_transition stages every 800-byte height map into the bordered LDS
// image and back with 2-byte accesses (profiles/r02_step_*: 524 288 edges in 468 us = 1.5 TB/s of HBM traffic, bound by
// ~25 narrow memory instructions per edge, not by HBM).  But one step changes ONE cell.  Here a wave takes 32 edges:
//   lane = edge   header, counter RNG or the caller's action, the six neighbour / path cells and the plan cell read straight
//                 from the source record (frame cells are -1 by their coordinates), K3D::step by selects;
//   per edge      the record is copied source -> destination in 16-byte lanes (50 lanes x 16 B), the built cell patched in
//                 the lane that holds it; lanes 0..48 gather the 7x7 window from the source record (patched the same way),
//                 lanes 49 / 50 take the scalar slots: one 408-byte row store.
// Four wide memory instructions per edge instead of ~25 narrow ones.  Semantics are K3D::step's (tests compare with the CPU
// restatement exactly as for k_transition); layout variants stay on the generic kernel.
template <bool DYN, typename OT, int WPB, bool INPLACE>
__global__ __launch_bounds__(WPB * 64) void k_transition3d(const KArgs a) {
    using K = K3D<DYN, 8>;
    constexpr int E = 32;
    __shared__ double sc_all[WPB][E][2];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int edge0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (edge0 >= a.n) return;
    const int nedge = min(E, a.n - edge0);
    const bool active = lane < nedge;
    const int edge = edge0 + (active ? lane : 0);
    const int srow = (int)row_of(a.src_index, a.pool, edge), drow = (int)row_of(a.dst_index, a.pool, edge);
    Lane s;
    s.unpack(a.hdr[srow]);
    int episode = a.episode[srow];
    const uint64_t gid = (uint64_t)(a.env_id_base + edge);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions) act = (int)a.actions[edge];
    if (a.step_size) k = (int)a.step_size[edge];
    k = min(max(k, 1), 3);
    const int16_t* const g16 = (const int16_t*)a.grid;
    const int16_t* const src = g16 + (size_t)srow * K::GE;
    // a cell of the source map in bordered coordinates: the frame is -1, a freshly reset env is empty
    auto cell = [&](int R, int C) -> int {
        const bool in = (unsigned)(R - 3) < 20u && (unsigned)(C - 3) < 20u;
        const int v = (in && !nr) ? (int)src[(R - 3) * 20 + (C - 3)] : 0;
        return in ? v : -1;
    };
    const int d = act & 3;
    const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
    const int n0 = cell(s.r, s.c - 1), n1 = cell(s.r, s.c + 1), n2 = cell(s.r + 1, s.c), n3 = cell(s.r - 1, s.c);
    const int c2 = cell(s.r + 2 * dr, s.c + 2 * dc), c3 = cell(s.r + 3 * dr, s.c + 3 * dc);
    const int tr = s.r + dr - 3, tc = s.c + dc - 3;
    const bool inside = (unsigned)tr < 20u && (unsigned)tc < 20u;
    const int tcell = inside ? tr * 20 + tc : 0;
    const int pl = ((const int16_t*)a.plans)[(size_t)s.pidx * K::GE + tcell];
    // K3D::step by selects (the same formulation as Roll3D::tick, without its deferral)
    const Rule3D u = rules3d<DYN>(s, act, k, n0, n1, n2, n3, c2, c3, active, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
    const bool built = u.built;
    const int newh = u.newh;
    s.cross += (built && newh <= pl) ? 1 : 0;
    bool done = u.done;
    const int reward = u.sel ? reward_check3d(newh, pl) : u.reward0;
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (active) {
        if (a.reward) a.reward[edge] = (float)reward;
        if (a.done) a.done[edge] = done ? 1 : 0;
        if (a.stats_on && done) {                                // snac_step: episodic sums
            const double v = K::iou(nullptr, s, 0);
            a.stat_episodes[drow] += 1;
            a.stat_return[drow] += s.ep_ret;
            a.stat_iou_fx[drow] += __double2ll_rn(v * FX40);
        }
    }
    // the two scalar observation slots of every edge -> LDS
    {
        const double c0 = (double)s.cb, c1 = (double)s.cs;
        double (*sc)[2] = sc_all[wv];
        if (lane < E) { sc[lane][0] = DYN ? c0 / (double)s.tb : c0; sc[lane][1] = DYN ? c1 / (double)a.total_step : c1; }
    }
    const int tpatch = built ? tcell : -1;                           // interior index of the cell this step changed
    const int key_r = s.r, key_c = s.c;
    const int wl = lane < K::W ? lane : 0, wi = wl / 7, wj = wl - 7 * wi;
    const uint4* const g4 = (const uint4*)a.grid;
    uint4* const g4w = (uint4*)a.grid;
    OT* const orow = a.obs ? (OT*)a.obs + (size_t)edge0 * K::D + lane : nullptr;
    // U edges at a time: every load of the group (record lanes and window cells, both from the SOURCE records) is issued before
    // the group's first store, so U records are in flight per wave instead of one -- the loop used to be a load -> store ->
    // load chain, the compiler may not move a load over a store into the same array.  Legal by the contract of
    // snac_transition (include/snac_hip.h): a destination row is never the source row of a different edge of the call.
    // INPLACE (identity rows: every snac_step): no record is copied at all -- a step writes its one changed cell, an auto-reset
    // writes the empty map -- so the whole tile's window gathers are issued up front (vmcnt retires in order: a later group's
    // loads would also wait for the row stores in front of them).
    constexpr int U = INPLACE ? 32 : 8;
    for (int e0 = 0; e0 < nedge; e0 += U) {                          // wave-uniform: readlane broadcasts an edge's scalars
        uint4 rec[INPLACE ? 1 : U];
        int wcell[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u, nedge - 1);                    // a short last group reloads its last edge (not stored)
            const int se = __builtin_amdgcn_readlane(srow, e);
            const int tp = __builtin_amdgcn_readlane(tpatch, e), nh = __builtin_amdgcn_readlane(newh, e);
            const bool fresh = __builtin_amdgcn_readlane((int)nr, e) != 0;
            // the 7x7 window around the NEW position, from the source record with the built cell patched in
            wcell[u] = -1;
            if (orow) {
                const int R = __builtin_amdgcn_readlane(key_r, e) - 3 + wi, C = __builtin_amdgcn_readlane(key_c, e) - 3 + wj;
                const bool in = (unsigned)(R - 3) < 20u && (unsigned)(C - 3) < 20u;
                const int idx = in ? (R - 3) * 20 + (C - 3) : 0;
                const int v = (in && !fresh && lane < K::W) ? (int)g16[(size_t)se * K::GE + idx] : 0;
                wcell[u] = in ? (idx == tp ? nh : v) : -1;
            }
            // a step in place (snac_step, or a tree edge onto its own row) changes ONE cell: no record copy
            if constexpr (!INPLACE) {
                const bool copy = fresh || se != __builtin_amdgcn_readlane(drow, e);
                rec[u] = (fresh || lane >= 50 || !copy) ? make_uint4(0u, 0u, 0u, 0u) : g4[(size_t)se * 50 + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u;
            if (e < nedge) {
                const int de = __builtin_amdgcn_readlane(drow, e);
                const int tp = __builtin_amdgcn_readlane(tpatch, e), nh = __builtin_amdgcn_readlane(newh, e);
                const bool copy = __builtin_amdgcn_readlane((int)nr, e) != 0 || (!INPLACE && de != __builtin_amdgcn_readlane(srow, e));
                if (!copy) {
                    if (tp >= 0 && lane == 0) ((int16_t*)a.grid)[(size_t)de * K::GE + tp] = (int16_t)nh;
                } else if (lane < 50) {
                    uint4 v = INPLACE ? make_uint4(0u, 0u, 0u, 0u) : rec[u];
                    if (tp >= 0 && (tp >> 3) == lane) {              // this lane's 8 cells hold the built one
                        const int hw = tp & 7, sh = (hw & 1) * 16;
                        const uint32_t keep = ~(0xFFFFu << sh), put = ((uint32_t)nh & 0xFFFFu) << sh;
                        if ((hw >> 1) == 0) v.x = (v.x & keep) | put;
                        else if ((hw >> 1) == 1) v.y = (v.y & keep) | put;
                        else if ((hw >> 1) == 2) v.z = (v.z & keep) | put;
                        else v.w = (v.w & keep) | put;
                    }
                    g4w[(size_t)de * 50 + lane] = v;
                }
                if (orow && lane < K::D) {
                    const double scal = sc_all[wv][e][lane >= K::W ? min(lane - K::W, 1) : 0];
                    orow[(size_t)e * K::D] = (OT)(lane < K::W ? (double)wcell[u] : scal);
                }
            }
        }
    }
    if (active) { a.hdr[drow] = s.pack(); a.episode[drow] = episode; }
}

// ------------------------------------------------------------------------------------------------
// 3D tree edges with gathered rows, round 4 (snac_transition: Env/3D/DMP_simulator_3d_dynamic_triangle_MCTS.py:195-277, one call per
// edge in script/MCTS/utils/mcts_Qvalue_dynamic.py:88,118).  k_transition3d issues, per edge, one record load, one record store, one
// 49-lane gather of 2-byte window cells and one 408-byte row store: half of its memory instructions are narrow (0.48 of the peak for
// its 2.06 KB per edge).  Here the records of a wave's 32 edges pass through LDS once and every memory instruction is wide:
//   in      the 32 source records (800 bytes each) arrive as 16-byte pieces, lane = piece of the group's 1600 (the owning edge's row by
//           ds_bpermute), 25 loads in flight, and lie in LDS as REC[edge][400 cells];
//   step    lane = edge: the six neighbour / path cells from its record in LDS, K3D::step by selects (k_transition3d's formulation),
//           the built cell patched into the record, the 7x7 window round the NEW position read back cell by cell (ds_read_i16);
//   out     the records leave again as 16-byte pieces (an edge onto its own row writes its one changed cell instead), and the 32
//           observation rows through emit_tile -- the staging tile reuses the records' LDS -- as 16-byte stores, 1 KiB per instruction.
// 25 + 25 + 13 wide memory instructions per 32 edges instead of 128.  Conditions: gathered / scattered rows (an index array given),
// canonical layout, observations wanted, m % 4 == 0 and a 16-byte aligned obs; everything else stays on k_transition3d.
template <bool DYN, typename OT, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_edges3d(const KArgs a) {
    using K = K3D<DYN, 8>;
    constexpr int E = 32, GE = K::GE, RECB = GE * 2;                 // 800 bytes per record
    constexpr int WAVE_BYTES = E * RECB > TILE_STG_BYTES ? E * RECB : TILE_STG_BYTES;
    __shared__ __attribute__((aligned(16))) char lds_all[WPB * WAVE_BYTES];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int edge0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (edge0 >= a.n) return;
    const int nedge = min(E, a.n - edge0);
    const bool active = lane < nedge;
    const int edge = edge0 + (active ? lane : 0);
    char* const rec = lds_all + wv * WAVE_BYTES;
    const int srow = (int)row_of(a.src_index, a.pool, edge), drow = (int)row_of(a.dst_index, a.pool, edge);
    Lane s;
    s.unpack(a.hdr[srow]);
    int episode = a.episode[srow];
    const uint64_t gid = (uint64_t)(a.env_id_base + edge);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    // ---- the records of the group: piece q = lane + 64 p of 1600, edge q / 50, 16-byte lane q % 50 of its record
    {
        const uint4* const g4 = (const uint4*)a.grid;
        uint4 pv[25];
#pragma unroll
        for (int p = 0; p < 25; ++p) {
            const int q = p * 64 + lane, e = q / 50, l = q - 50 * e;
            const int se = __shfl(srow, e);
            const bool fresh = __shfl((int)nr, e) != 0;
            pv[p] = (e < nedge && !fresh) ? g4[(size_t)se * 50 + l] : make_uint4(0u, 0u, 0u, 0u);   // a freshly reset env is empty
        }
#pragma unroll
        for (int p = 0; p < 25; ++p) *(uint4*)(rec + (p * 64 + lane) * 16) = pv[p];
    }
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions) act = (int)a.actions[edge];
    if (a.step_size) k = (int)a.step_size[edge];
    k = min(max(k, 1), 3);
    int16_t* const mine = (int16_t*)(rec + (lane & (E - 1)) * RECB);   // (lanes 32..63 shadow 0..31: nothing of theirs is stored)
    // a cell of the edge's map in bordered coordinates: the frame is -1
    auto cell = [&](int R, int C) -> int {
        const bool in = (unsigned)(R - 3) < 20u && (unsigned)(C - 3) < 20u;
        const int v = (int)mine[in ? (R - 3) * 20 + (C - 3) : 0];
        return in ? v : -1;
    };
    const int d = act & 3;
    const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
    const int tr = s.r + dr - 3, tc = s.c + dc - 3;
    const bool inside = (unsigned)tr < 20u && (unsigned)tc < 20u;
    const int tcell = inside ? tr * 20 + tc : 0;
    const int pl = ((const int16_t*)a.plans)[(size_t)s.pidx * GE + tcell];
    const int n0 = cell(s.r, s.c - 1), n1 = cell(s.r, s.c + 1), n2 = cell(s.r + 1, s.c), n3 = cell(s.r - 1, s.c);
    const int c2 = cell(s.r + 2 * dr, s.c + 2 * dc), c3 = cell(s.r + 3 * dr, s.c + 3 * dc);
    // K3D::step by selects (the formulation of k_transition3d / Roll3D::tick)
    const Rule3D u = rules3d<DYN>(s, act, k, n0, n1, n2, n3, c2, c3, active, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
    const bool built = u.built;
    const int newh = u.newh;
    s.cross += (built && newh <= pl) ? 1 : 0;
    bool done = u.done;
    const int reward = u.sel ? reward_check3d(newh, pl) : u.reward0;
    done = done && active;
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (built) mine[tcell] = (int16_t)newh;                          // the record and the window show the built cell
    if (active) {
        if (a.reward) a.reward[edge] = (float)reward;
        if (a.done) a.done[edge] = done ? 1 : 0;
        a.hdr[drow] = s.pack();
        a.episode[drow] = episode;
        if (a.stats_on && done) {                                    // (snac_step on gathered rows: episodic sums)
            const double v = K::iou(nullptr, s, 0);
            stat_add(a.stat_episodes + drow, 1);
            stat_add(a.stat_return + drow, s.ep_ret);
            stat_add(a.stat_iou_fx + drow, __double2ll_rn(v * FX40));
        }
    }
    // ---- the window round the new position, lane = edge
    int cellv[K::W];
#pragma unroll
    for (int el = 0; el < K::W; ++el) { const int i = el / 7, j = el - 7 * i; cellv[el] = cell(s.r - 3 + i, s.c - 3 + j); }
    // ---- the records leave: 16-byte pieces again; an edge onto its own row (not freshly reset) writes its one changed cell instead
    const bool copy = nr || drow != srow;
    if (active && !copy && built) ((int16_t*)a.grid)[(size_t)drow * GE + tcell] = (int16_t)newh;
    {
        uint4* const g4w = (uint4*)a.grid;
        uint4 pv[25];
#pragma unroll
        for (int p = 0; p < 25; ++p) pv[p] = *(const uint4*)(rec + (p * 64 + lane) * 16);
#pragma unroll
        for (int p = 0; p < 25; ++p) {
            const int q = p * 64 + lane, e = q / 50, l = q - 50 * e;
            const int de = __shfl(drow, e);
            const bool cp = __shfl((int)copy, e) != 0;
            if (e < nedge && cp) g4w[(size_t)de * 50 + l] = pv[p];
        }
    }
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const double v0 = DYN ? c0 / (double)s.tb : c0, v1 = DYN ? c1 / (double)a.total_step : c1;
    emit_tile<OT, ROWS_NT_EDGES>(rec, (char*)a.obs + (size_t)edge0 * K::D * sizeof(OT), lane, nedge, [&](int el) { return cellv[el]; }, v0, v1);
}

// ------------------------------------------------------------------------------------------------
// 2D single step / tree edge without the LDS image.  k_transition expands every 80-byte bit-board into the bordered two-bit
// LDS image and squeezes it back (20 rows per edge, for a window that shows 7 of them and a step that changes one bit).
// Here a wave takes E edges and nothing is staged:
//   lane = edge   header, counter RNG or the caller's action, K2D::step on the agent's row word and the plan's row word;
//                 then the 7 row words around the NEW position, cut to the 7 window columns and re-coded as two-bit cells
//                 (00 empty / 01 brick / 11 frame, as in the LDS image): the whole 7x7 window is 98 bits = 4 registers;
//   per edge      four v_readlane broadcast those registers, lane l < 49 extracts the signed two-bit field at bit 2 l
//                 (0 / 1 / -1), lanes 49 / 50 take the broadcast scalar slots: one 408-byte row store, no load, no LDS;
//   the record    in place (snac_step): only the row word a brick changed is written back.  Gathered / scattered rows
//                 (snac_transition): copied source -> destination three records per instruction (lane = row word), the
//                 changed word patched on the way.
// Semantics are K2D::step's (tests compare with the CPU restatement exactly as for k_transition); layout variants stay on
// the generic kernel.
template <bool DYN, typename OT, int WPB, int E>
__global__ __launch_bounds__(WPB * 64) void k_transition2d(const KArgs a) {
    using K = K2D<DYN, 64>;
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int edge0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (edge0 >= a.n) return;
    const int nedge = min(E, a.n - edge0);
    const bool active = lane < nedge;
    const int edge = edge0 + (active ? lane : 0);
    const int srow = (int)row_of(a.src_index, a.pool, edge), drow = (int)row_of(a.dst_index, a.pool, edge);
    Lane s;
    s.unpack(a.hdr[srow]);
    int episode = a.episode[srow];
    const uint64_t gid = (uint64_t)(a.env_id_base + edge);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions) act = (int)a.actions[edge];
    if (a.step_size) k = (int)a.step_size[edge];
    k = min(max(k, 1), 3);
    const uint32_t* const g32 = (const uint32_t*)a.grid;
    const uint32_t* const src = g32 + (size_t)srow * K::GE;
    const uint32_t* const prow = (const uint32_t*)a.plans + (size_t)s.pidx * K::GE;
    // ---- K2D::step (DMP_Env_2D_dynamic_usedata_plan.py:85-147) on the agent's row word; a freshly reset board is empty
    const int q0 = min(max(s.r - 3, 0), K::GE - 1), bit = min(max(s.c - 3, 0), 19);
    const uint32_t row0 = nr ? 0u : src[q0];
    const bool was = ((row0 >> bit) & 1u) != 0u, planned = ((prow[q0] >> bit) & 1u) != 0u;
    const uint32_t newrow = row0 | (1u << bit);
    const Rule2D u = rules2d(s, act, k, was, planned, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
    const bool drop = active && u.drop;
    const int patch = drop ? q0 : -1;                                // the board row this step changed
    const bool done = u.done;
    const int reward = u.reward;
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (active) {
        if (a.reward) a.reward[edge] = (float)reward;
        if (a.done) a.done[edge] = done ? 1 : 0;
        if (a.stats_on && done) {                                    // snac_step: episodic sums; the boolean IoU needs board and plan
            int inter = 0, uni = 0;
            uint32_t gq[K::GE], pq[K::GE];
#pragma unroll
            for (int q = 0; q < K::GE; ++q) { gq[q] = nr ? 0u : src[q]; pq[q] = prow[q]; }   // all 40 loads in flight together
#pragma unroll
            for (int q = 0; q < K::GE; ++q) {
                const uint32_t g = q == patch ? newrow : gq[q];
                inter += __popc(g & pq[q]); uni += __popc(g | pq[q]);
            }
            const double v = (double)inter / (double)uni;
            a.stat_episodes[drow] += 1;
            a.stat_return[drow] += s.ep_ret;
            a.stat_iou_fx[drow] += __double2ll_rn(v * FX40);
        }
    }
    // ---- the 7x7 window around the new position as 49 two-bit cells: window cell l = 7 i + j is the field at bit 2 l
    uint32_t win[4] = {0u, 0u, 0u, 0u};
    double sc0 = 0.0, sc1 = 0.0;
    if (a.obs) {
        const int sh = s.c - 3;                                      // first window column, bordered: 0..19
        constexpr uint32_t FRAME26 = 0x3800007u;                     // frame columns 0-2 and 23-25 of an interior row
        const uint32_t frm = spread16((FRAME26 >> sh) & 0x7Fu) * 3u; // 11 in every frame cell of the 7 columns
        uint32_t enc[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = s.r - 6 + i;                               // board row of window row i
            const bool in = (unsigned)q < (unsigned)K::GE;
            const int qc = in ? q : 0;
            uint32_t g = (in && !nr) ? src[qc] : 0u;
            g = qc == patch ? newrow : g;                            // a drop does not move: the changed row is window row 3
            enc[i] = in ? (spread16(((g << 3) >> sh) & 0x7Fu) | frm) : 0x3FFFu;
        }
        const uint64_t lo = (uint64_t)enc[0] | ((uint64_t)enc[1] << 14) | ((uint64_t)enc[2] << 28) | ((uint64_t)enc[3] << 42) | ((uint64_t)enc[4] << 56);
        const uint64_t hi = (uint64_t)(enc[4] >> 8) | ((uint64_t)enc[5] << 6) | ((uint64_t)enc[6] << 20);
        win[0] = (uint32_t)lo; win[1] = (uint32_t)(lo >> 32); win[2] = (uint32_t)hi; win[3] = (uint32_t)(hi >> 32);
        const double c0 = (double)s.cb, c1 = (double)s.cs;
        sc0 = DYN ? c0 / (double)s.tb : c0;
        sc1 = DYN ? c1 / (double)a.total_step : c1;
    }
    // ---- the record
    uint32_t* const g32w = (uint32_t*)a.grid;
    if (a.src_index || a.dst_index) {
        // three records per instruction: lane = (edge of the trio, row word)
        const int sub = lane / K::GE, q = lane - sub * K::GE;
        for (int e0 = 0; e0 < nedge; e0 += 3) {
            const int e = e0 + sub;
            const bool ok = sub < 3 && e < nedge;
            const int el = (ok ? e : e0) << 2;
            const int se = __builtin_amdgcn_ds_bpermute(el, srow), de = __builtin_amdgcn_ds_bpermute(el, drow);
            const int pe = __builtin_amdgcn_ds_bpermute(el, patch), fresh = __builtin_amdgcn_ds_bpermute(el, (int)nr);
            const uint32_t ne = (uint32_t)__builtin_amdgcn_ds_bpermute(el, (int)newrow);
            if (ok) {
                uint32_t v = fresh ? 0u : g32[(size_t)se * K::GE + q];
                v = q == pe ? ne : v;
                g32w[(size_t)de * K::GE + q] = v;
            }
        }
    } else if (active) {
        if (nr) for (int q = 0; q < K::GE; ++q) g32w[(size_t)drow * K::GE + q] = q == patch ? newrow : 0u;
        else if (drop) g32w[(size_t)drow * K::GE + q0] = newrow;
    }
    if (active) { a.hdr[drow] = s.pack(); a.episode[drow] = episode; }
    // ---- the observation rows: broadcast, extract, one store per edge
    if (a.obs) {
        const int wsel = min(lane >> 4, 3), wsh = 2 * (lane & 15);
        const int slo0 = (int)(uint32_t)__double_as_longlong(sc0), shi0 = (int)(uint32_t)(__double_as_longlong(sc0) >> 32);
        const int slo1 = (int)(uint32_t)__double_as_longlong(sc1), shi1 = (int)(uint32_t)(__double_as_longlong(sc1) >> 32);
        OT* const orow = (OT*)a.obs + (size_t)edge0 * K::D + lane;
        const bool is_win = lane < K::W;
        for (int e = 0; e < nedge; ++e) {                            // wave-uniform: readlane broadcasts edge e's registers
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)win[0], e), w1 = (uint32_t)__builtin_amdgcn_readlane((int)win[1], e);
            const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)win[2], e), w3 = (uint32_t)__builtin_amdgcn_readlane((int)win[3], e);
            const uint32_t ww = wsel == 0 ? w0 : (wsel == 1 ? w1 : (wsel == 2 ? w2 : w3));
            const int cellv = ((int)((ww >> wsh) << 30)) >> 30;      // signed 2-bit field: 0 / 1 / -1
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane(slo0, e), a1 = (uint32_t)__builtin_amdgcn_readlane(shi0, e);
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane(slo1, e), b1 = (uint32_t)__builtin_amdgcn_readlane(shi1, e);
            const double scal = __longlong_as_double((long long)(((uint64_t)(lane == K::W ? a1 : b1) << 32) | (lane == K::W ? a0 : b0)));
            if (lane < K::D) orow[(size_t)e * K::D] = (OT)(is_win ? (double)cellv : scal);
        }
    }
}


// E edges per wave.  The kernel is bound by HBM traffic from N = 2^19 down to where the launch itself dominates; 32 edges per
// wave were 4 % ahead of 64 there (two rounds of waves: the second round's loads run under the first round's row stores),
// small batches take 16 so that a step() on 4096 envs is still 256 waves.  SNAC_T2D_E overrides (tuning).
template <bool DYN, typename OT>
void launch_trans2d_e(const KArgs& a, hipStream_t s) {
    const int forced = snac_detail::tune(snac_detail::TN_T2D_E);
    const int E = (forced == 16 || forced == 32 || forced == 64) ? forced : (a.n >= 65536 ? 32 : 16);
    const int tiles = (a.n + E - 1) / E;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (E == 64) hipLaunchKernelGGL((k_transition2d<DYN, OT, 4, 64>), grid, block, 0, s, a);
    else if (E == 32) hipLaunchKernelGGL((k_transition2d<DYN, OT, 4, 32>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_transition2d<DYN, OT, 4, 16>), grid, block, 0, s, a);
}


// ------------------------------------------------------------------------------------------------
// k_edges2d (round 5): 2D tree edges with gathered rows, the records through LDS.  k_transition2d reads an edge's source record with eight
// 4-byte loads per LANE (the agent's row word + the seven window rows, every lane in another record) and then once more for the copy to the
// destination row -- and what a step pays for is the number of scattered lane requests (k_step3ds, profiles/r05_step_experiments.txt).
// Here a wave takes 64 edges; every source record (80 bytes = five 16-byte pieces) is fetched ONCE by five neighbouring lanes (piece g of the
// wave's 320 belongs to edge g / 5), lies in LDS ([edge][20] row words) for the transition and the window, and leaves for its destination
// row the same way; the rows go out through emit_tile.  Semantics are k_transition2d's (K2D::step on the agent's row word).  m % 4 = 0 and a
// 16-byte aligned obs; the canonical layout.
template <bool DYN, typename OT, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_edges2d(const KArgs a) {
    using K = K2D<DYN, 64>;
    constexpr int E = 64, GE = K::GE;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB * (TILE_STG_BYTES / 4)];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int edge0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (edge0 >= a.n) return;
    const int nedge = min(E, a.n - edge0);
    const bool active = lane < nedge;
    const int edge = edge0 + (active ? lane : 0);
    uint32_t* const rec = lds_all + wv * (TILE_STG_BYTES / 4);
    const int srow = (int)row_of(a.src_index, a.pool, edge), drow = (int)row_of(a.dst_index, a.pool, edge);
    // ---- the source records: five 16-byte pieces per edge, fetched by neighbouring lanes (plain loads: children share their parents)
    uint4 rv[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int g = i * 64 + lane, e = g / 5, part = g - 5 * e;
        const int se = __builtin_amdgcn_ds_bpermute(e << 2, srow);
        rv[i] = make_uint4(0u, 0u, 0u, 0u);
        if (g < nedge * 5) rv[i] = ((const uint4*)a.grid)[(size_t)se * 5 + part];
    }
    Lane s;
    s.unpack(a.hdr[srow]);
    int episode = a.episode[srow];
    const uint64_t gid = (uint64_t)(a.env_id_base + edge);
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions) act = (int)a.actions[edge];
    if (a.step_size) k = (int)a.step_size[edge];
    k = min(max(k, 1), 3);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const uint32_t* const prow = (const uint32_t*)a.plans + (size_t)s.pidx * GE;
    const int q0 = min(max(s.r - 3, 0), GE - 1), bit = min(max(s.c - 3, 0), 19);
    const uint32_t pword = prow[q0];                                 // the one dependent load: the plan row under the agent (L2)
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
    const Rule2D u = rules2d(s, act, k, was, planned, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
    const bool drop = active && u.drop;
    if (drop) mine[q0] = newrow;
    const bool done = active && u.done;
    const int reward = u.reward;
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (active) {
        if (a.reward) a.reward[edge] = (float)reward;
        if (a.done) a.done[edge] = done ? 1 : 0;
        a.hdr[drow] = s.pack();
        a.episode[drow] = episode;
    }
    if (a.stats_on && __builtin_expect(__any(done), 0)) {            // (snac_step with gathered rows does not exist; kept for completeness)
        if (done) {
            int inter = 0, uni = 0;
            for (int q = 0; q < GE; ++q) { const uint32_t g = mine[q], p = prow[q]; inter += __popc(g & p); uni += __popc(g | p); }
            const double v = (double)inter / (double)uni;
            stat_add(a.stat_episodes + drow, 1);
            stat_add(a.stat_return + drow, s.ep_ret);
            stat_add(a.stat_iou_fx + drow, __double2ll_rn(v * FX40));
        }
    }
    // ---- the (updated) records leave for their destination rows, five neighbouring lanes per record
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int g = i * 64 + lane, e = g / 5, part = g - 5 * e;
        const int de = __builtin_amdgcn_ds_bpermute(e << 2, drow);
        if (g < nedge * 5) ((uint4*)a.grid)[(size_t)de * 5 + part] = ((const uint4*)rec)[g];
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
    const double v0 = DYN ? c0 / (double)s.tb : c0, v1 = DYN ? c1 / (double)a.total_step : c1;
    emit_tile<OT, ROWS_NT_EDGES>((char*)rec, (char*)a.obs + (size_t)edge0 * K::D * sizeof(OT), lane, nedge,
                                 [&](int el) { const int i = el / 7, j = el - 7 * i; return ((int)(wr[i] << (30 - 2 * j))) >> 30; }, v0, v1);
}

}  // namespace

namespace snac_detail {

void launch_trans3d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    const int tiles = (a.n + 31) / 32;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (!a.src_index && !a.dst_index) {   // identity rows (snac_step, or a transition on rows i -> i)
        if (dyn) { if (f32) hipLaunchKernelGGL((k_transition3d<true, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_transition3d<true, double, 4, true>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_transition3d<false, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_transition3d<false, double, 4, true>), grid, block, 0, s, a); }
        return;
    }
    // gathered / scattered rows (tree edges): the records through LDS, every memory instruction wide (k_edges3d); SNAC_EDGES3D=0 keeps
    // them on k_transition3d (A/B timing, tests of both paths)
    const bool edges_off = tune(TN_EDGES3D) == 0;
    if (!edges_off && a.obs && (a.n & 3) == 0 && ((uintptr_t)a.obs & 15) == 0) {
        g_kernel = "k_edges3d";
        const dim3 g2((unsigned)((tiles + 1) / 2)), b2(128);     // two waves per block: 51 KB of LDS, three blocks per CU
        if (dyn) { if (f32) hipLaunchKernelGGL((k_edges3d<true, float, 2>), g2, b2, 0, s, a); else hipLaunchKernelGGL((k_edges3d<true, double, 2>), g2, b2, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_edges3d<false, float, 2>), g2, b2, 0, s, a); else hipLaunchKernelGGL((k_edges3d<false, double, 2>), g2, b2, 0, s, a); }
        return;
    }
    if (dyn) { if (f32) hipLaunchKernelGGL((k_transition3d<true, float, 4, false>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_transition3d<true, double, 4, false>), grid, block, 0, s, a); }
    else { if (f32) hipLaunchKernelGGL((k_transition3d<false, float, 4, false>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_transition3d<false, double, 4, false>), grid, block, 0, s, a); }
}

void launch_trans2d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    // gathered / scattered rows (tree edges): the records through LDS, every memory instruction wide (k_edges2d); SNAC_EDGES2D=0 keeps
    // them on k_transition2d (A/B timing, tests of both paths), SNAC_EDGES2D_MIN=n moves the wave size from which it takes them
    if ((a.src_index || a.dst_index) && tune(TN_EDGES2D) != 0 && a.n >= tune(TN_EDGES2D_MIN) && a.obs && (a.n & 3) == 0 && ((uintptr_t)a.obs & 15) == 0) {
        g_kernel = "k_edges2d";
        const dim3 grid((unsigned)(((a.n + 63) / 64 + 3) / 4)), block(256);
        if (dyn) { if (f32) hipLaunchKernelGGL((k_edges2d<true, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_edges2d<true, double, 4>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_edges2d<false, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_edges2d<false, double, 4>), grid, block, 0, s, a); }
        return;
    }
    if (dyn) f32 ? launch_trans2d_e<true, float>(a, s) : launch_trans2d_e<true, double>(a, s);
    else f32 ? launch_trans2d_e<false, float>(a, s) : launch_trans2d_e<false, double>(a, s);
}

}  // namespace snac_detail
