// k_nodes2d.hip -- 2D tree-search node pools with ONE record per node (round 6): snac_nodes2d_pack / _unpack / snac_transition_nodes2d
#include "snac_dev.h"

// A tree edge (Env/2D/DMP_ENV_2D_dynamic_MCTS.py:117-175: transition(state, action)) reads its parent's state at a RANDOM row of the node
// pool.  In the batch layout of snac_state that state lies in three arrays -- header 16 B, episode counter 4 B, bit board 80 B -- and
// the memory side of the L2 reads whole 128-byte lines (profiles/r06_rd_gran.txt: a lone 16-byte load costs the time of 128 bytes,
// whatever the stride from 128 B up): 1 + 1 + 1.6 lines = 460 bytes fetched for 100, and k_edges2d (round 5) runs at 0.55 of the peak
// by the algorithmic count because of it.  A snac_node2d holds the three in ONE 128-byte aligned line:
//     words 0-3 the header (snac_env_hdr) | 4 the episode counter | 5-7 zero | 8-27 the board (20 row words) | 28-31 zero
// so an edge reads one line and writes one line.  A wave takes 64 edges; a record's eight 16-byte pieces are fetched by eight
// neighbouring lanes (512 pieces = eight load instructions), lie in LDS for the transition and the window -- piece p of edge e at piece
// slot p ^ (e & 7): the lanes' reads of one logical word spread over eight bank groups like k_edges2d's stride of 20 words -- and
// leave for their destination records the same way, header and episode counter included; the rows go out through emit_tile.
// Semantics are k_edges2d's (K2D::step on the agent's row word: DMP_Env_2D_dynamic_usedata_plan.py:85-147), the canonical layout.
// VEC = false: rows written value by value (a wave of m % 4 != 0 edges, an unaligned obs).
namespace {

constexpr int NODE_WORDS = 32, NODE_PIECES = 8, NODE_BOARD = 8;     // a record in 4-byte words / 16-byte pieces; the board's first word

template <bool DYN, typename OT, int WPB, bool VEC, bool NT>
__global__ __launch_bounds__(WPB * 64) void k_edges2dp(const KArgs a) {
    using K = K2D<DYN, 64>;
    constexpr int E = 64, GE = K::GE;
    static_assert(E * NODE_WORDS * 4 <= TILE_STG_BYTES, "the records of a wave's edges fit its staging tile");
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB * (TILE_STG_BYTES / 4)];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int edge0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (edge0 >= a.n) return;
    const int nedge = min(E, a.n - edge0);
    const bool active = lane < nedge;
    const int edge = edge0 + (active ? lane : 0);
    uint32_t* const rec = lds_all + wv * (TILE_STG_BYTES / 4);
    uint4* const nodes = (uint4*)a.grid;                             // snac_node2d[pool]
    const int srow = (int)row_of(a.src_index, a.pool, edge), drow = (int)row_of(a.dst_index, a.pool, edge);
    // ---- the source records: eight 16-byte pieces per edge, fetched by neighbouring lanes (plain loads: children share their parents)
    uint4 rv[NODE_PIECES];
#pragma unroll
    for (int i = 0; i < NODE_PIECES; ++i) {
        const int g = i * 64 + lane, e = g >> 3, part = g & 7;
        const int se = __builtin_amdgcn_ds_bpermute(e << 2, srow);
        rv[i] = make_uint4(0u, 0u, 0u, 0u);
        if (g < nedge * NODE_PIECES) rv[i] = nodes[(size_t)se * NODE_PIECES + part];
    }
    const uint64_t gid = (uint64_t)(a.env_id_base + edge);
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions) act = (int)a.actions[edge];
    if (a.step_size) k = (int)a.step_size[edge];
    k = min(max(k, 1), 3);
#pragma unroll
    for (int i = 0; i < NODE_PIECES; ++i) {
        const int g = i * 64 + lane, e = g >> 3, part = g & 7;
        ((uint4*)rec)[e * NODE_PIECES + (part ^ (e & 7))] = rv[i];
    }
    uint32_t* const mine = rec + lane * NODE_WORDS;
    const int sw = lane & 7;
    auto word = [&](int wd) -> uint32_t& { return mine[(((wd >> 2) ^ sw) << 2) + (wd & 3)]; };   // logical word wd of this lane's record
    Lane s;
    {
        const uint4 h = *(const uint4*)&mine[(0 ^ sw) << 2];
        s.unpack(make_int4((int)h.x, (int)h.y, (int)h.z, (int)h.w));
    }
    int episode = (int)word(4);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
#pragma unroll
        for (int q = 0; q < GE; ++q) word(NODE_BOARD + q) = 0u;      // a freshly reset board is empty
    }
    const uint32_t* const prow = (const uint32_t*)a.plans + (size_t)s.pidx * GE;
    const int q0 = min(max(s.r - 3, 0), GE - 1), bit = min(max(s.c - 3, 0), 19);
    const uint32_t pword = prow[q0];                                 // the one dependent load: the plan row under the agent (L2)
    // ---- the 2D step (rules2d, snac_dev.h) on the agent's row word
    const uint32_t row0 = word(NODE_BOARD + q0);
    const bool was = ((row0 >> bit) & 1u) != 0u, planned = ((pword >> bit) & 1u) != 0u;
    const Rule2D u = rules2d(s, act, k, was, planned, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
    if (active && u.drop) word(NODE_BOARD + q0) = row0 | (1u << bit);
    const bool done = active && u.done;
    const int reward = u.reward;
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (active) {
        if (a.reward) a.reward[edge] = (float)reward;
        if (a.done) a.done[edge] = done ? 1 : 0;
        const int4 h = s.pack();
        *(uint4*)&mine[(0 ^ sw) << 2] = make_uint4((uint32_t)h.x, (uint32_t)h.y, (uint32_t)h.z, (uint32_t)h.w);
        word(4) = (uint32_t)episode;
    }
    // ---- the window round the new position as two-bit codes (00 empty / 01 brick / 11 frame), 14 bits per row -- read before the
    // records leave (the staging tile takes their place)
    uint32_t wr[7];
    {
        const int sh = s.c - 3;                                      // first window column, bordered: 0..19
        constexpr uint32_t FRAME26 = 0x3800007u;                     // frame columns 0-2 and 23-25 of an interior row
        const uint32_t frm = spread16((FRAME26 >> sh) & 0x7Fu) * 3u;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = s.r - 6 + i;                               // board row of window row i
            const bool in = (unsigned)q < (unsigned)GE;
            const uint32_t g = word(NODE_BOARD + (in ? q : 0));
            wr[i] = in ? (spread16(((g << 3) >> sh) & 0x7Fu) | frm) : 0x3FFFu;
        }
    }
    // ---- the (updated) records leave for their destination rows, eight neighbouring lanes per record
#pragma unroll
    for (int i = 0; i < NODE_PIECES; ++i) {
        const int g = i * 64 + lane, e = g >> 3, part = g & 7;
        const int de = __builtin_amdgcn_ds_bpermute(e << 2, drow);
        if (g < nedge * NODE_PIECES) nodes[(size_t)de * NODE_PIECES + part] = ((const uint4*)rec)[e * NODE_PIECES + (part ^ (e & 7))];
    }
    if (!a.obs) return;
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const double v0 = DYN ? c0 / (double)s.tb : c0, v1 = DYN ? c1 / (double)a.total_step : c1;
    auto cell = [&](int el) { const int i = el / 7, j = el - 7 * i; return ((int)(wr[i] << (30 - 2 * j))) >> 30; };
    if constexpr (VEC) {
        asm volatile("" ::: "memory");                               // (every read of the records above, every write of the rows below)
        emit_tile<OT, NT>((char*)rec, (char*)a.obs + (size_t)edge0 * K::D * sizeof(OT), lane, nedge, cell, v0, v1);
    } else if (active) {
        OT* const o = (OT*)a.obs + (size_t)edge * K::D;
#pragma unroll
        for (int el = 0; el < K::W; ++el) o[el] = (OT)cell(el);
        o[K::W] = (OT)v0; o[K::W + 1] = (OT)v1;
    }
}

// batch rows -> node records (PACK) and back: one lane per 16-byte piece
template <bool PACK>
__global__ __launch_bounds__(256) void k_nodes2d_copy(int4* hdr, int32_t* episode, uint4* grid, int nrows, uint4* nodes, int pool, const int32_t* rows,
                                                      const int32_t* node_rows, int m) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)m * NODE_PIECES) return;
    const int i = (int)(g >> 3), part = (int)(g & 7);
    const size_t r = row_of(rows, nrows, i), nr = row_of(node_rows, pool, i);
    if (PACK) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (part == 0) { const int4 h = hdr[r]; v = make_uint4((uint32_t)h.x, (uint32_t)h.y, (uint32_t)h.z, (uint32_t)h.w); }
        else if (part == 1) v.x = (uint32_t)episode[r];
        else if (part < 7) v = grid[r * 5 + (part - 2)];
        nodes[nr * NODE_PIECES + part] = v;
    } else {
        const uint4 v = nodes[nr * NODE_PIECES + part];
        if (part == 0) hdr[r] = make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
        else if (part == 1) episode[r] = (int32_t)v.x;
        else if (part < 7) grid[r * 5 + (part - 2)] = v;
    }
}

template <bool DYN, typename OT>
void launch_edges2dp_part(const KArgs& a, bool vec, hipStream_t s) {
    const dim3 grid((unsigned)(((a.n + 63) / 64 + 3) / 4)), block(256);
    if (!vec) hipLaunchKernelGGL((k_edges2dp<DYN, OT, 4, false, false>), grid, block, 0, s, a);
    else if (snac_detail::tune(snac_detail::TN_NODES2D_NT) != 0) hipLaunchKernelGGL((k_edges2dp<DYN, OT, 4, true, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_edges2dp<DYN, OT, 4, true, false>), grid, block, 0, s, a);
}

template <bool DYN, typename OT>
void launch_edges2dp(const KArgs& a, hipStream_t s) {
    // whole 16-byte pieces of rows (m % 4 = 0, an aligned obs) through emit_tile; otherwise rows value by value.  A wave of m % 4 != 0
    // edges -- five actions per parent make most expansions such -- runs its first m & ~3 edges the fast way and the last one to three as a
    // launch of their own (possible when both index arrays are given: edge i of the tail is edge head + i of the call)
    const bool aligned = !a.obs || ((uintptr_t)a.obs & 15) == 0;
    const int head = a.n & ~3;
    if (aligned && head == a.n) { launch_edges2dp_part<DYN, OT>(a, true, s); return; }
    if (!aligned || head == 0 || !a.src_index || !a.dst_index) { launch_edges2dp_part<DYN, OT>(a, false, s); return; }
    KArgs h = a, t = a;
    h.n = head;
    t.n = a.n - head;
    t.src_index += head; t.dst_index += head;
    if (t.actions) t.actions += head;
    if (t.step_size) t.step_size += head;
    if (t.reward) t.reward += head;
    if (t.done) t.done += head;
    if (t.obs) t.obs = (char*)t.obs + (size_t)head * 51 * sizeof(OT);
    t.env_id_base += head;                                           // the counter RNG is keyed by the edge's index in the call
    launch_edges2dp_part<DYN, OT>(h, true, s);
    launch_edges2dp_part<DYN, OT>(t, false, s);                      // (its rows start where the head's end: 16-byte alignment is not needed here)
}

}  // namespace

extern "C" {

static int nodes_check(const snac_env_desc* d, const snac_state* st, const void* nodes, int32_t pool_rows, int32_t m) {
    using namespace snac_detail;
    if (!d || !st || !nodes) return fail(SNAC_ERR_ARG, "null desc / state / nodes");
    if (d->kind != SNAC_ENV_2D) return fail(SNAC_ERR_UNSUPPORTED, "node records exist for the 2D kinds (a 3D record is 820 bytes: seven lines either way)");
    if (int rc = check_common(d, st)) return rc;
    if (pool_rows < 1) return fail(SNAC_ERR_ARG, "pool_rows must be >= 1");
    if (m < 0) return fail(SNAC_ERR_ARG, "m must be >= 0");
    if (((uintptr_t)nodes & 127) != 0) return fail(SNAC_ERR_ARG, "the node pool must be 128-byte aligned (one record = one line)");
    return SNAC_OK;
}

int snac_nodes2d_pack(const snac_env_desc* d, const snac_state* st, const int32_t* rows, int32_t m, snac_node2d* nodes, int32_t pool_rows,
                      const int32_t* node_rows, void* stream) {
    using namespace snac_detail;
    if (int rc = nodes_check(d, st, nodes, pool_rows, m)) return rc;
    if ((!rows && m > d->num_envs) || (!node_rows && m > pool_rows)) return fail(SNAC_ERR_ARG, "m exceeds the batch / the pool");
    if (m == 0) return SNAC_OK;
    g_kernel = "k_nodes2d_copy";
    hipLaunchKernelGGL((k_nodes2d_copy<true>), dim3((unsigned)(((long long)m * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int4*)st->hdr, st->episode,
                       (uint4*)st->grid, d->num_envs, (uint4*)nodes, pool_rows, rows, node_rows, m);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SNAC_OK : fail_hip(e, "snac_nodes2d_pack");
}

int snac_nodes2d_unpack(const snac_env_desc* d, const snac_node2d* nodes, int32_t pool_rows, const int32_t* node_rows, int32_t m, snac_state* st,
                        const int32_t* rows, void* stream) {
    using namespace snac_detail;
    if (int rc = nodes_check(d, st, nodes, pool_rows, m)) return rc;
    if ((!rows && m > d->num_envs) || (!node_rows && m > pool_rows)) return fail(SNAC_ERR_ARG, "m exceeds the batch / the pool");
    if (m == 0) return SNAC_OK;
    g_kernel = "k_nodes2d_copy";
    hipLaunchKernelGGL((k_nodes2d_copy<false>), dim3((unsigned)(((long long)m * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int4*)st->hdr, st->episode,
                       (uint4*)st->grid, d->num_envs, (uint4*)nodes, pool_rows, rows, node_rows, m);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SNAC_OK : fail_hip(e, "snac_nodes2d_unpack");
}

int snac_transition_nodes2d(const snac_env_desc* d, const snac_state* st, snac_node2d* nodes, int32_t pool_rows, int32_t m, const int32_t* src_index,
                            const int32_t* dst_index, uint32_t t, const int8_t* actions, const int8_t* step_size, void* obs, float* reward, uint8_t* done,
                            void* stream) {
    using namespace snac_detail;
    if (int rc = nodes_check(d, st, nodes, pool_rows, m)) return rc;
    if (int rc = check_layout(d)) return rc;
    if ((!src_index || !dst_index) && m > pool_rows) return fail(SNAC_ERR_ARG, "m exceeds the pool");
    if (m == 0) return SNAC_OK;
    KArgs a = make_args(d, st);
    if (a.variant) return fail(SNAC_ERR_UNSUPPORTED, "snac_transition_nodes2d writes the canonical rows (layout variants: snac_transition)");
    a.pool = pool_rows; a.n = m; a.src_index = src_index; a.dst_index = dst_index; a.grid = nodes; a.hdr = nullptr; a.episode = nullptr;
    a.T = 1; a.t0 = t; a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    a.auto_reset = 0; a.stats_on = 0;
    g_kernel = "k_edges2dp";
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    hipStream_t s = (hipStream_t)stream;
    if (dyn) f32 ? launch_edges2dp<true, float>(a, s) : launch_edges2dp<true, double>(a, s);
    else f32 ? launch_edges2dp<false, float>(a, s) : launch_edges2dp<false, double>(a, s);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SNAC_OK : fail_hip(e, "snac_transition_nodes2d");
}

}  // extern "C"
