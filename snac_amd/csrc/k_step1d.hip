// k_step1d.hip -- k_step1d: the canonical 1D snac_step on identity rows; k_edges1d: 1D tree edges with gathered rows (round 6)
#include "snac_dev.h"
#include "rows1d.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Until the end of round 6 a 1D snac_step ran on the tile kernel k_transition<K1D> at every batch size: 32 envs per wave, two-byte loads
// of the records, rows an element per lane, two divisions per lane: 39 us per tick at 524 288 envs where the tick's bytes -- a row of 56,
// reward, done flag, header in and out, the record read, one cell written: 193 per env -- would take 13 at the HBM peak.  This is
// k_step2d's shape for 1D: a wave takes 64 consecutive envs and every memory instruction is wide --
//   records   the tile's 64 records (64 x 64 bytes, contiguous) arrive as four 16-byte-per-lane loads and are laid out in LDS as K1D's
//             bordered rows (34 int16 per env, odd dword stride, the frame stored as -1);
//   step      lane l steps env l with rules1d() (snac_dev.h) on the cell under its agent; the plan's height there is the one dependent
//             load (L2);
//   rows      the 5 cells round the new position and the two scalars leave through Rows1D (rows1d.h): one run of 64 x 56 bytes.
// Write-back: the header, the ONE changed cell (a reset env: its record and episode counter); episodic sums by no-return atomics.
// Identity rows (snac_step, snac_step_scalar), N % 4 == 0 and a 16-byte aligned obs (or none), the canonical layout and (VAR) its variants;
// odd batches and unaligned outputs stay on k_transition<K1D>; tree edges with gathered rows: k_edges1d below.
// NTL / NTS: the records by non-temporal loads / the rows by non-temporal stores (k_step2d's forms; SNAC_STEP1D_FORM).
// VAR: the layout variants of snac_env_desc (rows1d.h: rows of a.ld <= 46 values); the staging tiles are dynamic LDS sized for the rows' length
// (8-value rows: 8.4 KB per wave, 16 waves per CU; 46-value rows: 28 KB, two waves per block).
// AUX: not a step -- snac_reset with a mask (the masked envs start over, every env reports its observation) and snac_observe on the same loads
// and rows: no action, no rules, no reward / done; a header is written only for an env that was reset.  (k_aux loads every record into LDS
// through two-byte accesses: 45 us per masked reset of 524 288 envs.)
template <bool DYN, typename OT, int WPB, bool NTL, bool NTS, bool VAR = false, bool AUX = false>
__global__ __launch_bounds__(WPB * 64) void k_step1d(const KArgs a) {
    using K = K1D<DYN, 64>;
    constexpr int E = 64, GE = K::GE, ES = K::ES, RW = ES / 2;      // 32 cells per record; 34 per bordered row = 17 dwords
    constexpr int IMG_WORDS = (E * RW + 3) & ~3, STG_WORDS = E * K::D * (int)sizeof(OT) / 4;
    __shared__ __attribute__((aligned(16))) uint32_t lds_fix[VAR ? 4 : WPB * (IMG_WORDS + STG_WORDS)];
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];   // VAR: the launch sizes the staging tiles for the rows' length (step1d_wave_words)
    uint32_t* const lds_all = VAR ? lds_dyn : lds_fix;
    const int WAVE_WORDS = VAR ? IMG_WORDS + ((E * a.ld * (int)sizeof(OT) / 4 + 3) & ~3) : IMG_WORDS + STG_WORDS;
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    uint32_t* const img = lds_all + wv * WAVE_WORDS;                 // [64][17] dwords: the bordered rows
    char* const stg = (char*)(img + IMG_WORDS);
    // ---- every load that does not depend on another: the tile's records (16 bytes per lane), header, episode counter
    uint4 rv[4];
    {
        const uint4* const g4 = (const uint4*)a.grid + (size_t)env0 * 4;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = i * 64 + lane;
            rv[i] = make_uint4(0u, 0u, 0u, 0u);
            if (g < nenv * 4) { const u32x4 t = load_nt_if<NTL>((const u32x4*)(g4 + g)); rv[i] = make_uint4(t.x, t.y, t.z, t.w); }
        }
    }
    Lane s;
    s.clear();
    s.r = 2;                                                         // idle lanes keep an in-range position and plan row 0
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
    const int16_t* const prow = (const int16_t*)a.plans + (size_t)s.pidx * GE;
    const int pl = AUX ? 0 : (int)prow[min(max(s.r - 2, 0), 29)];    // the one dependent load: the plan's height under the agent (L2)
    // ---- the records into K1D's bordered rows: piece p of env e holds cells 8 p .. 8 p + 7 = bordered 2 + 8 p ..: dwords 1 + 4 p .. of the row;
    // dword 0 (bordered cells 0, 1) and dword 16 (cells 32, 33: the record's two padding cells) are the frame
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = i * 64 + lane, e = g >> 2, p = g & 3;
        uint32_t* const d = img + e * RW + 1 + 4 * p;
        d[0] = rv[i].x; d[1] = rv[i].y; d[2] = rv[i].z;
        d[3] = p == 3 ? 0xFFFFFFFFu : rv[i].w;
        if (p == 0) d[-1] = 0xFFFFFFFFu;
    }
    int16_t* const mine = (int16_t*)(img + lane * RW);               // bordered cell index = position
    if (nr) {                                                        // a freshly reset row is empty
#pragma unroll
        for (int q = 1; q < 16; ++q) ((uint32_t*)mine)[q] = 0u;
    }
    // ---- the 1D step (rules1d, snac_dev.h) on the cell under the agent
    const int r_old = s.r;
    Rule1D u;
    u.drop = false; u.done = (s.flags & SNAC_FLAG_NEED_RESET) != 0; u.hnew = 0; u.reward = 0;   // AUX: SNAC_TAIL_RECORD outside a step reports the pending-reset flag
    if constexpr (!AUX) u = rules1d(s, act, k, (int)mine[r_old], pl, a.ts_done, a.brick_gt);
    const bool drop = active && u.drop;
    if (drop) mine[r_old] = (int16_t)u.hnew;
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
            uint4* const gw = (uint4*)a.grid + (size_t)env * 4;
            const uint32_t cell = drop ? ((uint32_t)u.hnew & 0xFFFFu) << (((r_old - 2) & 1) * 16) : 0u;   // empty but for the cell it built
            const int cw = (r_old - 2) >> 1;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                gw[q] = make_uint4(cw == 4 * q ? cell : 0u, cw == 4 * q + 1 ? cell : 0u, cw == 4 * q + 2 ? cell : 0u, cw == 4 * q + 3 ? cell : 0u);
        } else if (drop) {
            ((int16_t*)a.grid)[(size_t)env * GE + r_old - 2] = (int16_t)u.hnew;
        }
    }
    if (!AUX && a.stats_on && __builtin_expect(__any(done), 0)) {    // snac_step: episodic sums; iou :138-151 needs row and plan
        if (done) {
            int a1 = 0, a2 = 0, kk = 0;
            for (int i = 0; i < 30; ++i) {
                const int g = (int)mine[i + 2], p = (int)prow[i];
                a1 += p; a2 += g; kk += max(g - p, 0);
            }
            const int cross = a2 - kk;
            const double v = (double)cross / (double)(a1 + a2 - cross);
            stat_add(a.stat_episodes + env, 1);
            stat_add(a.stat_return + env, s.ep_ret);
            stat_add(a.stat_iou_fx + env, __double2ll_rn(v * FX40));
        }
    }
    if (!a.obs) return;
    int win[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) win[i] = (int)mine[s.r - 2 + i];
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const bool norm = VAR ? (a.sc_norm != 0) : DYN;
    const double v0 = norm ? c0 / (double)s.tb : c0, v1 = norm ? c1 / (double)a.total_step : c1;
    if constexpr (VAR) {
        const int recv[8] = {reward, done ? 1 : 0, s.r, 0, s.cb, s.cs, s.tb, s.pidx};   // SNAC_TAIL_RECORD's values (record_value)
        fill_row1d_var<OT>((OT*)stg + (size_t)lane * a.ld, a.tail, a.frame_val, win, v0, v1, s.r, prow, recv);
        flush_rows1d_var<OT, NTS>(stg, (char*)a.obs + (size_t)env0 * (size_t)a.ld * sizeof(OT), lane, nenv, a.ld);
    } else {
        Rows1D<OT> rows;
        rows.stage(stg, lane, win, v0, v1);
        rows.template flush<NTS>((char*)a.obs + (size_t)env0 * K::D * sizeof(OT), lane, nenv);
    }
}

// k_edges1d: 1D tree edges with gathered rows (snac_transition with index arrays: Env/1D/DMP_Env_1D_*_MCTS*.py transition(state, action)), the
// records through LDS like k_edges2d's.  An edge's source record (64 bytes = four 16-byte pieces) is fetched ONCE by four neighbouring lanes
// (piece g of the wave's 256 belongs to edge g / 4) into K1D's bordered rows, stepped by its lane with rules1d(), and leaves for its
// destination row the same way (out of place, or in place: every read of the wave comes before its first write); header and episode
// counter are gathered and scattered per lane.  VEC: rows as one run per wave (m % 4 == 0, an aligned obs); otherwise value by value.
template <bool DYN, typename OT, int WPB, bool VEC, bool NTS>
__global__ __launch_bounds__(WPB * 64) void k_edges1d(const KArgs a) {
    using K = K1D<DYN, 64>;
    constexpr int E = 64, GE = K::GE, ES = K::ES, RW = ES / 2;
    constexpr int IMG_WORDS = (E * RW + 3) & ~3, STG_WORDS = E * K::D * (int)sizeof(OT) / 4, WAVE_WORDS = IMG_WORDS + STG_WORDS;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB * WAVE_WORDS];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int edge0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (edge0 >= a.n) return;
    const int nedge = min(E, a.n - edge0);
    const bool active = lane < nedge;
    const int edge = edge0 + (active ? lane : 0);
    uint32_t* const img = lds_all + wv * WAVE_WORDS;
    char* const stg = (char*)(img + IMG_WORDS);
    const int srow = (int)row_of(a.src_index, a.pool, edge), drow = (int)row_of(a.dst_index, a.pool, edge);
    // ---- the source records: four 16-byte pieces per edge, fetched by neighbouring lanes (plain loads: children share their parents)
    uint4 rv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = i * 64 + lane, e = g >> 2, p = g & 3;
        const int se = __builtin_amdgcn_ds_bpermute(e << 2, srow);
        rv[i] = make_uint4(0u, 0u, 0u, 0u);
        if (g < nedge * 4) rv[i] = ((const uint4*)a.grid)[(size_t)se * 4 + p];
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
    const int16_t* const prow = (const int16_t*)a.plans + (size_t)s.pidx * GE;
    const int r_in = min(max(s.r, 2), 31);                            // (a hand-made header: stay inside the row)
    const int pl = (int)prow[r_in - 2];                               // the one dependent load: the plan's height under the agent (L2)
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                    // into K1D's bordered rows (k_step1d)
        const int g = i * 64 + lane, e = g >> 2, p = g & 3;
        uint32_t* const d = img + e * RW + 1 + 4 * p;
        d[0] = rv[i].x; d[1] = rv[i].y; d[2] = rv[i].z;
        d[3] = p == 3 ? 0xFFFFFFFFu : rv[i].w;
        if (p == 0) d[-1] = 0xFFFFFFFFu;
    }
    int16_t* const mine = (int16_t*)(img + lane * RW);
    if (nr) {
#pragma unroll
        for (int q = 1; q < 16; ++q) ((uint32_t*)mine)[q] = 0u;
    }
    s.r = r_in;
    const int r_old = s.r;
    const Rule1D u = rules1d(s, act, k, (int)mine[r_old], pl, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
    const bool drop = active && u.drop;
    if (drop) mine[r_old] = (int16_t)u.hnew;
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
            int a1 = 0, a2 = 0, kk = 0;
            for (int i = 0; i < 30; ++i) {
                const int g = (int)mine[i + 2], p = (int)prow[i];
                a1 += p; a2 += g; kk += max(g - p, 0);
            }
            const int cross = a2 - kk;
            const double v = (double)cross / (double)(a1 + a2 - cross);
            stat_add(a.stat_episodes + drow, 1);
            stat_add(a.stat_return + drow, s.ep_ret);
            stat_add(a.stat_iou_fx + drow, __double2ll_rn(v * FX40));
        }
    }
    // ---- the (updated) records leave for their destination rows, four neighbouring lanes per record (the two padding cells are zero)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = i * 64 + lane, e = g >> 2, p = g & 3;
        const int de = __builtin_amdgcn_ds_bpermute(e << 2, drow);
        const uint32_t* const d = img + e * RW + 1 + 4 * p;
        if (g < nedge * 4) ((uint4*)a.grid)[(size_t)de * 4 + p] = make_uint4(d[0], d[1], d[2], p == 3 ? 0u : d[3]);
    }
    if (!a.obs) return;
    int win[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) win[i] = (int)mine[s.r - 2 + i];
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const double v0 = DYN ? c0 / (double)s.tb : c0, v1 = DYN ? c1 / (double)a.total_step : c1;
    if constexpr (VEC) {
        Rows1D<OT> rows;
        rows.stage(stg, lane, win, v0, v1);
        rows.template flush<NTS>((char*)a.obs + (size_t)edge0 * K::D * sizeof(OT), lane, nedge);
    } else if (active) {
        OT* const o = (OT*)a.obs + (size_t)edge * K::D;
#pragma unroll
        for (int i = 0; i < 5; ++i) o[i] = (OT)win[i];
        o[5] = (OT)v0; o[6] = (OT)v1;
    }
}

template <bool DYN, typename OT>
void launch_e1(const KArgs& a, hipStream_t s) {
    const int tiles = (a.n + 63) / 64;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    const bool vec = !a.obs || ((a.n & 3) == 0 && ((uintptr_t)a.obs & 15) == 0);
    if (vec) hipLaunchKernelGGL((k_edges1d<DYN, OT, 4, true, ROWS_NT_EDGES>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_edges1d<DYN, OT, 4, false, false>), grid, block, 0, s, a);
}

template <bool DYN, typename OT>
void launch_a1(const KArgs& a, hipStream_t s) {                     // masked reset / observe: the canonical layout, plain loads and rows
    const int tiles = (a.n + 63) / 64;
    hipLaunchKernelGGL((k_step1d<DYN, OT, 4, false, false, false, true>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, a);
}

template <bool DYN, typename OT>
void launch_s1(const KArgs& a, hipStream_t s) {
    const int tiles = (a.n + 63) / 64;
    if (a.variant) {                                                 // the layout variants: plain loads and rows; four waves per block while their tiles fit 64 KB
        const size_t wave_bytes = (size_t)(((64 * 17 + 3) & ~3) + ((64 * a.ld * (int)sizeof(OT) / 4 + 3) & ~3)) * 4;
        if (wave_bytes * 4 <= 65536) hipLaunchKernelGGL((k_step1d<DYN, OT, 4, false, false, true>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), wave_bytes * 4, s, a);
        else hipLaunchKernelGGL((k_step1d<DYN, OT, 2, false, false, true>), dim3((unsigned)((tiles + 1) / 2)), dim3(128), wave_bytes * 2, s, a);
        return;
    }
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    switch (snac_detail::tune(snac_detail::TN_STEP1D_FORM) & 3) {    // bit 0: non-temporal record loads, bit 1: non-temporal row stores
        case 0: hipLaunchKernelGGL((k_step1d<DYN, OT, 4, false, false>), grid, block, 0, s, a); break;
        case 1: hipLaunchKernelGGL((k_step1d<DYN, OT, 4, true, false>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_step1d<DYN, OT, 4, false, true>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((k_step1d<DYN, OT, 4, true, true>), grid, block, 0, s, a); break;
    }
}

}  // namespace

namespace snac_detail {

void launch_step1d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_s1<true, float>(a, s) : launch_s1<true, double>(a, s);
    else f32 ? launch_s1<false, float>(a, s) : launch_s1<false, double>(a, s);
}

void launch_aux1d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_a1<true, float>(a, s) : launch_a1<true, double>(a, s);
    else f32 ? launch_a1<false, float>(a, s) : launch_a1<false, double>(a, s);
}

void launch_edges1d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_e1<true, float>(a, s) : launch_e1<true, double>(a, s);
    else f32 ? launch_e1<false, float>(a, s) : launch_e1<false, double>(a, s);
}

}  // namespace snac_detail
