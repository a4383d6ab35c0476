// k_roll2d.hip -- k_rollout2d: the headline kernel
#include "snac_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// 2D fused rollout for full-width tiles (round 3; the headline kernel).  k_rollout's phase 2 builds ONE observation row per
// wave-instruction -- lanes 0..50 each fetch a cell of the same env -- so a wave-tick of 64 envs costs 64 x ~12 instructions and 64
// stores of 408 B (f64) / 204 B (f32): with float32 rows the pass is bound by instruction issue (1.79 ms where HBM would allow 1.2,
// profiles/r02j_configs.txt).  Here phase 2 is lane-per-env as well:
//   extract   lane l reads the 7 row words of ITS env's window from the bordered two-bit image (7 ds_read_b64), shifts them to the
//             window's first column, and turns the 49 two-bit fields into values with one v_bfe_i32 + one convert each: 112
//             vector instructions per wave-tick for all 64 envs, instead of 64 x 12;
//   transpose the 51 values of lane l go to row l of a staging tile in LDS ([env][51], odd dword stride: conflict-free);
//   flush     the staged tile is the tile's contiguous piece of obs[t] (64 x 51 values = 13 056 B of float32, 26 112 B of float64 in
//             two halves of 32 envs), read back 16 bytes per lane (ds_read_b128) and stored with global_store_dwordx4: 1 KiB per
//             store instruction, 13 (f32) / 26 (f64) stores per wave-tick instead of 64.
// Further differences from k_rollout, all outside the semantics (K2D::step's, tests compare both kernels with the CPU restatement):
//   * every wave keeps its 64 lanes' CURRENT plan rows in LDS (pl[row * 65 + lane], 5 KB; a step reads its plan bit there), and an
//     env that starts over on a new row has the row's 20 words and its total_brick fetched through the SCALAR cache (s_load counts
//     in lgkmcnt, not vmcnt: k_rollout's per-env plan reload is a vector load and waits for every row stored before it) and written
//     into its column by its own lane.  (Round 3 kept the whole table, <= 512 rows, in the block's LDS: 42 KB per block, nothing
//     gained -- 2.314 against 2.310 ms per pass -- and tables from generate_plans() fell back to the tile kernel.)  Tables of any
//     size take this kernel: 2000 rows 2.33 ms, 32 767 rows 2.42 (the rows then miss the scalar cache), profiles/r04_2d_table_ab.txt;
//   * the boolean IoU (script/DQN/2d/DQN_2d_dynamic.py:63-71) is kept incrementally per lane -- |P and G| and |G| change by at most
//     one per drop, |P or G| = |P| + |G| - |P and G| -- instead of a 20-row popcount loop whenever some env of the wave finishes;
//   * cb / tb and cs / T by the exact reciprocal form of Roll3D (one division per episode instead of two per tick);
//   * EXPL: the caller's action / step-size bytes of tick t + 1 are requested a tick ahead (their latency is hidden; the wait for
//     them is still a `vmcnt(0)` across the loop's back edge, i.e. one drain of the rows per tick, as in the tile kernel).
// Tiles of 64 envs -- N >= 65 536 (pick_tile), float32 rows already from N = 32 768 (launch()) --, N % 4 == 0 and a 16-byte aligned obs
// (the 16-byte stores), canonical layout, every observation written (SNAC_OBS_ALL / SNAC_OBS_TILED): everything else stays on
// k_rollout.


template <bool DYN, typename OT, int WPB, bool EXPL, bool VAR>
__global__ __launch_bounds__(WPB * 64) void k_rollout2d(const KArgs a) {
    using K = K2D<DYN, 64>;
    constexpr int E = 64, D = K::D, RS = K::RS, GE = K::GE;
    constexpr int IMG_WORDS = 26 * RS * 2;                           // the bordered two-bit image: 26 rows x 65 x 8 B
    constexpr int PL_WORDS = (GE * 65 + 3) & ~3;                     // the lanes' plan rows [20][65]
    constexpr int CMP_WORDS = VAR ? E * VAR_CMP_WORDS : 0;           // layout variants: the compact records of emit_rows_var
    constexpr int STG_WORDS = (VAR ? VAR_STG_BYTES : TILE_STG_BYTES) / 4;
    constexpr int WAVE_WORDS = IMG_WORDS + STG_WORDS + PL_WORDS + CMP_WORDS;   // + the staging tile of emit_tile / emit_rows_var
    static_assert(IMG_WORDS % 4 == 0 && WAVE_WORDS % 4 == 0, "16-byte aligned staging tiles");
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB * WAVE_WORDS];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int tile = (int)blockIdx.x * WPB + wv;
    const int env0 = __builtin_amdgcn_readfirstlane(tile * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    uint32_t* const lds = lds_all + wv * WAVE_WORDS;
    uint64_t* const cells = K::cells(lds);
    char* const stg = (char*)(lds + IMG_WORDS);
    uint32_t* const pl = lds + IMG_WORDS + STG_WORDS;
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;                                                // idle lanes keep an in-range position and plan row 0
    int episode = 0;
    if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
    K::load_grid(lds, a, env0, nenv, lane);
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const EnvKeys sk = env_keys(a.key_step, gid), pk = env_keys(a.key_plan, gid);
    // |P|, |G|, |P and G| of the lane's env as the launch finds them
    int pcnt = 0, gcnt = 0, inter = 0;
    {
        const uint32_t* const prow = (const uint32_t*)a.plans + (size_t)s.pidx * GE;   // (idle lanes: row 0)
        for (int q = 0; q < GE; ++q) {
            const uint32_t p = prow[q];
            pl[q * 65 + lane] = p; pcnt += __popc(p);
            const uint32_t g = active ? K::decode_row(cells[(q + 3) * RS + lane]) : 0u;
            gcnt += __popc(g); inter += __popc(g & p);
        }
    }
    double dtb = (double)s.tb, rtb = 1.0 / dtb;
    const double dT = (double)a.total_step, rT = 1.0 / dT;
    int d_eps = 0, d_ret = 0;
    long long d_iou = 0;
    // this tile's first byte of step 0, and the distance to the same place one step later: [T][N][D], or tile-major
    const bool tl = a.obs_mode == SNAC_OBS_TILED;
    const int LD = VAR ? a.ld : D;                                   // values per row: the layout variants append a tail
    char* const obs0 = (char*)a.obs + (tl ? (((size_t)(env0 >> 6) * (size_t)a.tiled_T + (size_t)a.tiled_t0) * 64 + (size_t)(env0 & 63)) * LD : (size_t)env0 * LD) * sizeof(OT);
    const size_t tstride = (tl ? (size_t)64 * LD : (size_t)a.n * LD) * sizeof(OT);
    int na = 0, nk = 1;                                              // EXPL: the bytes of the coming tick
    if constexpr (EXPL) {
        if (active && a.actions) na = (int)a.actions[(size_t)env0 + lane];
        if (active && a.step_size) nk = (int)a.step_size[(size_t)env0 + lane];
    }
    for (int t = 0; t < a.T; ++t) {
        const size_t row = (size_t)t * (size_t)a.n + (size_t)env0;
        const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
        if (__builtin_expect(__any(nr), 0)) {                        // rare, out of line
            bool fresh = false;                                      // a new plan row (K2D::reset: it brings its total_brick; the same row keeps the header's)
            if (nr) {
                const int old_pidx = s.pidx;
                episode += 1;
                const int pidx = pick_plan<K>(a, pk, episode, old_pidx);
                if (pidx != old_pidx) { fresh = true; s.pidx = pidx; }
                s.r = 3; s.c = 3; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.cross = 0; s.flags = 0;
                gcnt = 0; inter = 0;
            }
            for (unsigned long long m = __ballot(nr); m; m &= m - 1) K::clear(lds, __ffsll(m) - 1, lane);
            for (unsigned long long m = __ballot(fresh); m; m &= m - 1) {
                const int e = __ffsll(m) - 1;
                const int pe = __builtin_amdgcn_readlane(s.pidx, e);   // wave-uniform: the row and its total_brick come through the scalar cache
                cmem_u32* const src = (cmem_u32*)(uintptr_t)a.plans + (size_t)pe * GE;
                cmem_u32* const tbw = (cmem_u32*)(uintptr_t)a.plan_tb + (pe >> 1);
                uint32_t rw[GE];
#pragma unroll
                for (int q = 0; q < GE; ++q) rw[q] = src[q];
                const int tbv = (int)(int16_t)((*tbw) >> ((pe & 1) * 16));
                int pc = 0;
#pragma unroll
                for (int q = 0; q < GE; ++q) pc += __popc(rw[q]);
                if (lane == e) {
#pragma unroll
                    for (int q = 0; q < GE; ++q) pl[q * 65 + lane] = rw[q];
                    s.tb = tbv; pcnt = pc;
                    dtb = (double)tbv; rtb = 1.0 / dtb;
                }
            }
        }
        // ---- phase 1: K2D::step (DMP_Env_2D_dynamic_usedata_plan.py:85-147), plan bit from the block's table
        const uint32_t w32 = rng_word(sk, a.t0 + (uint32_t)t);
        int act = (int)(((w32 >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w32 & 0xffffu) * 3u) >> 16);
        if constexpr (EXPL) {
            if (a.actions) act = na;
            if (a.step_size) k = min(max(nk, 1), 3);
            if (t + 1 < a.T) {                                       // ask for the next tick's bytes before this tick's rows are stored
                if (active && a.actions) na = (int)a.actions[row + (size_t)a.n + lane];
                if (active && a.step_size) nk = (int)a.step_size[row + (size_t)a.n + lane];
            }
        }
        uint64_t* const cw = cells + s.r * RS + lane;
        const uint64_t w = *cw;
        const int off = 2 * s.c;
        const bool was = ((w >> off) & 1ull) != 0ull;
        const bool planned = ((pl[(s.r - 3) * 65 + lane] >> (s.c - 3)) & 1u) != 0u;
        const bool first = s.cs == 0;
        const Rule2D u = rules2d(s, act, k, was, planned, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
        if (u.drop) {
            if (active) *cw = w | (1ull << off);
            gcnt += was ? 0 : 1;                                     // the running counts of the boolean IoU
            inter += (!was && planned) ? 1 : 0;
        }
        const bool done = active && u.done;
        const int reward = u.reward;
        s.ep_ret = clamp16(s.ep_ret + reward);
        s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
        if (active) {
            if (a.reward) a.reward[row + lane] = (float)reward;
            if (a.done) a.done[row + lane] = done ? 1 : 0;
            if (a.actions_out) a.actions_out[row + lane] = (int8_t)act;
            if (a.step_size_out) a.step_size_out[row + lane] = (int8_t)k;
            if (a.plan_idx_out) a.plan_idx_out[row + lane] = (int16_t)s.pidx;
            if (a.first_out) a.first_out[row + lane] = first ? 1 : 0;
        }
        if (__builtin_expect(__any(done), 0)) {                      // boolean IoU of the finished episode
            if (done) {
                const double v = (double)inter / (double)(pcnt + gcnt - inter);
                d_eps += 1; d_ret += s.ep_ret; d_iou += __double2ll_rn(v * FX40);
            }
        }
        // ---- phase 2, lane-per-env: the window's 7 row words, cut to its first column (14 bits = 7 two-bit cells each)
        uint32_t wr[7];
        {
            const uint64_t* const wp = cells + (s.r - 3) * RS + lane;
            const int sh = 2 * (s.c - 3);
#pragma unroll
            for (int i = 0; i < 7; ++i) wr[i] = (uint32_t)(wp[i * RS] >> sh);
        }
        double v0 = (double)s.cb, v1 = (double)s.cs;
        if (VAR ? (a.sc_norm != 0) : DYN) {                          // cb / tb, cs / T: correctly rounded (Roll3D, tests/native/recip_check.c)
            const double c0 = v0, c1 = v1, q0 = c0 * rtb, q1 = c1 * rT;
            v0 = __builtin_fma(__builtin_fma(-q0, dtb, c0), rtb, q0);
            v1 = __builtin_fma(__builtin_fma(-q1, dT, c1), rT, q1);
            if (__builtin_expect(__any(active && s.tb <= 0), 0)) {   // only a hand-made header; the asm keeps it a branch
                asm volatile("" ::: "memory");
                v0 = c0 / dtb;
            }
        }
        if constexpr (VAR) {
            const int recv[8] = {reward, done ? 1 : 0, s.r, s.c, s.cb, s.cs, s.tb, s.pidx};   // SNAC_TAIL_RECORD's values (record_value)
            emit_rows_var<OT>(stg, pl + PL_WORDS, obs0 + (size_t)t * tstride, lane, nenv, LD, a.tail, a.frame_val, wr, v0, v1, recv,
                              [&](int e, int row) { return pl[row * 65 + e]; });
        } else {
            // (float32 rows leave as non-temporal stores: 1.195 -> 1.168 ms per headline pass; float64 rows are level either way: r06_edges.txt)
            emit_tile<OT, sizeof(OT) == 4>(stg, obs0 + (size_t)t * tstride, lane, nenv,
                          [&](int el) { const int i = el / 7, j = el - 7 * i; return ((int)(wr[i] << (30 - 2 * j))) >> 30; },   // signed 2-bit field: 0 / 1 / -1
                          v0, v1);
        }
    }
    K::store_grid(lds, a, env0, nenv, lane);
    if (active) {
        a.hdr[env] = s.pack();
        a.episode[env] = episode;
        if (d_eps) {
            a.stat_episodes[env] += d_eps;
            a.stat_return[env] += d_ret;
            a.stat_iou_fx[env] += d_iou;
        }
    }
}


template <bool DYN, typename OT>
void launch_roll2d_w(const KArgs& a, hipStream_t s) {
    const int tiles = (a.n + 63) / 64;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    // the layout variants (a.variant: frame value, scalar form, row tail) are their own instantiations
    const bool expl = a.actions || a.step_size;
    if (a.variant) {
        if (expl) hipLaunchKernelGGL((k_rollout2d<DYN, OT, 4, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_rollout2d<DYN, OT, 4, false, true>), grid, block, 0, s, a);
    } else {
        if (expl) hipLaunchKernelGGL((k_rollout2d<DYN, OT, 4, true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_rollout2d<DYN, OT, 4, false, false>), grid, block, 0, s, a);
    }
}

}  // namespace

namespace snac_detail {

void launch_roll2d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll2d_w<true, float>(a, s) : launch_roll2d_w<true, double>(a, s);
    else f32 ? launch_roll2d_w<false, float>(a, s) : launch_roll2d_w<false, double>(a, s);
}

}  // namespace snac_detail
