// k_roll1dl.hip -- k_rollout1dl: 1D fused rollouts of LARGE batches, lane = env (round 6)
#include "snac_dev.h"
#include "rows1d.h"

namespace {

// ------------------------------------------------------------------------------------------------
// The time-parallel kernel (k_rollout1dt: a wave per env, lane = tick) is what small 1D batches need -- 4096 envs are 4096 waves --
// but it pays 5.4 vector instructions per env-step for its scans and ballots and levels off at 7-8.4e10 env-steps/s whatever the batch
// (profiles/r06_1d_pmc.txt: VALU-bound at the crossover with HBM); the tile kernel behind it (k_rollout<K1D<.., 32>>) steps 32 envs per
// wave, divides twice per lane and tick and writes its rows an element per lane.  From ~50 000 envs on there is a wave of 64 envs for
// every SIMD, and the headline kernel's shape fits 1D as well:
//   step      lane l steps ITS env with K1D::step (DMP_Env_1D_static.py:85-136; the rules' pieces: snac_dev.h) on the wave's LDS image
//             (K1D<DYN, 64>: bordered 34-cell rows of int16, odd dword stride per env) -- one instruction per 64 env-steps where the
//             time-parallel form needs its scans;
//   window    5 cells round the new position straight from that row (the frame is stored: -1), the two scalars by the exact
//             reciprocal form (one division per episode: Roll3D, tests/native/recip_check.c);
//   rows      the wave's rows of one tick are ONE run of 64 x 56 bytes (float32: 28): each lane files its 7 values in a staging tile
//             and the run leaves 16 bytes per lane, 3.5 (1.75) store instructions per wave-tick.
// Semantics are K1D's as in the tile kernel (reset, plan pick, iou, episodic sums: the same calls).  Canonical layout, every row written
// (SNAC_OBS_ALL / SNAC_OBS_TILED), N % 4 == 0 and a 16-byte aligned obs; the dispatch table's SNAC_1D_LANE* entries say from which N.
// VARLD > 0: the layout variants of snac_env_desc (rows of a.ld <= VARLD values: rows1d.h's fill / flush), the staging tile sized for VARLD.
template <bool DYN, typename OT, int WPB, bool EXPL, bool NT, bool REC, int VARLD = 0>
__global__ __launch_bounds__(WPB * 64) void k_rollout1dl(const KArgs a) {
    using K = K1D<DYN, 64>;
    constexpr bool VAR = VARLD > 0;
    constexpr int E = 64, D = K::D;
    constexpr int IMG_WORDS = (K::LDS_WORDS + 3) & ~3;               // heights and plans of the wave's 64 envs (K1D's image)
    constexpr int STG_WORDS = E * (VAR ? VARLD : D) * (int)sizeof(OT) / 4;
    constexpr int WAVE_WORDS = IMG_WORDS + STG_WORDS;
    static_assert(WAVE_WORDS % 4 == 0, "16-byte aligned staging tiles");
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB * WAVE_WORDS];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int tile = (int)blockIdx.x * WPB + wv;
    const int env0 = __builtin_amdgcn_readfirstlane(tile * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    uint32_t* const lds = lds_all + wv * WAVE_WORDS;
    char* const stg = (char*)(lds + IMG_WORDS);
    Lane s;
    s.clear();
    s.r = 2;                                                         // idle lanes keep an in-range position and plan row 0
    int episode = 0;
    if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
    asm volatile("" : "+v"(episode));                                // waited for HERE, not in the loop's reset branch behind the row stores
    K::load_grid(lds, a, env0, nenv, lane);
    {   // every lane its own plan row (K1D::load_plan's placement, 64 bytes per env: the lane's own loads instead of a wave-wide loop over the envs)
        const uint32_t* const src = (const uint32_t*)((const int16_t*)a.plans + (size_t)s.pidx * K::GE);
        uint32_t* const dst = lds + K::P_OFF + lane * (K::ES / 2);
        uint32_t pv[K::GE / 2];
#pragma unroll
        for (int q = 0; q < K::GE / 2; ++q) pv[q] = src[q];
#pragma unroll
        for (int q = 0; q < K::GE / 2; ++q) dst[q] = pv[q];
    }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const EnvKeys sk = env_keys(a.key_step, gid), pk = env_keys(a.key_plan, gid);
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
    // software pipeline (a lone wave per SIMD: what is on the chain of a tick is what the pass takes): the counter RNG's word of the NEXT
    // tick and the two cells the next step reads -- the height under the agent, the plan's height there -- are fetched while this tick's
    // rows make their round trip through the staging tile
    int16_t* const hrow = K::hmap(lds) + lane * K::ES;
    const int16_t* const prow = K::plan(lds) + lane * K::ES;
    uint32_t w32 = rng_word(sk, a.t0);
    int hold = (int)hrow[s.r], pcell = (int)prow[s.r - 2];
    Rows1D<OT> rows;
    for (int t = 0; t < a.T; ++t) {
        const size_t row = (size_t)t * (size_t)a.n + (size_t)env0;
        const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
        if (__builtin_expect(__any(nr), 0)) {                        // rare, out of line
            const int old_pidx = s.pidx, old_tb = s.tb;
            if (nr) {
                episode += 1;
                const int pidx = pick_plan<K>(a, pk, episode, old_pidx);
                K::reset(a, s, pidx == old_pidx ? -1 : pidx);        // -1: same plan again (static tables): keep tb, no load
                if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
                dtb = (double)s.tb; rtb = 1.0 / dtb;
            }
            for (unsigned long long m = __ballot(nr); m; m &= m - 1) {
                const int e = __ffsll(m) - 1;
                const int pe = __builtin_amdgcn_readlane(s.pidx, e);
                K::clear(lds, e, lane);
                if (pe != __builtin_amdgcn_readlane(old_pidx, e)) K::load_plan(lds, a, e, pe, lane);   // (a vector load: it waits for the rows stored before it -- once per episode)
            }
            hold = (int)hrow[s.r]; pcell = (int)prow[s.r - 2];       // the fetched cells are another episode's
        }
        // ---- the 1D step, lane = env
        int act = (int)(((w32 >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w32 & 0xffffu) * 3u) >> 16);
        if constexpr (EXPL) {
            if (a.actions) act = na;
            if (a.step_size) k = min(max(nk, 1), 3);
            if (t + 1 < a.T) {                                       // ask for the next tick's bytes before this tick's rows are stored
                if (active && a.actions) na = (int)a.actions[row + (size_t)a.n + lane];
                if (active && a.step_size) nk = (int)a.step_size[row + (size_t)a.n + lane];
            }
        }
        const int r_old = s.r;
        const Rule1D u = rules1d(s, act, k, hold, pcell, a.ts_done, a.brick_gt);   // the rules: snac_dev.h (K1D::step's)
        if (u.drop && active) hrow[r_old] = (int16_t)u.hnew;
        const bool done = active && u.done;
        const int reward = u.reward;
        s.ep_ret = clamp16(s.ep_ret + reward);
        s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
        // ---- the row: the 5 cells round the new position (the frame is in the image: -1) and the two scalars
        int win[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) win[i] = (int)hrow[s.r - 2 + i];
        hold = win[2]; pcell = (int)prow[s.r - 2];                   // the next step's cells
        if (active) {
            if (a.reward) a.reward[row + lane] = (float)reward;
            if (a.done) a.done[row + lane] = done ? 1 : 0;
            if constexpr (REC) {
                if (a.actions_out) a.actions_out[row + lane] = (int8_t)act;
                if (a.step_size_out) a.step_size_out[row + lane] = (int8_t)k;
                if (a.plan_idx_out) a.plan_idx_out[row + lane] = (int16_t)s.pidx;
                if (a.first_out) a.first_out[row + lane] = s.cs == 1 ? 1 : 0;   // first step of its episode
            }
        }
        if (__builtin_expect(__any(done), 0)) {                      // iou :138-151 of the finished episode
            const double v = K::iou(lds, s, lane);
            if (done) { d_eps += 1; d_ret += s.ep_ret; d_iou += __double2ll_rn(v * FX40); }
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
            const int recv[8] = {reward, done ? 1 : 0, s.r, 0, s.cb, s.cs, s.tb, s.pidx};   // SNAC_TAIL_RECORD's values (record_value)
            fill_row1d_var<OT>((OT*)stg + (size_t)lane * LD, a.tail, a.frame_val, win, v0, v1, s.r, prow, recv);
            w32 = rng_word(sk, a.t0 + (uint32_t)t + 1u);
            flush_rows1d_var<OT, NT>(stg, obs0 + (size_t)t * tstride, lane, nenv, LD);
        } else {
            rows.stage(stg, lane, win, v0, v1);
            w32 = rng_word(sk, a.t0 + (uint32_t)t + 1u);             // (behind the staging tile's reads, in front of the stores that wait for them)
            rows.template flush<NT>(obs0 + (size_t)t * tstride, lane, nenv);
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

template <bool DYN, typename OT, bool EXPL, bool REC>
void launch_roll1dl_x(const KArgs& a, hipStream_t s) {
    const int tiles = (a.n + 63) / 64;
    if (a.variant) {                                                 // the staging tile sized for rows of 16 / 38 / 46 values (k_rollout1dt's classes)
        const dim3 g4((unsigned)((tiles + 3) / 4)), b4(256), g2((unsigned)((tiles + 1) / 2)), b2(128);
        if (a.ld <= 16) hipLaunchKernelGGL((k_rollout1dl<DYN, OT, 4, EXPL, false, REC, 16>), g4, b4, 0, s, a);
        else if (a.ld <= 38) hipLaunchKernelGGL((k_rollout1dl<DYN, OT, 2, EXPL, false, REC, 38>), g2, b2, 0, s, a);
        else hipLaunchKernelGGL((k_rollout1dl<DYN, OT, 2, EXPL, false, REC, 46>), g2, b2, 0, s, a);
        return;
    }
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (snac_detail::tune(snac_detail::TN_1D_LANE_NT) != 0) hipLaunchKernelGGL((k_rollout1dl<DYN, OT, 4, EXPL, true, REC>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_rollout1dl<DYN, OT, 4, EXPL, false, REC>), grid, block, 0, s, a);
}
template <bool DYN, typename OT>
void launch_roll1dl_w(const KArgs& a, hipStream_t s) {
    const bool expl = a.actions || a.step_size;
    const bool rec = a.actions_out || a.step_size_out || a.plan_idx_out || a.first_out;   // snac_rollout_rec: the per-step record outputs
    if (expl) rec ? launch_roll1dl_x<DYN, OT, true, true>(a, s) : launch_roll1dl_x<DYN, OT, true, false>(a, s);
    else rec ? launch_roll1dl_x<DYN, OT, false, true>(a, s) : launch_roll1dl_x<DYN, OT, false, false>(a, s);
}

}  // namespace

namespace snac_detail {

void launch_roll1dl(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll1dl_w<true, float>(a, s) : launch_roll1dl_w<true, double>(a, s);
    else f32 ? launch_roll1dl_w<false, float>(a, s) : launch_roll1dl_w<false, double>(a, s);
}

}  // namespace snac_detail
