// k_roll3d.hip -- k_rollout3d: 3D rollouts of small / odd batches
#include "snac_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// 3D fused rollout, software-pipelined around the store stream.  A 3D tile is 8 envs (the height maps cap it), so BASELINE
// config 5 (N = 16 384) runs on 2048 waves = 2 per SIMD.  What bounds such a wave is not its instruction count but WHERE it
// waits (profiles/r02_3d_*, tools/wr_shape3d.hip): vmcnt retires in order, so the wait for any vector load also waits for
// every observation store issued before it.  The generic k_rollout consumes the plan cell of a build in the middle of the
// tick, right behind the previous rows: tick = store latency (~1.2 us under load) + the rest of the transition = 1.5 us.
// Here every wave keeps exactly one tick of stores in flight UNDER its next transition instead:
//   A   LDS reads of the rows of step t-1 (window cells + scalar slots) into registers
//   R   (rare, wave-uniform) auto-reset of envs whose last step returned done; total_brick from an LDS copy of plan_tb
//   B   step t, branch-free, everything that does not need the plan: RNG, the six neighbour / path cells (one LDS round
//       trip), move / build by selects, the one height-map write, done; then the two observation scalars -> LDS
//   W   the ONLY vmcnt wait: the plan cell of step t-1's build target (loaded a whole tick ago, so what is really waited
//       for is the store burst of the previous iteration, by now one transition old) -> reward of step t-1, the running
//       sum for iou(), episodic sums of episodes that ended at t-1
//   S   the store burst: 8 rows + reward / done / record of step t-1
//   L   issue the plan-cell load of step t
// so tick = max(transition, store latency) + the burst's issue.  The reward of a terminal step never depends on the plan
// (0 or -100) and neither does done, so the deferred part is only reward_check and min(height, plan).
// LDS operations of one wave execute in order: A's reads see step t-1's map and scalar slots although B overwrites them
// later in the same iteration.  Semantics are K3D::step's (k_rollout, k_transition and k_aux keep using it; the tests
// compare both paths with the CPU restatement).  Layout variants, OBS_LAST / OBS_NONE, more than TB_MAX plans: generic kernel.

// reward [T][N] float and done [T][N] uint8 of one wave's 8 envs are 32-byte and 8-byte pieces: written per tick they cost
// 15 % of a whole 3D pass (sub-64-byte writes, tools/wr_shape3d.hip).  The pipelined rollout (k_rollout3d: 8 envs
// per wave) stages 16 steps per block in LDS and write whole runs (WPB = 8: 256 B and 64 B).  Two stage halves: ONE barrier per
// 16 steps (a wave has flushed half A before it meets the barrier that releases half B's flush).  Called by every wave of the
// block at the same steps, idle waves included.  benv: the block's first env.
template <int WPB>
__device__ void flush_stage(const KArgs& a, const float* srew, const uint8_t* sdone, int tp, int benv, int wv, int lane) {
    constexpr int BE = WPB * 8;
    __syncthreads();
    const int t0 = tp & ~15, rows = tp - t0 + 1;
    const int env = benv + lane;
    for (int r = wv; r < rows; r += WPB) {
        const int slot = ((t0 + r) & 31) * BE;
        const size_t orow = (size_t)(t0 + r) * (size_t)a.n;
        if (a.reward && lane < BE && env < a.n) a.reward[orow + env] = srew[slot + lane];
        if (a.done) {
            if ((((uintptr_t)a.done | (uintptr_t)a.n) & 3) == 0) {   // dword runs (the caller's array and its rows are 4-byte aligned)
                if (lane < BE / 4 && benv + 4 * lane < a.n) ((uint32_t*)(a.done + orow + benv))[lane] = ((const uint32_t*)(sdone + slot))[lane];
            } else if (lane < BE && env < a.n) a.done[orow + env] = sdone[slot + lane];
        }
    }
}


template <bool DYN, typename OT, int WPB, bool EXPL, bool FULL>
struct Roll3D {
    using K = K3D<DYN, 8>;
    static constexpr int E = 8;
    const KArgs& a;
    uint32_t* lds;
    const int16_t* tbtab;                                            // LDS copy of plan_tb
    float* srew;                                                     // block stage of reward / done: [2][16][WPB * 8]
    uint8_t* sdone;
    const int lane, env0, nenv, wv;
    const bool active;
    static constexpr bool STAGE = WPB >= 4;
    static constexpr int BE = WPB * 8;                               // envs per block
    Lane s;
    int episode = 0, d_eps = 0, d_ret = 0;
    long long d_iou = 0;
    EnvKeys sk, pk;
    int pv[E];                                                       // rows in flight: one window cell per lane and env ...
    double psv[E];                                                   // ... and the scalar slot of lanes 49 / 50
    // what step t-1 left open (per lane = per env)
    int q_pl = 0, q_newh = 0, q_reward = 0, q_act = 0, q_k = 0, q_pidx = 0, q_cross = 0, q_cb = 0, q_tb = 1, q_ret = 0;
    bool q_built = false, q_sel = false, q_done = false, q_first = false;
    // Observation scalars cb / tb and cs / T without a division per tick: with r = RN(1 / d),
    //     q = RN(n * r);  n / d = RN(q + RN(n - q * d) * r)          (one multiplication, two fused multiply-adds)
    // is the correctly rounded quotient for ALL integers 0 <= n <= 32767, 1 <= d <= 32767 -- checked exhaustively
    // (tests/native/recip_check.c, 2^30 pairs) -- so 1 / total_brick is divided once per episode and 1 / total_step once
    // per launch.  total_brick <= 0 (only a hand-made header) takes the plain division.
    double rtb = 0.0, dtb = 1.0, rT = 0.0, dT = 1.0;
    uint32_t wq = 0;                                                 // counter-RNG words of 8 ticks: lane e + 8 j holds (env e, tick + j)
    // EXPL: the caller's actions / step sizes, 16 steps at a time.  A load consumed in the middle of a tick would wait for the
    // burst just issued, so a window's bytes are loaded a window ahead (into pa / pz), put into this wave's LDS slice
    // (sin: [2 halves][16 steps][8 envs] actions, then the same for step sizes) right behind the W wait of the window's last
    // step, and a step reads its byte from LDS.
    int8_t* sin = nullptr;
    int pa[2] = {0, 0}, pz[2] = {0, 0};

    __device__ __forceinline__ Roll3D(const KArgs& a_, uint32_t* lds_, const int16_t* tbtab_, float* srew_, uint8_t* sdone_, int8_t* sin_,
                                      int lane_, int env0_, int nenv_, int wv_)
        : a(a_), lds(lds_), tbtab(tbtab_), srew(srew_), sdone(sdone_), lane(lane_), env0(env0_), nenv(nenv_), wv(wv_), active(lane_ < nenv_),
          sin(sin_) {}

    __device__ __forceinline__ void issue_reads() {                 // A
        const int wl = lane < K::W ? lane : 0;
        const int wi = wl / 7, wj = wl - 7 * wi;
        const char* base = (const char*)lds + (wi * 26 + wj) * 2;
        const double* scp = K::sc(lds) + (lane >= K::W ? min(lane - K::W, 1) : 0);
        const int k0 = K::key0(s);
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const int s0 = __builtin_amdgcn_readlane(k0, u);
            pv[u] = *(const int16_t*)(base + (u * K::ES * 2 + s0));
            psv[u] = scp[2 * u];
        }
    }
    __device__ __forceinline__ void new_tb() { dtb = (double)s.tb; rtb = 1.0 / dtb; }
    __device__ __forceinline__ void auto_reset() {                   // R
        const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
        if (__any(nr)) {
            if (nr) {
                episode += 1;
                const int np = pick_plan<K>(a, pk, episode, s.pidx);
                if (np != s.pidx) { s.pidx = np; s.tb = tbtab[np]; new_tb(); }   // K::reset: a new row brings its total_brick, the same row keeps the header's
                s.r = 3; s.c = 3; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.cross = 0; s.flags = 0;
            }
            for (unsigned long long m = __ballot(nr); m; m &= m - 1) K::clear(lds, __ffsll(m) - 1, lane);
        }
    }
    __device__ __forceinline__ void issue_inputs(int w) {            // global -> registers: steps 16 w .. 16 w + 15 of this wave's envs
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tt = 16 * w + 8 * h + (lane >> 3), e = lane & 7;
            const bool ok = tt < a.T && e < nenv;
            const size_t at = (size_t)tt * (size_t)a.n + (size_t)(env0 + e);
            pa[h] = (ok && a.actions) ? (int)a.actions[at] : 0;
            pz[h] = (ok && a.step_size) ? (int)a.step_size[at] : 1;
        }
    }
    __device__ __forceinline__ void commit_inputs(int w) {           // registers -> LDS half w & 1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = (w & 1) * 128 + 64 * h + lane;
            sin[idx] = (int8_t)pa[h];
            sin[256 + idx] = (int8_t)pz[h];
        }
    }

    // W + S: resolve and write everything of the previous step.  `live`: the env was not reset since (its header is still
    // that episode's).  prev = this lane's slot of the tile's row 0 of that step, prow = its [T][N] row index.
    __device__ __forceinline__ void finish_prev(OT* prev, size_t prow, bool was_reset, int tp) {
        double val[E];
        const bool is_win = lane < K::W;
#pragma unroll
        for (int u = 0; u < E; ++u) val[u] = is_win ? (double)pv[u] : psv[u];
        // ---- W: the first use of q_pl is the iteration's only vmcnt wait
        const bool le = q_newh <= q_pl;
        const int rc = reward_check3d(q_newh, q_pl);                     // reward_check on the built cell (snac_dev.h)
        const int reward = q_sel ? rc : q_reward;
        const int inc = (q_built && le) ? 1 : 0;                     // min(height, plan) grows by one
        if constexpr (EXPL) {
            if ((tp & 15) == 14) commit_inputs((tp + 2) >> 4);       // behind the wait: the next window's bytes have long arrived
        }
        if (!was_reset) { s.cross += inc; s.ep_ret = clamp16(s.ep_ret + (q_sel ? rc : 0)); }
        const bool fin_ep = active && q_done;
        if (__any(fin_ep)) {                                         // iou (:257-276) = sum(min(g, plan)) / (tb + cb - sum)
            const int cross = q_cross + inc;
            const double v = (double)cross / (double)(q_tb + q_cb - cross);
            if (fin_ep) { d_eps += 1; d_ret += q_ret; d_iou += __double2ll_rn(v * FX40); }
        }
        // ---- S
#pragma unroll
        for (int u = 0; u < E; ++u)
            if (lane < K::D && (FULL || u < nenv)) prev[u * K::D] = (OT)val[u];
        if constexpr (STAGE) {
            if (lane < E) {                                          // idle lanes of a ragged tile stage values nobody writes out
                const int slot = (tp & 31) * BE + wv * E + lane;
                srew[slot] = (float)reward;
                sdone[slot] = q_done ? 1 : 0;
            }
        }
        if (active) {
            if constexpr (!STAGE) {
                if (a.reward) a.reward[prow + lane] = (float)reward;
                if (a.done) a.done[prow + lane] = q_done ? 1 : 0;
            }
            if (a.actions_out) a.actions_out[prow + lane] = (int8_t)q_act;
            if (a.step_size_out) a.step_size_out[prow + lane] = (int8_t)q_k;
            if (a.plan_idx_out) a.plan_idx_out[prow + lane] = (int16_t)q_pidx;
            if (a.first_out) a.first_out[prow + lane] = q_first ? 1 : 0;
        }
        if constexpr (STAGE) {
            if ((tp & 15) == 15 || tp == a.T - 1) flush_stage<WPB>(a, srew, sdone, tp, env0 - wv * E, wv, lane);
        }
    }
    // step t; EMIT: the outputs of step t-1 are resolved and written on the way
    template <bool EMIT>
    __device__ __forceinline__ void tick(int t, OT* prev) {
        const size_t row = (size_t)t * (size_t)a.n + (size_t)env0;
        if constexpr (EMIT) issue_reads();
        const bool was_reset = a.auto_reset && q_done;
        auto_reset();
        // ---- B
        // counter RNG: phase 1 keeps 8 of the 64 lanes busy, so every 8th tick ALL lanes hash -- lane e + 8 j the word of
        // (env e, tick t + j) -- and a tick fetches its word with one bpermute
        if ((t & 7) == 0) wq = rng_word(sk, a.t0 + (uint32_t)t + (uint32_t)(lane >> 3));
        const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & 7) + 8 * (t & 7)) << 2, (int)wq);
        int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
        if constexpr (EXPL) {
            const int idx = ((t >> 4) & 1) * 128 + (t & 15) * 8 + (lane & 7);
            if (a.actions) act = (int)sin[idx];
            if (a.step_size) k = min(max((int)sin[256 + idx], 1), 3);
        }
        const int slot = lane & (E - 1);                             // idle lanes only READ some env's map
        int16_t* h = K::hmap(lds) + slot * K::ES + s.r * 26 + s.c;
        const int d = act & 3;
        const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
        const int dl = dr * 26 + dc;
        const int n0 = h[-1], n1 = h[1], n2 = h[26], n3 = h[-26];    // check_sur: left, right, "up" (row + 1), "down"
        const int c2 = h[2 * dl], c3 = h[3 * dl];                    // within the frame: |offset| <= 3 cells
        const int tr = s.r + dr - 3, tc = s.c + dc - 3;              // the build target in plan coordinates
        const bool inside = (unsigned)tr < 20u && (unsigned)tc < 20u;
        const int16_t* plp = (const int16_t*)a.plans + ((size_t)s.pidx * K::GE + (inside ? tr * 20 + tc : 0));
        const bool first = s.cs == 0;
        // the rules: snac_dev.h (every lane steps -- idle lanes read some env's map and store nothing: `active` is applied at the stores)
        const Rule3D u = rules3d<DYN>(s, act, k, n0, n1, n2, n3, c2, c3, true, a.ts_done, a.brick_gt);
        const bool built = u.built;
        const int newh = u.newh;
        if (active && built) h[dl] = (int16_t)newh;
        const bool done = u.done;
        const int reward0 = u.reward0;                               // the part of the reward that does not need the plan
        const bool sel = u.sel;                                      // reward = reward_check(built cell)
        s.ep_ret = clamp16(s.ep_ret + reward0);
        s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
        {   // the two scalar observation slots -> LDS (write_scalars of the generic kernel)
            const double c0 = (double)s.cb, c1 = (double)s.cs;
            double v0 = c0, v1 = c1;
            if (DYN) {
                const double q0 = c0 * rtb, q1 = c1 * rT;
                v0 = __builtin_fma(__builtin_fma(-q0, dtb, c0), rtb, q0);
                v1 = __builtin_fma(__builtin_fma(-q1, dT, c1), rT, q1);
                if (__any(active && s.tb <= 0)) {                    // never in practice; the asm keeps it a branch (no if-conversion)
                    asm volatile("" ::: "memory");
                    v0 = c0 / dtb;
                }
            }
            if (lane < E) { double2 v; v.x = v0; v.y = v1; *(double2*)(K::sc(lds) + 2 * lane) = v; }
        }
        // ---- W, S: the previous step (adds its plan-dependent parts to s.cross / s.ep_ret unless the env was reset since)
        if constexpr (EMIT) finish_prev(prev, row - (size_t)a.n, was_reset, t - 1);
        // ---- L: what this step leaves open; q_cross / q_ret: the running sums without this step's plan-dependent part
        q_pl = *plp;
        q_newh = newh; q_built = built; q_sel = sel; q_reward = reward0; q_done = done; q_act = act; q_k = k; q_pidx = s.pidx;
        q_first = first; q_cross = s.cross; q_cb = s.cb; q_tb = s.tb; q_ret = s.ep_ret;
        if constexpr (EXPL) {
            if ((t & 15) == 15) issue_inputs((t >> 4) + 2);
        }
    }

    __device__ __forceinline__ void run() {
        const int env = env0 + (active ? lane : 0);
        s.clear();
        s.r = 3; s.c = 3;                                            // idle lanes keep an in-range position and plan row 0
        if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
        K::load_grid(lds, a, env0, nenv, lane);
        const uint64_t gid = (uint64_t)(a.env_id_base + env);
        pk = env_keys(a.key_plan, gid);
        sk = env_keys(a.key_step, (uint64_t)(a.env_id_base + env0 + (lane & 7)));   // every lane hashes for env (lane & 7)
        new_tb();
        dT = (double)a.total_step; rT = 1.0 / dT;
        // this lane's slot of the tile's row 0 at step 0, and the distance to the same slot one step later: [T][N][D], or tile-major
        // [ceil(N / 64)][tiled_T][64][D] (SNAC_OBS_TILED: the 8 waves of a 64-env block share one tile region)
        const bool tl = a.obs_mode == SNAC_OBS_TILED;
        OT* const obs = (OT*)a.obs + (tl ? (((size_t)(env0 >> 6) * (size_t)a.tiled_T + (size_t)a.tiled_t0) * 64 + (size_t)(env0 & 63)) * K::D
                                         : (size_t)env0 * K::D) + lane;
        const size_t tstride = tl ? (size_t)64 * K::D : (size_t)a.n * K::D;
        if constexpr (EXPL) { issue_inputs(0); commit_inputs(0); issue_inputs(1); }
        tick<false>(0, nullptr);
        for (int t = 1; t < a.T; ++t) tick<true>(t, obs + (size_t)(t - 1) * tstride);
        issue_reads();
        finish_prev(obs + (size_t)(a.T - 1) * tstride, (size_t)(a.T - 1) * (size_t)a.n + (size_t)env0, false, a.T - 1);
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
};

template <bool DYN, typename OT, int WPB, bool EXPL>
__global__ __launch_bounds__(WPB * 64) void k_rollout3d(const KArgs a) {
    using K = K3D<DYN, 8>;
    constexpr int STAGE_WORDS = WPB >= 4 ? (2 * 16 * WPB * 8 * 5 + 3) / 4 : 0;      // reward float + done byte, two halves of 16 steps
    constexpr int IN_WORDS = EXPL ? WPB * 128 : 0;                                  // 512 bytes of staged inputs per wave
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB * K::LDS_WORDS + TB_MAX / 2 + STAGE_WORDS + IN_WORDS];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous eighth of the env range, so that the rows of
    // one tick that an XCD's L2 collects are neighbours in memory (+7 % at N = 65 536, nothing at 16 384)
    const int chunk = ((int)gridDim.x + 7) >> 3;
    const int blk = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
    const int env0 = __builtin_amdgcn_readfirstlane((blk * WPB + wv) * 8);
    float* srew = (float*)(lds_all + WPB * K::LDS_WORDS + TB_MAX / 2);
    uint8_t* sdone = (uint8_t*)(srew + 2 * 16 * WPB * 8);
    if (env0 >= a.n) {
        // a wave without envs: nothing to step, but its block's flushes are barriers -- keep them company (a whole block
        // without envs simply leaves)
        if constexpr (WPB >= 4) {
            if (blk * WPB * 8 < a.n)
                for (int tp = 0; tp < a.T; ++tp)
                    if ((tp & 15) == 15 || tp == a.T - 1) flush_stage<WPB>(a, srew, sdone, tp, blk * WPB * 8, wv, lane);
        }
        return;
    }
    const int nenv = min(8, a.n - env0);
    uint32_t* lds = lds_all + wv * K::LDS_WORDS;
    // plan_tb -> LDS.  Every wave writes the whole (identical) table itself: its own LDS operations are ordered, so it needs
    // no barrier with the block's other waves.
    int16_t* tbtab = (int16_t*)(lds_all + WPB * K::LDS_WORDS);
    for (int i = lane; i < a.num_plans; i += 64) tbtab[i] = a.plan_tb[i];
    int8_t* sin = (int8_t*)(lds_all + WPB * K::LDS_WORDS + TB_MAX / 2 + STAGE_WORDS) + wv * 512;
    if (nenv == 8) { Roll3D<DYN, OT, WPB, EXPL, true> r(a, lds, tbtab, srew, sdone, sin, lane, env0, nenv, wv); r.run(); }
    else { Roll3D<DYN, OT, WPB, EXPL, false> r(a, lds, tbtab, srew, sdone, sin, lane, env0, nenv, wv); r.run(); }
}


template <bool DYN, typename OT, int WPB>
void launch_roll3d_w(const KArgs& a, hipStream_t s) {
    const int tiles = (a.n + 7) / 8, blocks = (tiles + WPB - 1) / WPB;
    const dim3 grid((unsigned)(((blocks + 7) / 8) * 8)), block(WPB * 64);   // a multiple of 8: the XCD remap covers every tile
    if (a.actions || a.step_size) hipLaunchKernelGGL((k_rollout3d<DYN, OT, WPB, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_rollout3d<DYN, OT, WPB, false>), grid, block, 0, s, a);
}

}  // namespace

namespace snac_detail {

void launch_roll3d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (a.n < 8192) {   // one-wave blocks reach every CU with small batches
        if (dyn) f32 ? launch_roll3d_w<true, float, 1>(a, s) : launch_roll3d_w<true, double, 1>(a, s);
        else f32 ? launch_roll3d_w<false, float, 1>(a, s) : launch_roll3d_w<false, double, 1>(a, s);
    } else if (a.n >= 16384) {   // 64 envs per block: reward / done leave as whole 256-byte / 64-byte runs
        if (dyn) f32 ? launch_roll3d_w<true, float, 8>(a, s) : launch_roll3d_w<true, double, 8>(a, s);
        else f32 ? launch_roll3d_w<false, float, 8>(a, s) : launch_roll3d_w<false, double, 8>(a, s);
    } else {
        if (dyn) f32 ? launch_roll3d_w<true, float, 4>(a, s) : launch_roll3d_w<true, double, 4>(a, s);
        else f32 ? launch_roll3d_w<false, float, 4>(a, s) : launch_roll3d_w<false, double, 4>(a, s);
    }
}

}  // namespace snac_detail
