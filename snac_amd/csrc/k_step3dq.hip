// k_step3dq.hip -- k_step3dq: the canonical 3D step() of trainer-sized batches, 16 envs per wave, four lanes per env
#include "snac_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// k_step3dq (round 5).  At 65 536 envs and below a 3D tick is ONE wave's chain of dependent trips (k_step3d: header -> seven row loads
// per lane -> transition -> the rows of the envs that moved again -> 26 row stores; 22 800 cycles of wave life at one wave per SIMD,
// profiles/r05_step_experiments.txt part 3), and k_step3ds' coalesced span loads -- one trip instead of two -- lose there what they gain,
// because its wave has to take its 64 envs through the staging tile half a wave at a time.  Here a wave takes 16 envs, FOUR lanes per
// env:
//   * all four lanes of an env hold the env's state and do the (cheap) transition redundantly -- no broadcast, no idle lanes to wait for;
//   * the span of ten rows a tick can need (k_step3ds: 26 pieces of 16 bytes per env, picked by the action, pieces the tick cannot touch
//     skipped) is fetched by the wave's lanes together, 416 pieces in seven load instructions, into 6.5 KB of LDS -- no halves;
//   * the 49 window cells round the new position are extracted by the env's four lanes, 13 / 12 / 12 / 12 cells each, straight into the
//     rows' staging tile (which takes the spans' place: every cell is read into registers first), and the wave's 16 rows leave as one
//     run of 16 x 408 bytes, 16 bytes per lane.
// Four times the waves of k_step3d (4096 at 65 536 envs: four per SIMD), each with a quarter of its loads, a quarter of its window
// cells and a quarter of its row stores.  Semantics are K3D::step's in k_step3ds' formulation.  Canonical layout, identity rows,
// N % 4 = 0, aligned obs; the dispatch table's SNAC_STEP3D_QUARTER_* entries say for which N.
// NTL / NTS: the spans by non-temporal loads / the rows by non-temporal stores.  Which pays depends on what fits the 256 MB Infinity Cache
// (profiles/r06_step_loads.txt): while the STATE fits (below ~360 000 envs) plain loads keep it there for the next tick and non-temporal
// rows stay out of its way ("resident": 65 536 envs 12.9 -> 10.4 us per tick); from there to ~600 000 envs the state streams (non-temporal
// loads) and plain rows do best; beyond, the rows alone overflow the cache and must not go through it.  The launch picks the form.
// AUX: not a step -- snac_reset with a mask (the masked envs start over, every env reports its observation) and snac_observe on the same span
// loads and rows: no action (the span is the window's seven rows), no rules, no reward / done; a header is written only for an env that was
// reset (k_aux: 415 us per masked reset of 524 288 envs).
template <bool DYN, typename OT, int WPB, bool NTL, bool NTS, bool AUX = false>
__global__ __launch_bounds__(WPB * 64) void k_step3dq(const KArgs a) {
    using K = K3D<DYN, 8>;
    constexpr int E = 16, GE = K::GE, NPC = 26, SPAN = NPC * 16, NIT = (E * NPC + 63) / 64;   // pieces and bytes per env; load instructions per wave
    constexpr int ROWB = K::D * (int)sizeof(OT), WAVE_LDS = E * SPAN;                         // 6656 B: the spans, then the 16 rows (6528 / 3264 B)
    static_assert(E * ROWB <= WAVE_LDS, "the rows' staging tile takes the spans' place");
    __shared__ __attribute__((aligned(16))) char lds_all[WPB * WAVE_LDS];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const int le = lane >> 2, part = lane & 3;                       // the lane's env within the wave; which quarter of the env's work
    const bool active = le < nenv;
    const int env = env0 + (active ? le : 0);
    char* const scr = lds_all + wv * WAVE_LDS;
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;
    int episode = 0;
    if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    int act = -1, k = 1;                                             // (AUX: no action -- none of the span's extensions below applies)
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
    const int d = act & 3;
    const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
    const int tr = s.r + dr - 3, tc = s.c + dc - 3;                  // the build target in plan (interior) coordinates
    const bool inside = (unsigned)tr < 20u && (unsigned)tc < 20u;
    const int tcell = inside ? tr * 20 + tc : 0;
    const int pl = AUX ? 0 : (int)((const int16_t*)a.plans)[(size_t)s.pidx * GE + tcell];
    // ---- the span (k_step3ds): interior rows rlo .. rlo + 9; of them this tick can touch rows qlo .. qhi and columns clo .. chi
    const int rlo = min(max(s.r - 6 - ((act == 3) ? 3 : 0), 0), 10);
    const int boff = rlo * 40, ab = boff & ~15, mis = boff & 15;
    const int qlo = min(max(s.r - 6 - ((act == 3) ? 3 : 0), 0), 19), qhi = min(max(s.r + ((act == 2) ? 3 : 0), 0), 19);
    const int clo = min(max(s.c - 6 - ((act == 0) ? 3 : 0), 0), 19), chi = min(max(s.c + ((act == 1) ? 3 : 0), 0), 19);
    {
        const int word = (ab >> 4) | (clo << 5) | (chi << 10) | (qlo << 15) | (qhi << 20) | ((nr || !active) ? (1 << 30) : 0);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        uint4 pc[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int P = it * 64 + lane, e = min(P / NPC, E - 1), pp = P - NPC * e;   // piece P of the wave belongs to env P / 26
            const int we = __builtin_amdgcn_ds_bpermute(e << 4, word);   // (lane 4 e: the first of the env's four)
            const int abe = (we & 31) << 4, cl = (we >> 5) & 31, ch = (we >> 10) & 31, ql = (we >> 15) & 31, qh = (we >> 20) & 31, sk = we >> 30;
            const int off = abe + pp * 16;
            const int f = off >> 1, q1 = f / 20, c1 = f - 20 * q1, e1 = min(c1 + 7, 19), e2 = c1 + 7 - 20;
            const bool hit = (q1 >= ql && q1 <= qh && c1 <= ch && e1 >= cl) || (e2 >= 0 && q1 + 1 >= ql && q1 + 1 <= qh && cl <= e2);
            pc[it] = make_uint4(0u, 0u, 0u, 0u);
            if (P < E * NPC && !sk && off + 16 <= GE * 2 && hit) {
                const u32x4 t = load_nt_if<NTL>((const u32x4*)((const char*)a.grid + (size_t)(env0 + e) * (GE * 2) + off));
                pc[it] = make_uint4(t.x, t.y, t.z, t.w);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (it * 64 + lane < E * NPC) ((uint4*)scr)[it * 64 + lane] = pc[it];
    }
    const char* const mine = scr + le * SPAN + mis - boff;           // + q * 40 + col * 2: the cell (q, col) of this lane's env, q in [rlo, rlo + 10)
    auto cell_at = [&](int q, int col) -> int {                      // interior coordinates; outside the map: the frame
        const bool in = (unsigned)q < 20u && (unsigned)col < 20u;
        const int qq = min(max(q, rlo), rlo + 9), cc = min(max(col, 0), 19);
        const int v = nr ? 0 : (int)*(const int16_t*)(mine + qq * 40 + cc * 2);
        return in ? v : -1;
    };
    const int qa = s.r - 3, ca = s.c - 3;                            // the agent's cell, interior coordinates
    // ---- K3D::step by selects (the formulation of k_step3d / k_step3ds / Roll3D::tick), the same in the env's four lanes
    const int n0 = cell_at(qa, ca - 1), n1 = cell_at(qa, ca + 1), n2 = cell_at(qa + 1, ca), n3 = cell_at(qa - 1, ca);   // check_sur: left, right, "up" (row + 1), "down"
    const int c2 = cell_at(qa + 2 * dr, ca + 2 * dc), c3 = cell_at(qa + 3 * dr, ca + 3 * dc);
    Rule3D u;
    u.built = false; u.sel = false; u.done = false; u.newh = 0; u.reward0 = 0;
    if constexpr (!AUX) u = rules3d<DYN>(s, act, k, n0, n1, n2, n3, c2, c3, active, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
    const bool built = u.built;
    const int newh = u.newh;
    s.cross += (built && newh <= pl) ? 1 : 0;
    bool done = u.done;
    const int reward = u.sel ? reward_check3d(newh, pl) : u.reward0;
    done = done && active;
    if constexpr (!AUX) {
        s.ep_ret = clamp16(s.ep_ret + reward);
        s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    }
    if (active && part == 0) {                                       // the env's stores, by the first of its lanes
        if constexpr (!AUX) {
            if (a.reward) a.reward[env] = (float)reward;
            if (a.done) a.done[env] = done ? 1 : 0;
        }
        if (!AUX || nr) a.hdr[env] = s.pack();
        if (nr) a.episode[env] = episode;
        if (built && !nr) ((int16_t*)a.grid)[(size_t)env * GE + tcell] = (int16_t)newh;
        if (!AUX && a.stats_on && done) {                            // snac_step: episodic sums
            const double v = K::iou(nullptr, s, 0);
            stat_add(a.stat_episodes + env, 1);
            stat_add(a.stat_return + env, s.ep_ret);
            stat_add(a.stat_iou_fx + env, __double2ll_rn(v * FX40));
        }
    }
    // ---- the window round the NEW position: cells part, part + 4, ..., read before anything of the staging tile is written
    int cellv[13];
    if (a.obs) {
        const int q0 = s.r - 6, cl = s.c - 6;
#pragma unroll
        for (int j4 = 0; j4 < 13; ++j4) {
            const int el = min(part + 4 * j4, K::W - 1), i = el / 7, j = el - 7 * i;
            const int v = cell_at(q0 + i, cl + j);
            cellv[j4] = (built && q0 + i == tr && cl + j == tc) ? newh : v;   // the built cell shows (a build does not move)
        }
    }
    for (unsigned long long mk = __ballot(nr && part == 0); mk; mk &= mk - 1) {   // a reset env's record: empty, but for the cell it built
        const int L = __ffsll(mk) - 1, e = L >> 2;
        const int tp = __builtin_amdgcn_readlane(built ? tcell : -1, L), nh = __builtin_amdgcn_readlane(newh, L);
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
    asm volatile("" ::: "memory");                                   // (every read of the spans above, every write of the rows below)
    {
        OT* const S = (OT*)scr + le * K::D;
#pragma unroll
        for (int j4 = 0; j4 < 13; ++j4)
            if (part + 4 * j4 < K::W) S[part + 4 * j4] = (OT)cellv[j4];
        const double c0 = (double)s.cb, c1 = (double)s.cs;
        if (part == 1) S[K::W] = (OT)(DYN ? c0 / (double)s.tb : c0);
        if (part == 2) S[K::W + 1] = (OT)(DYN ? c1 / (double)a.total_step : c1);
    }
    // ---- the wave's rows leave: one run of nenv x 408 (204) bytes, 16 bytes per lane
    {
        const int validb = nenv * ROWB;                              // a multiple of 16: N % 4 = 0
        char* const g = (char*)a.obs + (size_t)env0 * ROWB;
        constexpr int NF = (E * ROWB + 1023) / 1024;
        uint4 fv[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) fv[i] = *(const uint4*)(scr + min(i * 1024 + lane * 16, WAVE_LDS - 16));
#pragma unroll
        for (int i = 0; i < NF; ++i)
            if (i * 1024 + lane * 16 < validb) store16<NTS>(g + i * 1024 + lane * 16, fv[i]);
    }
}

template <bool DYN, typename OT>
void launch_q(const KArgs& a, hipStream_t s) {
    const int tiles = (a.n + 15) / 16;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    using namespace snac_detail;
    int form = tune(TN_STEP3D_FORM);                                 // bit 0: non-temporal span loads, bit 1: non-temporal row stores; < 0: by batch size
    if (form < 0) form = a.n < tune(TN_STEP3D_NTLOAD_MIN) ? 2 : (a.n < tune(TN_STEP3D_HUGE_MIN) ? 1 : tune(TN_STEP3D_HUGE_FORM));
    switch (form & 3) {
        case 0: hipLaunchKernelGGL((k_step3dq<DYN, OT, 4, false, false>), grid, block, 0, s, a); break;
        case 1: hipLaunchKernelGGL((k_step3dq<DYN, OT, 4, true, false>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_step3dq<DYN, OT, 4, false, true>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((k_step3dq<DYN, OT, 4, true, true>), grid, block, 0, s, a); break;
    }
}

template <bool DYN, typename OT>
void launch_qa(const KArgs& a, hipStream_t s) {                     // masked reset / observe: plain span loads and rows
    const int tiles = (a.n + 15) / 16;
    hipLaunchKernelGGL((k_step3dq<DYN, OT, 4, false, false, true>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, a);
}

}  // namespace

namespace snac_detail {

void launch_aux3d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_qa<true, float>(a, s) : launch_qa<true, double>(a, s);
    else f32 ? launch_qa<false, float>(a, s) : launch_qa<false, double>(a, s);
}

void launch_step3dq(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_q<true, float>(a, s) : launch_q<true, double>(a, s);
    else f32 ? launch_q<false, float>(a, s) : launch_q<false, double>(a, s);
}

}  // namespace snac_detail
