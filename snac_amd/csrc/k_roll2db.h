// k_roll2db.h -- k_rollout2db: 2D rollouts by blocks of 64 / 128 / 256 envs (round 5; the kernel: instantiated by k_roll2db.hip for the canonical
// rows and by k_roll2dbv.hip for the layout variants without the plan tail)
#pragma once
#include "snac_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// 2D fused rollout, one BLOCK per 64 envs: the middle batches.  The lane-per-env kernels put a tick's stepping AND its 26 KB of rows into
// one wave (k_rollout2d: 2.5 us per tick and wave, the tile kernel's 32-env tiles 1.75 us), so a launch of N <= 65 536 envs -- at most one
// wave per SIMD -- takes 600 x that whatever N: 1.05 ms for 20 480 envs (5.0 GB of rows: 0.63 ms at the HBM write rate), 1.52 ms for 49 152
// on k_rollout2d; the time-parallel kernel k_rollout2dt is bound by instruction issue at 5.1-5.5 TB/s.  Here nine waves share 64 envs and
// split a tick by WORK, as k_rollout3db does:
//   wave 0, the stepper (lane = env): auto-reset, counter RNG, K2D::step on the bordered two-bit image of the 64 boards in LDS (the image
//       is the stepper's alone: no other wave reads or writes it), the plan bit from the lanes' plan rows in LDS (an env that starts over
//       on a new row has it fetched through the scalar cache, as in k_rollout2d), reward, done, the incremental boolean IoU; then the 7 row
//       words of the window round the new position, cut to its first column (14 bits = 7 two-bit cells each) and packed into FOUR dwords,
//       the two scalar slots (exact-reciprocal quotients), reward and done -> the tick's half of a small double buffer.  No global stores
//       (the record outputs of snac_rollout_rec excepted);
//   waves 1-8, the writers (8 envs each): behind the tick's barrier every lane assembles its 16-byte pieces of the wave's 8 rows -- value
//       g of the slice is element g % 51 of env g / 51, and WHERE that is in the publication never changes: a dword and a bit offset per
//       value, fixed per lane (one ds_read_b32 + v_bfe_i32 + a conversion per window cell; the pieces that hold a scalar slot in a store
//       instruction of their own) -- and stores them: 8 x 408 bytes as one run, 16 bytes per lane; the tick's reward / done runs by two of the writers.
// ONE barrier per tick: the stepper computes tick t + 1 into the other half while the writers write tick t; the barrier after that finds
// the writers done with the half the stepper takes next.
// Semantics are K2D::step's as k_rollout2d formulates them.  Conditions: every row written (SNAC_OBS_ALL / SNAC_OBS_TILED), the canonical
// layout or a variant without the plan tail (VAR below), N % 4 = 0 and a 16-byte aligned output; the dispatch table's SNAC_2D_BLOCK_* entries say for which N.

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte pieces of `rows` consecutive 51-value rows (VP values per piece) that hold element 49 or 50 of some row
constexpr int special_pieces(int rows, int VP) {
    int c = 0;
    for (int e = 0; e < rows; ++e) c += 1 + (((51 * e + 50) / VP != (51 * e + 49) / VP) ? 1 : 0);
    return c;
}

// NS: stepper waves per block (1: 64 envs, 2: 128 envs, 4: 256 envs for the batches just above 32 768 envs -- two steppers of ONE block sit on different SIMDs, the steppers of two
// co-resident blocks need not: with 257 .. 511 blocks of 64 envs the CUs that hold two of them decide the launch, and those run their
// two steppers' ~150 instructions a tick -- 64-bit shifts, the RNG's multiplies -- one after the other when they share a SIMD: 20 480 envs
// 0.94 ms on 320 blocks of 64 envs, as slow as the tile kernel).  Eight writer waves either way: 8 NS envs each.
// VAR: the layout variants of snac_env_desc without the plan tail (rows of a.ld = 51 .. 61 values: frame value 2, raw / normalised scalar
// slots, position and record tails -- the L-Net rows, rows that carry their own record).  The stepper also publishes the record values;
// a writer lane works out once per launch where each value of its pieces lives (a descriptor word per value: dword and bit offset of a
// window cell's code | an 8-byte slot of the publication | the kind), as k_rollout3db's variant form does.  (Before: the tile kernel, at
// 0.15-0.18 of the peak for 59-value rows between 6144 and 65 536 envs: 3.3 ms per 600 ticks at 16 384 envs.)
template <bool DYN, typename OT, bool EXPL, int NS, bool VAR>
__global__ __launch_bounds__((NS + 8) * 64) void k_rollout2db(const KArgs a) {
    using K = K2D<DYN, 64>;
    constexpr int D = K::D, ROWB = D * (int)sizeof(OT), GE = K::GE, RS = K::RS, BE = 64 * NS, ER = 8 * NS;   // envs per block / per writer
    constexpr int IMG_WORDS = 26 * RS * 2;
    __shared__ __attribute__((aligned(16))) uint32_t img_all[NS * IMG_WORDS]; // per stepper: the bordered two-bit image, 26 rows x 65 x 8 B (its alone)
    __shared__ uint32_t pl_all[NS * GE * 65];                                  // per stepper: the lanes' plan rows [20][65]
    // TB ticks per barrier: the stepper publishes TB ticks into one half of the buffers below, then the barrier; the writers take them in turn
    // (float32 rows, where a launch is mostly the tick's floor: 20 480 envs 0.509 -> 0.461 ms with TB = 4; float64 rows are bound by the HBM rate from
    // 16 384 envs and lost 3-6 % to the burstier stores: TB = 1)
    constexpr int TB = (sizeof(OT) == 4 && NS <= 2) ? 4 : 1;
    __shared__ __attribute__((aligned(16))) uint4 spw[2 * TB][BE];                  // window rows 0|1, 2|3, 4|5, 6 as 14-bit codes, two per dword
    // per parity: the two scalar slots of every env (8-byte slots 0 .. 2 BE - 1); VAR: then reward, done, row, column, count_brick,
    // count_step, total_brick, plan row of every env as int32 pairs (slots 2 BE + 4 env ..)
    __shared__ __attribute__((aligned(16))) double ssc[2 * TB][VAR ? 6 * BE : 2 * BE];
    __shared__ float srew[2 * TB][BE];
    __shared__ __attribute__((aligned(16))) uint8_t sdone[2 * TB][BE];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chunk = ((int)gridDim.x + 7) >> 3;                     // an XCD takes a contiguous eighth of the envs
    const int blk = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
    const int benv0 = blk * BE;                                      // the block's first env
    if (benv0 >= a.n) return;                                        // the whole block
    const int bnenv = min(BE, a.n - benv0);
    if (wv < NS) {
        // ================================ the steppers: one env per lane ================================
        const int env0 = benv0 + 64 * wv, nenv = min(max(bnenv - 64 * wv, 0), 64), pub = 64 * wv;   // this stepper's envs; where it publishes
        uint32_t* const img = img_all + wv * IMG_WORDS;
        uint32_t* const pl = pl_all + wv * (GE * 65);
        const bool active = lane < nenv;
        const int env = min(env0 + (active ? lane : 0), a.n - 1);
        uint64_t* const cells = K::cells(img);
        Lane s;
        s.clear();
        s.r = 3; s.c = 3;                                            // idle lanes keep an in-range position and plan row 0
        int episode = 0;
        if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
        K::load_grid(img, a, env0, nenv, lane);
        const uint64_t gid = (uint64_t)(a.env_id_base + env);
        const EnvKeys sk = env_keys(a.key_step, gid), pk = env_keys(a.key_plan, gid);
        int pcnt = 0, gcnt = 0, inter = 0;                           // |P|, |G|, |P and G| of the lane's env as the launch finds them
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
        auto inputs_of = [&](int t, int& aa, int& kk) {              // counter RNG of tick t
            const uint32_t w = rng_word(sk, a.t0 + (uint32_t)t);
            aa = (int)(((w >> 16) * (uint32_t)K::A) >> 16); kk = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
        };
        auto load_inputs = [&](int t, int& aa, int& kk) {            // EXPL: the caller's bytes of tick t over the counter-RNG values
            inputs_of(t, aa, kk);
            if (t < a.T) {
                const size_t at = (size_t)t * (size_t)a.n + (size_t)env;
                if (a.actions) aa = (int)a.actions[at];
                if (a.step_size) kk = (int)a.step_size[at];
            }
        };
        int act = 0, k = 1, act_n = 0, k_n = 1;
        if constexpr (EXPL) { load_inputs(0, act, k); load_inputs(1, act_n, k_n); }
        else inputs_of(0, act, k);
        // The tick's chain of dependent steps is what a launch of <= 256 blocks costs (600 x the tick), so what the NEXT tick needs first is
        // asked for while this tick's window rows are on their way: the agent's row word and plan word at the new position (an env that
        // starts over reads them again, below) and the counter-RNG word.
        uint64_t w_pf = cells[s.r * RS + lane];
        uint32_t pl_pf = pl[(s.r - 3) * 65 + lane];
        for (int t = 0; t < a.T; ++t) {
            const int par = ((t / TB) & 1) * TB + t % TB;           // which of the 2 x TB tick buffers
            const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
            if (__builtin_expect(__any(nr), 0)) {                    // rare, out of line
                bool fresh = false;                                  // a new plan row (K2D::reset: it brings its total_brick; the same row keeps the header's)
                if (nr) {
                    const int old_pidx = s.pidx;
                    episode += 1;
                    const int pidx = pick_plan<K>(a, pk, episode, old_pidx);
                    if (pidx != old_pidx) { fresh = true; s.pidx = pidx; }
                    s.r = 3; s.c = 3; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.cross = 0; s.flags = 0;
                    gcnt = 0; inter = 0;
                }
                for (unsigned long long m = __ballot(nr); m; m &= m - 1) K::clear(img, __ffsll(m) - 1, lane);
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
                w_pf = cells[s.r * RS + lane];                       // (the board was cleared, the plan row may be another)
                pl_pf = pl[(s.r - 3) * 65 + lane];
            }
            // ---- K2D::step (DMP_Env_2D_dynamic_usedata_plan.py:85-147)
            k = min(max(k, 1), 3);
            uint64_t* const cw = cells + s.r * RS + lane;
            const uint64_t w = w_pf;
            const int off = 2 * s.c;
            const bool was = ((w >> off) & 1ull) != 0ull;
            const bool planned = ((pl_pf >> (s.c - 3)) & 1u) != 0u;
            const bool first = s.cs == 0;
            const Rule2D u = rules2d(s, act, k, was, planned, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
            if (u.drop) {
                if (active) *cw = w | (1ull << off);
                gcnt += was ? 0 : 1;                                 // the running counts of the boolean IoU
                inter += (!was && planned) ? 1 : 0;
            }
            const bool done = active && u.done;
            const int reward = u.reward;
            s.ep_ret = clamp16(s.ep_ret + reward);
            s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
            if (active) {
                const size_t row = (size_t)t * (size_t)a.n + (size_t)env;
                if (a.actions_out) a.actions_out[row] = (int8_t)act;
                if (a.step_size_out) a.step_size_out[row] = (int8_t)k;
                if (a.plan_idx_out) a.plan_idx_out[row] = (int16_t)s.pidx;
                if (a.first_out) a.first_out[row] = first ? 1 : 0;
            }
            if (__builtin_expect(__any(done), 0)) {                  // boolean IoU of the finished episode
                if (done) {
                    const double v = (double)inter / (double)(pcnt + gcnt - inter);
                    d_eps += 1; d_ret += s.ep_ret; d_iou += __double2ll_rn(v * FX40);
                }
            }
            // ---- the tick's outputs -> its half of the double buffer
            {
                const uint64_t* const wp = cells + (s.r - 3) * RS + lane;
                const int sh = 2 * (s.c - 3);
                uint64_t wq[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) wq[i] = wp[i * RS];
                w_pf = cells[s.r * RS + lane];                       // the next tick's first reads and its RNG word, in the shadow of the seven above
                pl_pf = pl[(s.r - 3) * 65 + lane];
                if constexpr (!EXPL) inputs_of(t + 1, act, k);
                uint32_t wr[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) wr[i] = (uint32_t)(wq[i] >> sh) & 0x3FFFu;
                spw[par][pub + lane] = make_uint4(wr[0] | (wr[1] << 14), wr[2] | (wr[3] << 14), wr[4] | (wr[5] << 14), wr[6]);
                double v0 = (double)s.cb, v1 = (double)s.cs;
                if (VAR ? (a.sc_norm != 0) : DYN) {                  // cb / tb, cs / T: correctly rounded (Roll3D, tests/native/recip_check.c)
                    const double c0 = v0, c1 = v1, q0 = c0 * rtb, q1 = c1 * rT;
                    v0 = __builtin_fma(__builtin_fma(-q0, dtb, c0), rtb, q0);
                    v1 = __builtin_fma(__builtin_fma(-q1, dT, c1), rT, q1);
                    if (__builtin_expect(__any(active && s.tb <= 0), 0)) {   // only a hand-made header; the asm keeps it a branch
                        asm volatile("" ::: "memory");
                        v0 = c0 / dtb;
                    }
                }
                double2 sv; sv.x = v0; sv.y = v1;
                *(double2*)&ssc[par][2 * (pub + lane)] = sv;
                if constexpr (VAR) {
                    int4* const rec = (int4*)&ssc[par][2 * BE + 4 * (pub + lane)];
                    rec[0] = make_int4(reward, done ? 1 : 0, s.r, s.c);
                    rec[1] = make_int4(s.cb, s.cs, s.tb, s.pidx);
                }
                srew[par][pub + lane] = (float)reward;
                sdone[par][pub + lane] = done ? 1 : 0;
            }
            if constexpr (EXPL) {                                    // the next tick's bytes are here; ask for those of the tick after it
                act = act_n; k = k_n;
                load_inputs(t + 2, act_n, k_n);
            }
            if (t % TB == TB - 1 || t == a.T - 1) lds_barrier();     // this group of ticks is published; the writers are done with the group before
        }
        K::store_grid(img, a, env0, nenv, lane);
        if (active) {
            a.hdr[env] = s.pack();
            a.episode[env] = episode;
            if (d_eps) {
                a.stat_episodes[env] += d_eps;
                a.stat_return[env] += d_ret;
                a.stat_iou_fx[env] += d_iou;
            }
        }
        return;
    }
    // ================================ the writers: 8 NS envs per wave ================================
    const int wr = wv - NS, e0 = wr * ER;                            // writer 0 .. 7; its first env within the block
    const int rows = min(max(bnenv - e0, 0), ER);                    // rows of this wave that exist
    const bool tl = a.obs_mode == SNAC_OBS_TILED;
    const int envw = benv0 + e0;                                     // (a multiple of 8 NS: the wave's rows lie in one 64-env tile)
    const int LD = VAR ? a.ld : D, ROWBV = LD * (int)sizeof(OT);     // values / bytes per row
    char* const obs0 = (char*)a.obs + (tl ? ((size_t)(envw >> 6) * (size_t)a.tiled_T + (size_t)a.tiled_t0) * 64 + (size_t)(envw & 63) : (size_t)envw) * ROWBV;
    const size_t tstride = (tl ? (size_t)64 : (size_t)a.n) * ROWBV;
    auto reward_done = [&](int t, int par) __attribute__((always_inline)) {   // the tick's reward / done runs: writers 0 .. NS - 1 the rewards of 64 envs each, writers NS .. 2 NS - 1 the done flags
        if (wr < 2 * NS) {
            const int c0 = 64 * (wr % NS), cn = min(max(bnenv - c0, 0), 64);
            const size_t row = (size_t)t * (size_t)a.n + (size_t)(benv0 + c0);
            if (wr < NS) {
                if (a.reward && lane < cn) a.reward[row + lane] = srew[par][c0 + lane];
            } else if (a.done) {
                if (cn == 64 && ((((uintptr_t)a.done) | (uintptr_t)a.n) & 3) == 0) {
                    if (lane < 16) ((uint32_t*)(a.done + row))[lane] = ((const uint32_t*)(sdone[par] + c0))[lane];
                } else if (lane < cn) a.done[row + lane] = sdone[par][c0 + lane];
            }
        }
    };
    if constexpr (VAR) {
        // one descriptor per value of this lane's pieces (piece lane + 64 q, value u): bits 0-11 the byte offset of the dword that holds a window
        // cell's code in the tick's spw half, 12-16 the code's bit offset, 17-26 an 8-byte slot of the tick's ssc half, 30-31 the kind:
        // 0 cell, 1 float64 slot, 2 / 3 low / high int32 of a slot.  widemask: bit q = some lane's piece of iteration q holds a slot value.
        constexpr int VP = 16 / (int)sizeof(OT), NQV = (ER * 61 * (int)sizeof(OT) / 16 + 63) / 64;   // iterations for the longest row (61 values)
        uint32_t desc[NQV * VP];
        uint32_t widemask = 0;
        {
            const int pos_n = (a.tail & SNAC_TAIL_POSITION) ? 2 : 0;
            const float rLD = 1.0f / (float)LD;
            const int nvals = max(rows * LD, 1);
#pragma unroll
            for (int q = 0; q < NQV; ++q) {
                bool wide = false;
#pragma unroll
                for (int u = 0; u < VP; ++u) {
                    const int gi = min((lane + 64 * q) * VP + u, nvals - 1);
                    const int e = (int)(((float)gi + 0.5f) * rLD), x = gi - e * LD, ea = e0 + e;   // exact: gi < 1000, the product is off by < 1e-5
                    const int xc = min(x, 48), i = (xc * 37) >> 8, j = xc - 7 * i;                // x / 7, x % 7 for x < 49
                    const int ti = x - D;
                    const bool is_win = x < 49, is_sc = (unsigned)(x - 49) < 2u;
                    const int jr = min(max(ti < pos_n ? 2 + ti : ti - pos_n, 0), 7);               // which record value, for a value past the scalar slots
                    const int off = ea * 16 + (i >> 1) * 4, sh = (i & 1) * 14 + 2 * j;
                    const int slot = is_sc ? ea * 2 + (x - 49) : 2 * BE + ea * 4 + (jr >> 1);
                    const int kind = is_win ? 0 : (is_sc ? 1 : 2 + (jr & 1));
                    desc[q * VP + u] = (uint32_t)off | ((uint32_t)sh << 12) | ((uint32_t)slot << 17) | ((uint32_t)kind << 30);
                    wide = wide || kind != 0;
                }
                widemask |= __any(wide) ? 1u << q : 0u;
            }
        }
        const int npieces = rows * ROWBV / 16;
        const int fv = a.frame_val;
        for (int t = 0; t < a.T; ++t) {
            const int par = ((t / TB) & 1) * TB + t % TB;           // which of the 2 x TB tick buffers
            if (t % TB == 0) lds_barrier();
            char* const g = obs0 + (size_t)t * tstride;
            const char* const wb = (const char*)spw[par];
            const char* const sb = (const char*)ssc[par];
#pragma unroll
            for (int q = 0; q < NQV; ++q) {
                if (q * 64 >= npieces) break;                            // (wave-uniform)
                int cv[VP];
#pragma unroll
                for (int u = 0; u < VP; ++u) {
                    const uint32_t d = desc[q * VP + u];
                    const int c = (int)__builtin_amdgcn_sbfe((int)*(const uint32_t*)(wb + (d & 0xfffu)), (d >> 12) & 31u, 2);   // 0 / 1 / -1 (frame)
                    cv[u] = c < 0 ? fv : c;
                }
                OT val[VP];
                if ((widemask >> q) & 1u) {
                    uint2 w2[VP];
#pragma unroll
                    for (int u = 0; u < VP; ++u) w2[u] = *(const uint2*)(sb + ((desc[q * VP + u] >> 14) & 0x1ff8u));
#pragma unroll
                    for (int u = 0; u < VP; ++u) {                       // by masks, not selects (the compiler turns selects round a conversion into branches)
                        const uint32_t kd = desc[q * VP + u] >> 30;
                        int iv = cv[u];
                        iv = kd == 2 ? (int)w2[u].x : iv;
                        iv = kd == 3 ? (int)w2[u].y : iv;
                        const uint32_t m = (uint32_t)-(int)(kd == 1);
                        if constexpr (VP == 2) {
                            const uint64_t b = (uint64_t)__double_as_longlong((double)iv);
                            const uint32_t lo = (m & w2[u].x) | (~m & (uint32_t)b), hi = (m & w2[u].y) | (~m & (uint32_t)(b >> 32));
                            val[u] = __hiloint2double((int)hi, (int)lo);
                        } else {
                            const float fs = (float)__hiloint2double((int)w2[u].y, (int)w2[u].x);
                            val[u] = __uint_as_float((m & __float_as_uint(fs)) | (~m & __float_as_uint((float)iv)));
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < VP; ++u) val[u] = (OT)cv[u];
                }
                const int pc = lane + 64 * q;
                if (pc < npieces) {
                    if constexpr (VP == 2) { double2 o; o.x = val[0]; o.y = val[1]; *(double2*)(g + (uint32_t)pc * 16u) = o; }
                    else { float4 o; o.x = val[0]; o.y = val[1]; o.z = val[2]; o.w = val[3]; *(float4*)(g + (uint32_t)pc * 16u) = o; }
                }
            }
            reward_done(t, par);
        }
        return;
    }
    // The wave's 8 rows leave as NP 16-byte pieces of VP values.  Where each value comes from does not change from tick to tick: a window
    // cell = two bits of a dword of the tick's spw half (element x < 49 of env e: row x / 7, column x % 7), or one of the wave's 16 scalar
    // slots.  The pieces that hold a scalar slot (12 of a wave's 204 with float64 rows, 8-16 of 102 with float32) are the LAST pieces a
    // lane takes: the first lanes in a store instruction of their own; the pure-cell pieces before them, in order, need no selects at
    // all -- one ds_read_b32, one v_bfe_i32 and one conversion per value (with the selects in every piece a writer's tick was ~100
    // instructions, 40 of them selects, and a CU stepped one block per 0.74 us whatever else was resident: profiles/r05_2d_block.txt).
    constexpr int VP = 16 / (int)sizeof(OT), NP = ER * ROWB / 16, NSMAX = 2 * ER, NQP = (NP - special_pieces(ER, VP) + 63) / 64;
    static_assert(special_pieces(ER, VP) <= 64, "the pieces with scalar slots leave in one store instruction");
    auto specials_below = [&](int P) -> int {                        // how many special pieces have an index < P
        int c = 0;
        for (int e = 0; e < ER; ++e) {
            const int pa = (51 * e + 49) / VP, pb = (51 * e + 50) / VP;
            c += (pa < P ? 1 : 0) + ((pb != pa && pb < P) ? 1 : 0);
        }
        return c;
    };
    const int NSP = specials_below(NP);                              // (the same for every lane)
    int ppiece[NQP];                                                 // the lane's pure pieces (>= NP: none)
#pragma unroll
    for (int q = 0; q < NQP; ++q) {
        const int kth = lane + 64 * q;                               // the kth pure piece: p with p - specials_below(p + 1) == kth, p not special
        int pp = kth;
        for (int it = 0; it < NSMAX + 1; ++it) pp = kth + specials_below(pp + 1);
        ppiece[q] = kth < NP - NSP ? pp : NP;
    }
    int spiece = NP;                                                 // the lane's special piece: the lane-th one
    {
        int c = 0;
        for (int e = 0; e < ER; ++e) {
            const int pa = (51 * e + 49) / VP, pb = (51 * e + 50) / VP;
            if (c == lane) spiece = pa;
            c += 1;
            if (pb != pa) { if (c == lane) spiece = pb; c += 1; }
        }
    }
    int woff[NQP * VP], wsh[NQP * VP];                               // pure pieces: dword and bit offset of each value's two-bit code
#pragma unroll
    for (int q = 0; q < NQP; ++q)
#pragma unroll
        for (int u = 0; u < VP; ++u) {
            const int g = min(ppiece[q], NP - 1) * VP + u, e = g / 51, x = min(g - 51 * e, 48), i = x / 7, j = x - 7 * i;
            woff[q * VP + u] = (e0 + e) * 16 + (i >> 1) * 4;
            wsh[q * VP + u] = (i & 1) * 14 + 2 * j;
        }
    int soff[VP], ssh[VP], ssrc[VP];                                 // the special piece: the same, and the scalar slot (-1: a cell)
#pragma unroll
    for (int u = 0; u < VP; ++u) {
        const int g = min(spiece, NP - 1) * VP + u, e = g / 51, x = g - 51 * e, xc = min(x, 48), i = xc / 7, j = xc - 7 * i;
        soff[u] = (e0 + e) * 16 + (i >> 1) * 4;
        ssh[u] = (i & 1) * 14 + 2 * j;
        ssrc[u] = x >= 49 ? ((e0 + e) * 2 + (x - 49)) * 8 : -1;
    }
    const int npieces = rows * ROWB / 16;
    for (int t = 0; t < a.T; ++t) {
        const int par = ((t / TB) & 1) * TB + t % TB;
        if (t % TB == 0) lds_barrier();
        char* const g = obs0 + (size_t)t * tstride;
        const char* const wb = (const char*)spw[par];
        const char* const scs = (const char*)ssc[par];
        uint32_t ww[NQP * VP], sw[VP];
        double sd[VP];
#pragma unroll
        for (int i = 0; i < NQP * VP; ++i) ww[i] = *(const uint32_t*)(wb + woff[i]);
#pragma unroll
        for (int u = 0; u < VP; ++u) { sw[u] = *(const uint32_t*)(wb + soff[u]); sd[u] = *(const double*)(scs + max(ssrc[u], 0)); }
#pragma unroll
        for (int q = 0; q < NQP; ++q) {
            OT v[VP];
#pragma unroll
            for (int u = 0; u < VP; ++u) v[u] = (OT)(int)__builtin_amdgcn_sbfe((int)ww[q * VP + u], wsh[q * VP + u], 2);   // signed 2-bit field: 0 / 1 / -1 (frame)
            if (ppiece[q] < npieces) {
                if constexpr (VP == 2) { double2 o; o.x = v[0]; o.y = v[1]; *(double2*)(g + ppiece[q] * 16) = o; }
                else { float4 o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3]; *(float4*)(g + ppiece[q] * 16) = o; }
            }
        }
        {
            OT v[VP];
#pragma unroll
            for (int u = 0; u < VP; ++u) {
                const OT c = (OT)(int)__builtin_amdgcn_sbfe((int)sw[u], ssh[u], 2);
                v[u] = ssrc[u] >= 0 ? (OT)sd[u] : c;
            }
            if (spiece < npieces) {
                if constexpr (VP == 2) { double2 o; o.x = v[0]; o.y = v[1]; *(double2*)(g + spiece * 16) = o; }
                else { float4 o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3]; *(float4*)(g + spiece * 16) = o; }
            }
        }
        reward_done(t, par);
    }
}

template <bool DYN, typename OT, int NS, bool VAR>
void launch_roll2db_w(const KArgs& a, hipStream_t s) {
    const int blocks = (a.n + 64 * NS - 1) / (64 * NS);
    const dim3 grid((unsigned)(((blocks + 7) / 8) * 8)), block((NS + 8) * 64);   // a multiple of 8: the XCD remap covers every block
    if (a.actions || a.step_size) hipLaunchKernelGGL((k_rollout2db<DYN, OT, true, NS, VAR>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_rollout2db<DYN, OT, false, NS, VAR>), grid, block, 0, s, a);
}

template <bool DYN, typename OT, bool VAR>
void launch_roll2db_n(const KArgs& a, int ns, hipStream_t s) {
    if constexpr (!VAR) {
        if (ns == 4) { launch_roll2db_w<DYN, OT, 4, VAR>(a, s); return; }
    }
    if (ns >= 2) launch_roll2db_w<DYN, OT, 2, VAR>(a, s);
    else launch_roll2db_w<DYN, OT, 1, VAR>(a, s);
}

}  // namespace
