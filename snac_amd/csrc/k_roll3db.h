// k_roll3db.h -- k_rollout3db: 3D rollouts by blocks of 64 envs (the kernel; instantiated by k_roll3db.hip for the canonical rows and by
// k_roll3dbv.hip for the layout variants)
#pragma once
#include "snac_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// 3D fused rollout, one BLOCK per 64 envs (round 3).  k_rollout3d is bound by its tile: 8 envs per wave keep 8 of 64 lanes busy
// in the transition (39 instructions per env-step, profiles/r03_3d_summary.txt), and sixteen per wave lose what they save to
// latency nothing hides (profiles/r03_3d_wide_experiment.txt).  Here nine waves share 64 envs and split a tick by WORK:
//   wave 0, the stepper (lane = env): auto-reset, counter RNG, K3D::step by selects on the 64 bordered height maps in LDS -- one env
//       per lane, no redundant lanes --, reward, done; it publishes position, scalar slots, reward, done, the one cell the tick
//       built (and what the episodic sums need) into the tick's half of a small double buffer.  It issues NO stores, so its one
//       vector-memory wait per tick -- the plan cell of the build target, loaded at the end of the tick before (the next action is a
//       counter-RNG word or a byte loaded two ticks ahead, the next position is known, a pending reset is applied to the scalars
//       early) -- waits for loads only.  It never WRITES a map either;
//   waves 1-8, the writers (8 envs each, lane = (env, window row 0 .. 6 or the two scalar slots)) own the maps of their envs: behind
//       the tick's barrier they bring them up to date (the map of an env that started over is cleared, the built cell written),
//       gather the 7x7 window round the published position into the wave's slice of a staging tile of int16 cells (one aligned
//       16-byte write per lane and window row), read it back in store order, convert on the way out and write the 8 rows as one run
//       of 8 x 408 bytes, 16 bytes per lane (a wave's own LDS operations are ordered: no further barrier); the tick's reward / done
//       runs; IoU and sums of episodes that ended.
// ONE barrier per tick: the stepper computes tick t + 1 while the writers apply, gather and write tick t.  So the maps the stepper
// reads lag by one tick: it patches the cell it built a tick ago into what it reads, and takes the cells of an env that started
// over (now, or a tick ago: its map may not be cleared yet) from their coordinates -- an empty map is 0 inside, -1 on the frame.
// What a writer applies at tick t was published before barrier t; the stepper reads the maps for tick t + 2 behind barrier t + 1,
// which the writers reach after they are done with tick t.  tick = max(stepper, slowest writer) + one barrier: ~2500 cycles at
// N = 16 384 (stepper 1940 -- 1000 when it runs alone --, barrier 460), 1.06 ms per 1000 ticks against 1.27 for k_rollout3d
// (float32 rows 0.98 against 1.25); what was tried on the way (two barriers with the stepper writing the maps, four writers of
// 16 envs, a scratch-spilled flush, idle waves on the stepper's SIMD, wave priorities, the staging tile as float64 / misaligned
// int16 / none) is in profiles/r03_3d_block_kernel.txt.
// Semantics are K3D::step's, formulated as in k_step3d / Roll3D::tick.  Conditions: every row written (SNAC_OBS_ALL /
// SNAC_OBS_TILED), canonical layout, <= TB_MAX plans, N % 4 = 0 and a 16-byte aligned output, N >= 6144 (float32 rows: 4096; below,
// k_rollout3d's one-wave blocks are faster); the rest stays on k_rollout3d.

// A barrier between waves that exchange data through LDS only (no wait for the writers' global stores).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// VAR (round 5): the layout variants of snac_env_desc (rows of a.ld = 51 + tail values, raw or normalised scalar slots; 3D has no frame
// value 2).  The stepper also publishes the record values (reward, done, position, counters, plan row: srec), the plan rows of the
// block's envs live in LDS (plw; rtab gives its 16 KB up, the stepper divides on a reset instead), and a writer assembles a tick's 8 rows
// 16 bytes per lane from wherever each value lives: value g of the wave's slice is element g % LD of env g / LD -- a window cell, a
// scalar slot, a position / record value or a plan cell.  WHERE does not change from tick to tick: every lane works it out once per
// launch as one descriptor word per value (byte offset of a 2-byte cell | 8-byte slot | kind), and a tick is an and, a 2-byte LDS read and
// a conversion per value (the few scalar slots / record values: an 8-byte read in the iterations that hold one).
// An env that starts an episode on another plan row needs that row in LDS before the writers assemble the tick.  The writers cannot
// fetch it: vmcnt retires in order, so a load behind a tick's row stores waits for all of them (17 instead of 8.4 us per tick of 451-value
// rows when tried).  The stepper issues no stores: at the start of a tick it loads the rows of the envs that change (lanes 0 .. 49, 16
// bytes each, PS envs in flight), writes them into plw behind the tick's barrier -- the writers are done with the tick before -- and a
// second barrier releases the writers' assembly (rows with the plan tail only; their tick is 8 us of stores).
// MODE 0: the canonical rows; 1: a layout variant without the plan tail (51 .. 61 values); 2: with it (451 .. 461 values)
template <bool DYN, typename OT, bool EXPL, int MODE>
__global__ __launch_bounds__(576) void k_rollout3db(const KArgs a) {
    constexpr bool VAR = MODE != 0, PT = MODE == 2;
    using K = K3D<DYN, 64>;
    constexpr int D = K::D, ROWB = D * (int)sizeof(OT), GE = K::GE, NT = 576;
    __shared__ __attribute__((aligned(16))) uint32_t hm[64 * K::ES / 2];      // 64 bordered height maps, 1356 bytes apart (odd dword stride)
    __shared__ __attribute__((aligned(16))) int16_t cells[PT ? 64 * 56 + 64 * GE : 64 * 56];
    auto& stg16 = cells;                                             // the tick's 64 windows as int16 cells: [env][window row][8], a row = one 16-byte write
    constexpr int PLW = 64 * 56;                                     // VAR: cells + PLW = the plan rows of the block's envs, [env][400] (one address space with the windows)
    __shared__ double rtab[PT ? 1 : TB_MAX];                        // 1 / total_brick per plan row: no division in the stepper
    __shared__ int16_t tbtab[TB_MAX];
    // per parity: the two scalar slots of every env (8-byte slots 0 .. 127), VAR: then reward, done, row, column, count_brick, count_step,
    // total_brick, plan row of every env as int32 pairs (slots 128 + 4 env ..)
    constexpr int WSLOTS = VAR ? 128 + 256 : 128;
    __shared__ __attribute__((aligned(16))) double ssc[2][WSLOTS];
    constexpr int PLAN_PIECES = GE * 2 / 16, PS = 8;                 // 16-byte pieces of a plan row; rows the stepper holds in flight
    __shared__ int4 sfin[2][64];                                     // (sum of min(height, plan), tb + cb - sum, episode return) of an env that finished
    __shared__ int2 spub[2][64];                                     // x: row | col << 8 | started over << 16;  y: built cell index | height << 16, or -1
    __shared__ float srew[2][64];
    __shared__ __attribute__((aligned(16))) uint8_t sdone[2][64];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chunk = ((int)gridDim.x + 7) >> 3;                     // an XCD takes a contiguous eighth of the envs (as k_rollout3d)
    const int blk = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
    const int env0 = blk * 64;
    if (env0 >= a.n) return;                                         // the whole block
    const int nenv = min(64, a.n - env0);
    for (int i = tid; i < a.num_plans; i += NT) {
        const int tb = a.plan_tb[i];
        tbtab[i] = (int16_t)tb;
        if constexpr (!PT) rtab[i] = 1.0 / (double)tb;
    }
    {   // records -> LDS: everything frame, then the interiors (idle lanes keep all-frame maps: their steps change nothing)
        for (int i = tid; i < 64 * K::ES / 2; i += NT) hm[i] = 0xFFFFFFFFu;
        __syncthreads();
        const int16_t* src = (const int16_t*)a.grid + (size_t)env0 * GE;
        int16_t* h = K::hmap(hm);
        for (int i = tid; i < nenv * GE; i += NT) {
            const int e = i / GE, cell = i - e * GE, r = cell / 20, c = cell - r * 20;
            h[e * K::ES + (r + 3) * 26 + c + 3] = src[i];
        }
    }
    __syncthreads();
    constexpr bool plan_tail = PT;
    auto plan_rows_in = [&]() __attribute__((always_inline)) {                                      // VAR, before tick 0: every env's plan row -> plw (all nine waves)
        for (int i = tid; i < 64 * PLAN_PIECES; i += NT) {
            const int e = i / PLAN_PIECES, pc = i - e * PLAN_PIECES;
            const int pid = ((const int*)&ssc[0][128 + 4 * e])[7];
            ((uint4*)(cells + PLW + e * GE))[pc] = ((const uint4*)((const int16_t*)a.plans + (size_t)pid * GE))[pc];
        }
    };
    if (wv == 0) {
        // ================================ the stepper: one env per lane ================================
        const bool active = lane < nenv;
        const int env = env0 + (active ? lane : 0);
        const int16_t* const hmine = K::hmap(hm) + lane * K::ES;
        Lane s;
        s.clear();
        s.r = 3; s.c = 3;
        int episode = 0;
        if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
        const uint64_t gid = (uint64_t)(a.env_id_base + env);
        const EnvKeys sk = env_keys(a.key_step, gid), pk = env_keys(a.key_plan, gid);
        double dtb = (double)s.tb, rtb = 1.0 / dtb;
        const double dT = (double)a.total_step, rT = 1.0 / dT;
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
        bool chg = false;                                            // VAR: the env starts the coming tick on another plan row
        auto reset_scalars = [&]() {                                 // K3D::reset without the map
            episode += 1;
            const int np = pick_plan<K>(a, pk, episode, s.pidx);
            if (np != s.pidx) {                                      // K::reset: a new row brings its total_brick, the same row keeps the header's
                chg = true;
                s.pidx = np; s.tb = tbtab[np];
                dtb = (double)s.tb;
                if constexpr (PT) rtb = 1.0 / dtb; else rtb = rtab[np];
            }
            s.r = 3; s.c = 3; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.cross = 0; s.flags = 0;
        };
        auto target_cell = [&](int aa) -> int {                      // the build target of action aa from the current position, plan coordinates
            const int d = aa & 3;
            const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
            const int tr = s.r + dr - 3, tc = s.c + dc - 3;
            return ((unsigned)tr < 20u && (unsigned)tc < 20u) ? tr * 20 + tc : 0;
        };
        bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);   // starts a new episode with the coming tick
        if (nr) reset_scalars();
        uint4 R0, R1, R2, R3, R4, R5, R6, R7;                        // VAR: plan rows on their way to plw (lane = 16-byte piece; PS = 8 named registers:
        R0 = R1 = R2 = R3 = R4 = R5 = R6 = R7 = make_uint4(0u, 0u, 0u, 0u);   // as an array they end up in scratch memory)
        auto issue1 = [&](unsigned long long& m, int pidv, uint4& r) __attribute__((always_inline)) {
            if (m) {
                const int e = __ffsll(m) - 1;
                m &= m - 1;
                const int pid = __builtin_amdgcn_readlane(pidv, e);
                r = ((const uint4*)((const int16_t*)a.plans + (size_t)pid * GE))[min(lane, PLAN_PIECES - 1)];
            }
        };
        auto write1 = [&](unsigned long long& m, const uint4& r) __attribute__((always_inline)) {
            if (m) {
                const int e = __ffsll(m) - 1;
                m &= m - 1;
                if (lane < PLAN_PIECES) ((uint4*)(cells + PLW + e * GE))[lane] = r;
            }
        };
        auto rows_issue = [&](unsigned long long& m, int pidv) __attribute__((always_inline)) {   // loads of the rows of the first PS envs of m; m loses them
            issue1(m, pidv, R0); issue1(m, pidv, R1); issue1(m, pidv, R2); issue1(m, pidv, R3);
            issue1(m, pidv, R4); issue1(m, pidv, R5); issue1(m, pidv, R6); issue1(m, pidv, R7);
        };
        auto rows_write = [&](unsigned long long m) __attribute__((always_inline)) {               // -> the rows of the first PS envs of m
            write1(m, R0); write1(m, R1); write1(m, R2); write1(m, R3); write1(m, R4); write1(m, R5); write1(m, R6); write1(m, R7);
        };
        static_assert(PS == 8, "eight rows in flight");
        if constexpr (VAR) {
            if (plan_tail) {                                         // the rows of tick 0, by all waves
                ((int*)&ssc[0][128 + 4 * lane])[7] = s.pidx;
                __syncthreads();
                plan_rows_in();
                __syncthreads();
            }
            chg = false;
        }
        bool nr_prev = false;                                        // started one a tick ago: the map may not be cleared yet
        int pb_idx = -1, pb_h = 0;                                   // the cell built a tick ago: may not be in the map yet
        int act = 0, k = 1, act_n = 0, k_n = 1;
        if constexpr (EXPL) { load_inputs(0, act, k); load_inputs(1, act_n, k_n); }
        else inputs_of(0, act, k);
        int pl = (int)((const int16_t*)a.plans)[(size_t)s.pidx * GE + target_cell(act)];
        for (int t = 0; t < a.T; ++t) {
            const int par = t & 1;
            unsigned long long pm = 0, pm_rest = 0;                      // VAR: envs that start this tick on another plan row / those not loaded yet
            const int pid_t = s.pidx;
            if constexpr (VAR) {
                if (plan_tail) {
                    pm = __ballot(chg);
                    pm_rest = pm;
                    chg = false;
                    if (pm) rows_issue(pm_rest, pid_t);
                }
            }
            k = min(max(k, 1), 3);
            const int hidx = s.r * 26 + s.c;                             // the agent's cell
            const int d = act & 3;
            const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
            const int dl = dr * 26 + dc;
            const int16_t* const h = hmine + hidx;
            int n0 = h[-1], n1 = h[1], n2 = h[26], n3 = h[-26];          // check_sur: left, right, "up" (row + 1), "down"
            int c2 = h[2 * dl], c3 = h[3 * dl];
            if (__any(nr || nr_prev)) {                                  // cells of an empty map by their coordinates: 0 inside, -1 on the frame
                asm volatile("" ::: "memory");
                if (nr || nr_prev) {
                    auto at = [&](int rr, int cc) { return ((unsigned)(rr - 3) < 20u && (unsigned)(cc - 3) < 20u) ? 0 : -1; };
                    n0 = at(s.r, s.c - 1); n1 = at(s.r, s.c + 1); n2 = at(s.r + 1, s.c); n3 = at(s.r - 1, s.c);
                    c2 = at(s.r + 2 * dr, s.c + 2 * dc); c3 = at(s.r + 3 * dr, s.c + 3 * dc);
                }
            }
            if (!nr && pb_idx >= 0) {                                    // the cell built a tick ago (this episode's)
                const int o = pb_idx - hidx;
                n0 = o == -1 ? pb_h : n0; n1 = o == 1 ? pb_h : n1; n2 = o == 26 ? pb_h : n2; n3 = o == -26 ? pb_h : n3;
                c2 = o == 2 * dl ? pb_h : c2; c3 = o == 3 * dl ? pb_h : c3;
            }
            const bool first = s.cs == 0;
            const Rule3D u = rules3d<DYN>(s, act, k, n0, n1, n2, n3, c2, c3, active, a.ts_done, a.brick_gt);   // the rules: snac_dev.h
            const bool built = u.built;
            const int newh = u.newh;
            s.cross += (built && newh <= pl) ? 1 : 0;                    // the tick's only vector-memory wait: pl, loaded a tick ago
            bool done = u.done;
            const int reward = u.sel ? reward_check3d(newh, pl) : u.reward0;
            done = done && active;
            s.ep_ret = clamp16(s.ep_ret + reward);
            s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
            pb_idx = built ? hidx + dl : -1; pb_h = newh;
            {   // the tick's outputs -> its half of the double buffer; the scalar slots by the exact-reciprocal quotients of Roll3D
                const double c0 = (double)s.cb, c1 = (double)s.cs;
                double v0 = c0, v1 = c1;
                if (VAR ? a.sc_norm != 0 : DYN) {
                    const double q0 = c0 * rtb, q1 = c1 * rT;
                    v0 = s.tb > 0 ? __builtin_fma(__builtin_fma(-q0, dtb, c0), rtb, q0) : c0 / dtb;
                    v1 = __builtin_fma(__builtin_fma(-q1, dT, c1), rT, q1);
                }
                double2 sv; sv.x = v0; sv.y = v1;
                *(double2*)&ssc[par][2 * lane] = sv;
                spub[par][lane] = make_int2(s.r | (s.c << 8) | (nr ? 1 << 16 : 0), built ? ((hidx + dl) | (newh << 16)) : -1);
                srew[par][lane] = (float)reward;
                sdone[par][lane] = done ? 1 : 0;
                if (done) sfin[par][lane] = make_int4(s.cross, s.tb + s.cb - s.cross, s.ep_ret, 0);
                if constexpr (VAR) {
                    int4* const rec = (int4*)&ssc[par][128 + 4 * lane];
                    rec[0] = make_int4(reward, done ? 1 : 0, s.r, s.c);
                    rec[1] = make_int4(s.cb, s.cs, s.tb, s.pidx);
                }
            }
            if (active) {
                const size_t row = (size_t)t * (size_t)a.n + (size_t)env;
                if (a.actions_out) a.actions_out[row] = (int8_t)act;
                if (a.step_size_out) a.step_size_out[row] = (int8_t)k;
                if (a.plan_idx_out) a.plan_idx_out[row] = (int16_t)s.pidx;
                if (a.first_out) a.first_out[row] = first ? 1 : 0;
            }
            // ---- the next tick's inputs, the scalars of a pending reset, and the plan cell of the next build target
            nr_prev = nr;
            if (t + 1 < a.T) {
                nr = done && a.auto_reset;
                if (__any(nr)) { if (nr) reset_scalars(); }
                if constexpr (EXPL) { act = act_n; k = k_n; }
                else inputs_of(t + 1, act, k);
                pl = (int)((const int16_t*)a.plans)[(size_t)s.pidx * GE + target_cell(act)];
                if constexpr (EXPL) load_inputs(t + 2, act_n, k_n);
            }
            lds_barrier();                                               // tick t is published; the writers are done with tick t - 1
            if constexpr (VAR) {
                if (plan_tail) {                                         // the tick's new plan rows -> plw, then the writers may assemble
                    if (pm) {
                        rows_write(pm);
                        while (pm_rest) {                                // more than PS at once (a time limit that many envs reach together)
                            const unsigned long long m = pm_rest;
                            rows_issue(pm_rest, pid_t);
                            rows_write(m);
                        }
                    }
                    lds_barrier();
                }
            }
        }
        __syncthreads();                                                 // the writers have brought the maps up to the last tick
        {
            int16_t* dst = (int16_t*)a.grid + (size_t)env0 * GE;
            const int16_t* hh = K::hmap(hm);
            for (int i = tid; i < nenv * GE; i += NT) {
                const int e = i / GE, cell = i - e * GE, r = cell / 20, c = cell - r * 20;
                dst[i] = hh[e * K::ES + (r + 3) * 26 + c + 3];
            }
        }
        if (active) { a.hdr[env] = s.pack(); a.episode[env] = episode; }
        return;
    }
    // ================================ the writers: 8 envs per wave ================================
    const int e0 = (wv - 1) * 8, el = lane >> 3, qt = lane & 7;     // qt: window row 0 .. 6, or 7: the two scalar slots
    const int we = e0 + el;                                          // this lane's env within the block
    const int rows = min(max(nenv - e0, 0), 8);                      // rows of this wave that exist
    const bool tl = a.obs_mode == SNAC_OBS_TILED;
    const int LD = VAR ? a.ld : D, ROWBV = LD * (int)sizeof(OT);     // values / bytes per row
    char* const obs0 = (char*)a.obs + ((tl ? ((size_t)(env0 >> 6) * (size_t)a.tiled_T + (size_t)a.tiled_t0) * 64 : (size_t)env0) + (size_t)e0) * ROWBV;
    const size_t tstride = (tl ? (size_t)64 : (size_t)a.n) * ROWBV;
    int16_t* const srow = stg16 + we * 56 + min(qt, 6) * 8;          // where this lane's window row goes (part 7: nowhere)
    int16_t* const hme = K::hmap(hm) + we * K::ES;
    const int16_t* const hq = hme + (min(qt, 6) - 3) * 26 - 3;
    // The wave's 8 rows leave as NP 16-byte pieces of VP values, piece lane + 64 q in lane's q-th store.  Value g of the slice is
    // element g % 51 of env g / 51: a window cell (an int16 of the staging tile, converted on the way out) or one of the two scalar
    // slots (a float64 the stepper published).  Where each of a lane's values comes from does not change from tick to tick:
    constexpr int VP = 16 / (int)sizeof(OT), NP = 8 * ROWB / 16, NQ = (NP + 63) / 64, NV = NQ * VP;
    // Only 16 of a wave's 408 values are scalar slots, at most KS of them in one lane's pieces: those are read as a short list (round 3
    // read a float64 for every value of every lane: 8 of a writer's ~28 LDS instructions per tick -- and what the nine waves of a
    // block do in LDS is what stretches the stepper's tick from 1000 to 1940 cycles).
    constexpr int KS = VP == 2 ? 3 : 4;
    int src[NV];                                                     // byte offset into stg16 (window cells)
    int ksel[NV];                                                    // -1: a window cell; else which entry of the lane's scalar list
    int ssrc[KS];                                                    // byte offsets into the tick's ssc half (unused entries: slot 0)
#pragma unroll
    for (int kq = 0; kq < KS; ++kq) ssrc[kq] = 0;
    {
        int nk = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int u = 0; u < VP; ++u) {
                const int pc = min(lane + 64 * q, NP - 1), g = pc * VP + u, e = g / 51, x = g - 51 * e;
                const bool sc_slot = x >= 49 && lane + 64 * q < NP;
                src[q * VP + u] = x < 49 ? ((e0 + e) * 56 + (x / 7) * 8 + x % 7) * 2 : 0;
                ksel[q * VP + u] = sc_slot ? min(nk, KS - 1) : -1;
#pragma unroll
                for (int kq = 0; kq < KS; ++kq) ssrc[kq] = (sc_slot && nk == kq) ? ((e0 + e) * 2 + (x - 49)) * 8 : ssrc[kq];
                nk += sc_slot ? 1 : 0;
            }
    }
    const int npieces = rows * ROWBV / 16;
    // VAR: one descriptor per value of this lane's pieces (piece lane + 64 q, value u): bits 0-15 the byte offset of a 2-byte cell in
    // `cells` (a window cell of the staging tile or a plan cell), 16-24 an 8-byte slot of the tick's ssc half, 30-31 the kind: 0 cell,
    // 1 float64 slot, 2 / 3 low / high int32 of a slot.  widemask: bit q = some lane's piece of iteration q holds a slot value.
    constexpr int NQV = PT ? (VP == 2 ? 29 : 15) : (VAR ? (VP == 2 ? 4 : 2) : 1), BQ = 8 / VP;   // iterations for the longest row (461 / 61 values); store instructions per batch
    uint32_t desc[NQV * VP];
    uint32_t widemask = 0;
    if constexpr (VAR) {
        const int pos_n = (a.tail & SNAC_TAIL_POSITION) ? 2 : 0, plan_n = PT ? GE : 0;
        const float rLD = 1.0f / (float)LD;
        const int nvals = max(rows * LD, 1);
#pragma unroll
        for (int q = 0; q < NQV; ++q) {
            bool wide = false;
#pragma unroll
            for (int u = 0; u < VP; ++u) {
                const int gi = min((lane + 64 * q) * VP + u, nvals - 1);
                const int e = (int)(((float)gi + 0.5f) * rLD), x = gi - e * LD, ea = e0 + e;   // exact: gi < 3700, the product is off by < 1e-5
                const int wi = (x * 37) >> 8;                            // x / 7 for x < 49
                const int ti = x - D, tp = ti - pos_n, tr = tp - plan_n;
                const bool is_win = x < 49, is_sc = (unsigned)(x - 49) < 2u, is_plan = (unsigned)tp < (unsigned)plan_n;
                const int j = min(max(ti < pos_n ? 2 + ti : tr, 0), 7);  // which record value, for a value that is none of the above
                const int off = is_plan ? (64 * 56 + ea * GE + tp) * 2 : (ea * 56 + (is_win ? wi * 8 + (x - 7 * wi) : 0)) * 2;
                const int slot = is_sc ? ea * 2 + (x - 49) : 128 + ea * 4 + (j >> 1);
                const int kind = (is_win || is_plan) ? 0 : (is_sc ? 1 : 2 + (j & 1));
                desc[q * VP + u] = (uint32_t)off | ((uint32_t)slot << 16) | ((uint32_t)kind << 30);
                asm volatile("" : "+v"(desc[q * VP + u]));               // one register per value from here on (not its parts)
                wide = wide || kind != 0;
            }
            widemask |= __any(wide) ? 1u << q : 0u;
        }
        if (plan_tail) {                                                 // the rows of tick 0 (with the stepper, above)
            __syncthreads();
            plan_rows_in();
            __syncthreads();
        }
    }
    int d_eps = 0, d_ret = 0;
    long long d_iou = 0;
    for (int t = 0; t < a.T; ++t) {
        const int par = t & 1;
        lds_barrier();
        const int2 pub = spub[par][we];
        // ---- the maps of this wave's envs take the tick: an env that started over is cleared (all lanes, row by row), then the built cell
        for (unsigned long long mk = __ballot(qt == 0 && (pub.x >> 16) != 0); mk; mk &= mk - 1) {
            const int e = e0 + ((__ffsll(mk) - 1) >> 3);
            if (lane < 20) {                                             // a row's first interior cell has an odd index: 1 + 9 x 2 + 1 cells
                int16_t* const r16 = K::hmap(hm) + e * K::ES + (lane + 3) * 26 + 3;
                r16[0] = 0;
                uint32_t* const r32 = (uint32_t*)(r16 + 1);
#pragma unroll
                for (int q = 0; q < 9; ++q) r32[q] = 0u;
                r16[19] = 0;
            }
        }
        if (qt == 0 && pub.y >= 0) hme[pub.y & 0xffff] = (int16_t)(pub.y >> 16);
        // ---- gather: every read before the first write (the compiler cannot tell the staging tile from the maps)
        // the window row's 7 cells start at any cell of the map: the 8 cells from the even cell at or below it are four aligned dwords
        // (two ds_read2_b32 instead of seven ds_read_u16), shifted down a cell when the row starts on an odd one.  Part 7 reads a row
        // it does not use; the eighth cell lies inside the block's maps for every position.
        const int coff = (int)(hq - K::hmap(hm)) + (pub.x & 0xff) * 26 + ((pub.x >> 8) & 0xff);   // the row's first cell, in cells
        const uint32_t* const cw = hm + (coff >> 1);
        const uint32_t d0 = cw[0], d1 = cw[1], d2 = cw[2], d3 = cw[3];
        if (qt < 7) {                                                    // one aligned 16-byte write (a misaligned 14-byte row, then read back
            const uint32_t shb = (uint32_t)(coff & 1) * 16u;             // cell by cell, cost 0.45 us per tick: 1.52 instead of 1.08 ms)
            uint4 w;
            w.x = __builtin_amdgcn_alignbit(d1, d0, shb); w.y = __builtin_amdgcn_alignbit(d2, d1, shb);
            w.z = __builtin_amdgcn_alignbit(d3, d2, shb); w.w = (d3 >> shb) & 0xffffu;
            *(uint4*)srow = w;
        }
        // the wave's rows leave: its own LDS writes are visible to its own reads in order
        if constexpr (VAR) {
            if (plan_tail) lds_barrier();                                // the stepper has put the tick's new plan rows into plw
            char* const g = obs0 + (size_t)t * tstride;
            const char* const cb = (const char*)cells;
            const char* const wb = (const char*)ssc[par];
#pragma unroll
            for (int q0 = 0; q0 < NQV; q0 += BQ) {
                if (q0 * 64 >= npieces) break;                           // (wave-uniform)
                constexpr int NB = BQ * VP;
                // PT: the descriptors are opaque to the compiler in every tick -- left alone it hoists three derived offsets per descriptor out
                // of the tick loop (3 x 58 registers) and spills, and a spill reload inside the loop waits behind the row stores like any load
                uint32_t dw[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    uint32_t& d = desc[min(q0 * VP + i, NQV * VP - 1)];
                    if constexpr (PT) asm volatile("" : "+v"(d));
                    dw[i] = d;
                }
                int cv[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) cv[i] = (int)*(const int16_t*)(cb + (dw[i] & 0xffffu));
                // iterations with a scalar slot / record value among their values (every fourth or so of a 451-value row's): 8-byte reads
                // (float64 rows with the plan tail: an iteration's 8-byte reads just before they are used -- read up front for the whole batch
                // they cost 16 registers the kernel does not have: spills inside the tick loop, 1.90 instead of 1.73 ms per 200 ticks at 16 384 envs)
                constexpr bool EARLY = !(PT && VP == 2);
                uint2 w2[NB];
#pragma unroll
                for (int kq = 0; kq < BQ; ++kq) {
                    if (EARLY && ((widemask >> (q0 + kq)) & 1u)) {
#pragma unroll
                        for (int u = 0; u < VP; ++u) w2[kq * VP + u] = *(const uint2*)(wb + ((dw[kq * VP + u] >> 13) & 0xff8u));
                    }
                }
                OT val[NB];
#pragma unroll
                for (int kq = 0; kq < BQ; ++kq) {
                    if ((widemask >> (q0 + kq)) & 1u) {
                        if constexpr (!EARLY) {
#pragma unroll
                            for (int u = 0; u < VP; ++u) w2[kq * VP + u] = *(const uint2*)(wb + ((dw[kq * VP + u] >> 13) & 0xff8u));
                        }
#pragma unroll
                        for (int u = 0; u < VP; ++u) {                   // by masks, not selects: the compiler turns selects round a conversion into branches
                            const int i = kq * VP + u;
                            const uint32_t kd = dw[i] >> 30;
                            int iv = cv[i];
                            iv = kd == 2 ? (int)w2[i].x : iv;
                            iv = kd == 3 ? (int)w2[i].y : iv;
                            const uint32_t m = (uint32_t)-(int)(kd == 1);
                            if constexpr (VP == 2) {
                                const uint64_t b = (uint64_t)__double_as_longlong((double)iv);
                                const uint32_t lo = (m & w2[i].x) | (~m & (uint32_t)b), hi = (m & w2[i].y) | (~m & (uint32_t)(b >> 32));
                                val[i] = __hiloint2double((int)hi, (int)lo);
                            } else {
                                const float fs = (float)__hiloint2double((int)w2[i].y, (int)w2[i].x);
                                val[i] = __uint_as_float((m & __float_as_uint(fs)) | (~m & __float_as_uint((float)iv)));
                            }
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < VP; ++u) val[kq * VP + u] = (OT)cv[kq * VP + u];
                    }
                }
#pragma unroll
                for (int kq = 0; kq < BQ; ++kq) {
                    const int pc = lane + 64 * (q0 + kq);
                    if (q0 + kq < NQV && pc < npieces) {
                        if constexpr (VP == 2) {
                            double2 o; o.x = val[2 * kq]; o.y = val[2 * kq + 1];
                            *(double2*)(g + (uint32_t)pc * 16u) = o;
                        } else {
                            float4 o; o.x = val[4 * kq]; o.y = val[4 * kq + 1]; o.z = val[4 * kq + 2]; o.w = val[4 * kq + 3];
                            *(float4*)(g + (uint32_t)pc * 16u) = o;
                        }
                    }
                }
            }
        } else {
            char* const g = obs0 + (size_t)t * tstride;
            const char* const cells = (const char*)stg16;
            const char* const scs = (const char*)ssc[par];
            int ci[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) ci[i] = (int)*(const int16_t*)(cells + src[i]);
            double sc[KS];                                               // all reads of the tile before anything waits
#pragma unroll
            for (int kq = 0; kq < KS; ++kq) sc[kq] = *(const double*)(scs + ssrc[kq]);
            OT val[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                double sv = sc[0];
#pragma unroll
                for (int kq = 1; kq < KS; ++kq) sv = ksel[i] == kq ? sc[kq] : sv;
                val[i] = ksel[i] >= 0 ? (OT)sv : (OT)ci[i];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int pc = lane + 64 * q;
                if (pc < npieces) {
                    if constexpr (VP == 2) {
                        double2 o; o.x = val[2 * q]; o.y = val[2 * q + 1];
                        *(double2*)(g + pc * 16) = o;
                    } else {
                        float4 o; o.x = val[4 * q]; o.y = val[4 * q + 1]; o.z = val[4 * q + 2]; o.w = val[4 * q + 3];
                        *(float4*)(g + pc * 16) = o;
                    }
                }
            }
        }
        const size_t row = (size_t)t * (size_t)a.n + (size_t)env0;
        if (wv == 1 && a.reward && lane < nenv) a.reward[row + lane] = srew[par][lane];
        if (wv == 2 && a.done) {
            if (nenv == 64 && ((((uintptr_t)a.done) | (uintptr_t)a.n) & 3) == 0) {
                if (lane < 16) ((uint32_t*)(a.done + row))[lane] = ((const uint32_t*)sdone[par])[lane];
            } else if (lane < nenv) a.done[row + lane] = sdone[par][lane];
        }
        // iou (:257-276) = sum(min(g, plan)) / (tb + cb - sum) and the sums of an episode that ended, kept by quarter 0 of the env
        const bool fin = qt == 0 && sdone[par][we] != 0;
        if (__builtin_expect(__any(fin), 0)) {
            asm volatile("" ::: "memory");
            if (fin) {
                const int4 f = sfin[par][we];
                const double v = (double)f.x / (double)f.y;
                d_eps += 1; d_ret += f.z; d_iou += __double2ll_rn(v * FX40);
            }
        }
    }
    __syncthreads();
    {
        int16_t* dst = (int16_t*)a.grid + (size_t)env0 * GE;
        const int16_t* hh = K::hmap(hm);
        for (int i = tid; i < nenv * GE; i += NT) {
            const int e = i / GE, cell = i - e * GE, r = cell / 20, c = cell - r * 20;
            dst[i] = hh[e * K::ES + (r + 3) * 26 + c + 3];
        }
    }
    if (qt == 0 && we < nenv && d_eps) {
        a.stat_episodes[env0 + we] += d_eps;
        a.stat_return[env0 + we] += d_ret;
        a.stat_iou_fx[env0 + we] += d_iou;
    }
}


template <bool DYN, typename OT, int MODE>
void launch_roll3db_w(const KArgs& a, hipStream_t s) {
    const int blocks = (a.n + 63) / 64;
    const dim3 grid((unsigned)(((blocks + 7) / 8) * 8)), block(576);   // a multiple of 8: the XCD remap covers every block
    if (a.actions || a.step_size) hipLaunchKernelGGL((k_rollout3db<DYN, OT, true, MODE>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_rollout3db<DYN, OT, false, MODE>), grid, block, 0, s, a);
}

}  // namespace
