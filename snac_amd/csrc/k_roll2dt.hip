// k_roll2dt.hip -- k_rollout2dt: time-parallel 2D rollouts
#include "snac_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// 2D fused rollout, TIME-parallel (round 4) -- the small-batch counterpart of k_rollout2d.  Below ~16 000 envs every 2D rollout kernel
// is bound by the chain of its ticks (0.5-0.7 ms per 600 ticks whatever N: one wave walks 600 dependent steps), and that is the range
// the reference is used in (multiprocess.py:96: --num_envs 3; every script/* drives one env).  In 2D, too, the CONTROL of an episode
// depends on the actions alone (DMP_Env_2D_dynamic_usedata_plan.py:85-147: moves only clamp, a drop never moves, count_step counts
// ticks, count_brick counts drops), so one wavefront takes ONE env and 64 consecutive ticks, lane j = tick t0 + j, as k_rollout1dt:
//   counters   count_step / count_brick by lane index and drop-ballot prefix; done = the first lane whose counters say so; the lanes up
//              to it are a segment, the rest of the chunk a second one behind the reset (wave-uniform state);
//   position   row and column are two chains of x -> min(max(x + d, 3), 22): two inclusive DPP scans of the composed clamps;
//   the board  at the chunk's start: 20 row words in LDS.  A tick's window = those rows OR the bricks dropped earlier in the chunk:
//              the droppers are walked in a wave-uniform loop (their cells by v_readlane), every later lane marks the cell in a 49-bit
//              mask if it falls into its window, a later dropper on the same cell learns that the cell was taken ("was"); afterwards
//              each dropper ORs its bit into the board (ds_or_b32).  No per-cell lane masks, no prefix-OR over rows: ~14 vector
//              instructions per dropper, ~13 droppers per chunk;
//   the rows   every lane files its row COMPACT -- the 7 window row codes (2 bits per cell, k_rollout2d's encoding) and the two scalar
//              slots, 32 bytes -- in a staging tile [tick][env of the block]; behind a barrier the block's threads expand it on the way
//              out: a tick's rows of the block's EB envs are one run of EB x 408 bytes, stored 16 bytes per lane (the source of every
//              lane's values in a run does not depend on the tick and is worked out once per launch).
// ~8 + 4.5 wave-instructions per env-step (the lane-per-env kernel: 4.2), but nothing waits for the tick before: N = 1024 x 600 ticks
// takes ~0.03 ms instead of 0.51.  Semantics are K2D::step's; counter-RNG or explicit inputs; SNAC_OBS_ALL / SNAC_OBS_TILED, canonical layout.
// Row assembly for k_rollout2dt's layout variants: the rows of ONE tick's nenv (<= 4, even) envs from their compact records
// rec[e * 16 ..] (codes 2 per dword, the two scalar doubles, the record's eight ints) and the envs' plan rows planw[e * 20 ..], through
// the calling wave's staging tile (STG bytes) to g, 16 bytes per lane.  emit_rows_var's scheme -- lane = value: lanes 0 .. 60 the head
// (window cells, scalar slots, position, record), lane + 64 i the plan cells -- cut down to few registers (one or two envs at a time,
// nothing kept across them), so that sixteen waves of 128 registers fit a CU: with emit_rows_var inlined the writers needed 256.
template <typename OT, int STG>
__device__ __forceinline__ void emit_rows_lean(char* stg, const uint32_t* rec, const uint32_t* planw, char* g, int lane, int nenv, int LD,
                                               int tail, int frame_val) {
    constexpr int D = 51, W = 49;
    const int RB = LD * (int)sizeof(OT);
    const int G = 4 * RB <= STG ? 4 : 2;                             // envs per flush: G * RB is a multiple of 16 (float32 rows: always 4)
    const int pos_n = (tail & SNAC_TAIL_POSITION) ? 2 : 0, plan_n = (tail & SNAC_TAIL_PLAN) ? 400 : 0, rec_n = (tail & SNAC_TAIL_RECORD) ? 8 : 0;
    const int NE = D + pos_n + rec_n;
    // this lane's head value: dword of the record, first bit of a cell's field, kind masks, place in the row
    int src, off = 0, dst = lane;
    uint32_t m_sc = 0u, m_int = 0u;
    if (lane < W) { const int i = lane / 7, j = lane - 7 * i; src = i >> 1; off = 2 * j + 16 * (i & 1); }
    else if (lane < D) { src = 4 + 2 * (lane - W); m_sc = ~0u; }
    else {
        int k = lane - D;
        m_int = ~0u;
        if (k < pos_n) { src = 10 + k; dst = D + k; }
        else { k -= pos_n; src = 8 + min(k, 7); dst = D + pos_n + plan_n + k; }
    }
    for (int e0 = 0; e0 < nenv; e0 += G) {
        const int ge = min(G, nenv - e0);
#pragma unroll 1
        for (int e = e0; e < e0 + ge; e += 2) {                      // two envs at a time: their LDS reads first
            uint32_t lo[2], hi[2], pw[2][7];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t* const c = rec + (e + u) * 16 + src;
                lo[u] = c[0]; hi[u] = c[1];
                if (plan_n) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) pw[u][i] = planw[(e + u) * 20 + min(lane + 64 * i, 399) / 20];
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cv = __builtin_amdgcn_sbfe((int)lo[u], (uint32_t)off, 2u);        // 0 / 1 / -1 (frame)
                const uint32_t iv = bfi32(m_int, lo[u], (uint32_t)(cv < 0 ? frame_val : cv));
                const uint64_t cb = (uint64_t)__double_as_longlong((double)(int)iv);
                const uint32_t rl = bfi32(m_sc, lo[u], (uint32_t)cb), rh = bfi32(m_sc, hi[u], (uint32_t)(cb >> 32));
                const double val = __longlong_as_double((long long)(((uint64_t)rh << 32) | rl));
                OT* const row = (OT*)stg + (e + u - e0) * LD;
                if (lane < NE) row[dst] = (OT)val;
                if (plan_n) {
                    OT* const q = row + D + pos_n;
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        const int pc = min(lane + 64 * i, 399);
                        if (i < 6 || lane < 16) q[lane + 64 * i] = bit_as<OT>(pw[u][i], pc - 20 * (pc / 20));
                    }
                }
            }
        }
        // the group leaves: ge * RB bytes, a multiple of 16
        const int valid = ge * RB;
        char* const gh = g + (size_t)e0 * RB + lane * 16;
        const char* const sh = stg + lane * 16;
        for (int i = 0; i * 1024 < valid; i += 4) {                  // four 1 KiB store instructions at a time, their LDS reads first
            uint4 fv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) fv[k] = *(const uint4*)(sh + min((i + k) * 1024, STG - 1024));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((i + k) * 1024 + lane * 16 < valid) *(uint4*)(gh + (i + k) * 1024) = fv[k];
        }
    }
}

// VAR: the layout variants of snac_env_desc (frame value, raw / normalised counters, position / plan / record tails: rows of a.ld
// values).  The steppers file eight more dwords per row (reward, done, position, counters, total_brick, plan row), and only the WR
// writer waves expand: a writer assembles a tick's EB rows from their compact rows with emit_rows_lean (k_rollout2d's scheme:
// lane = value, groups of envs through a staging tile of its own, 16 bytes per lane out).  The plan tail's cells come
// from a per-writer copy of each env's plan row in LDS, refilled through the scalar cache when a tick's row differs from the copy
// (any number of resets per chunk).  N % 4 = 0 and a 16-byte aligned output.
template <bool DYN, typename OT, int EB, bool EXPL, bool VAR = false, int WR = EB>
__global__ __launch_bounds__((EB + WR) * 64) void k_rollout2dt(const KArgs a) {
    using K = K2D<DYN, 64>;
    constexpr int D = K::D, GE = K::GE;
    constexpr int ROWB = D * (int)sizeof(OT);                        // 408 / 204 bytes per row
    constexpr int RECW = VAR ? 16 : 8;                               // dwords per compact row: 7 codes in 4 dwords, two doubles (+ the record's 8 values)
    constexpr int TSTR = EB * RECW + 4;                              // staging dwords per tick (+4: the lanes' 16-byte writes spread over the banks)
    constexpr int VSTG = 8192;                                       // VAR: a writer's staging tile (two 451-value float64 rows)
    static_assert(VAR || WR == EB, "the canonical layout splits reward / done by writer wave");
    // A block is 2 EB waves: EB STEPPERS (one env each: the control chain of a chunk of 64 ticks, compact rows into staging buffer c & 1)
    // and EB WRITERS, which expand the chunk before (buffer (c - 1) & 1) while the steppers are at the next one -- one barrier per
    // chunk.  With one wave per SIMD (N <= 1024) a chunk costs max(stepping, expanding) instead of their sum.  The ticks to expand are
    // a queue both kinds of wave draw from (the steppers once their chunk is stepped): the two halves of a chunk level out at every N.
    __shared__ uint32_t sG[EB][GE], sP[EB][GE];
    __shared__ __align__(16) uint32_t stage2[2][64 * TSTR];
    __shared__ float sR2[2][64][EB + 1];
    __shared__ __align__(16) uint8_t sD2[2][64][EB];
    __shared__ unsigned int tickq[2];                                // next tick to expand, per staging buffer
    __shared__ __align__(16) char vstg[VAR ? WR : 1][VAR ? VSTG : 16];
    __shared__ uint32_t vplan[VAR ? WR : 1][VAR ? EB * GE : 1];
    const int tid = (int)threadIdx.x, lane = tid & 63, wall = tid >> 6;
    const bool stepper = wall < EB;
    const int wv = stepper ? wall : (VAR ? wall - EB : (wall & (EB - 1)));   // the stepper's env of the block / the writer's index
    const int env0 = (int)blockIdx.x * EB;
    const int nenv = min(EB, a.n - env0);                            // block-uniform; > 0 by the grid
    const bool own = stepper && wv < nenv;                           // steppers past the batch only keep the barriers company
    const int env = env0 + ((stepper && wv < nenv) ? wv : 0);
    uint32_t* const G = sG[stepper ? wv : 0];                        // the board as the current chunk found it: 20 interior row words
    uint32_t* const P = sP[stepper ? wv : 0];                        // the env's plan rows
    Lane s;
    s.unpack(a.hdr[env]);
    int episode = a.episode[env];
    asm volatile("" : "+v"(episode));
    if (stepper && lane < GE) {                                      // (the writers share the index wv: they must not touch these)
        G[lane] = ((const uint32_t*)a.grid)[(size_t)env * GE + lane];
        P[lane] = ((const uint32_t*)a.plans)[(size_t)s.pidx * GE + lane];
    }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const EnvKeys sk = env_keys(a.key_step, gid), pk = env_keys(a.key_plan, gid);
    // wave-uniform env state (every lane holds the same values)
    int r0 = s.r, c0 = s.c, cb0 = s.cb, cs0 = s.cs, ret0 = s.ep_ret, tb = s.tb, pidx = s.pidx;
    asm volatile("" : "+v"(tb));                                     // the header has arrived HERE, not at a wait inside the loop
    double dtb = (double)tb, rtb = 1.0 / dtb;                        // once per episode (Roll3D, tests/native/recip_check.c)
    bool need_reset = a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    bool flag_done = (s.flags & SNAC_FLAG_NEED_RESET) != 0;
    int d_eps = 0, d_ret = 0;
    long long d_iou = 0;
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);      // lanes <= this one
    const bool tl = a.obs_mode == SNAC_OBS_TILED;
    const double dT = (double)a.total_step, rT = 1.0 / dT;
    // the block's rows of one tick are one run of nenv x ROWB bytes; 16-byte pieces when every run starts and ends on 16 bytes
    const int RB = VAR ? a.ld * (int)sizeof(OT) : ROWB;              // bytes per row
    const size_t ostr = (tl ? (size_t)64 : (size_t)a.n) * RB;        // bytes from one tick's run to the next
    const bool vec = ((((uintptr_t)a.obs) | (uintptr_t)ostr | (uintptr_t)((size_t)nenv * RB)) & 15) == 0;
    const bool dvec = EB == 16 && a.done && nenv == EB && ((((uintptr_t)a.done) | (uintptr_t)a.n) & 15) == 0;
    int ptag[EB];                                                    // VAR writers: the plan row each env's LDS copy holds
#pragma unroll
    for (int e = 0; e < EB; ++e) ptag[e] = -1;
    // ---- what this lane expands when a run leaves: piece lane + 64 q of the run holds VP values; value v of it is element el of env e of
    // the block -- a window cell (source: code i of the env's compact row, 2-bit field j) or a scalar slot.  The same for every tick.
    constexpr int VP = 16 / (int)sizeof(OT);                         // values per 16-byte piece
    constexpr int PTMAX = EB * ROWB / 16, NQ = (PTMAX + 63) / 64;
    int fsrc[NQ][VP];                                                // dword offset in the tick's staging row | first bit of the cell in its code word << 16 | scalar << 24
    if (!VAR && vec) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int v = 0; v < VP; ++v) {
                const int gel = min((lane + 64 * q) * VP + v, EB * D - 1);
                const int e = gel / D, el = gel - e * D;
                if (el < K::W) {
                    const int i = el / 7, j = el - 7 * i;
                    fsrc[q][v] = (e * RECW + (i >> 1)) | ((2 * j + (i & 1) * 16) << 16);   // the cell's two bits: their place in the code word
                } else {
                    fsrc[q][v] = (e * RECW + 4 + 2 * (el - K::W)) | (1 << 24);
                }
            }
    }
    const int nchunks = (a.T + 63) / 64;
    for (int ch = 0; ch <= nchunks; ++ch) {
        const int t0 = ch * 64;
        const int nl = min(64, a.T - t0);
        uint32_t* const stage = stage2[ch & 1];
        float (*const sR)[EB + 1] = sR2[ch & 1];
        uint8_t (*const sD)[EB] = sD2[ch & 1];
        if (tid == 2 * EB * 64 - 1) tickq[ch & 1] = 0;               // the queue of THIS chunk's ticks, drawn from in the next round
        if (own && ch < nchunks) {
        const bool valid = lane < nl;
        const int t = t0 + lane;
        const size_t row = (size_t)t * (size_t)a.n + (size_t)env;
        const uint32_t w = rng_word(sk, a.t0 + (uint32_t)t);
        int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
        if constexpr (EXPL) {
            if (a.actions && valid) act = (int)a.actions[row];
            if (a.step_size && valid) k = min(max((int)a.step_size[row], 1), 3);
        }
        int first_lane = 0;                                          // the segment's first lane
        while (first_lane < nl) {
            if (need_reset) {                                        // K2D::reset in the uniform state (rare: once per episode)
                episode += 1;
                const int np = pick_plan<K>(a, pk, episode, pidx);
                if (np != pidx) {                                    // a new row brings its total_brick, the same row keeps the header's
                    pidx = np; tb = (int)a.plan_tb[np];
                    asm volatile("" : "+v"(tb));
                    dtb = (double)tb; rtb = 1.0 / dtb;
                    if (lane < GE) P[lane] = ((const uint32_t*)a.plans)[(size_t)np * GE + lane];
                }
                if (lane < GE) G[lane] = 0u;
                r0 = 3; c0 = 3; cb0 = 0; cs0 = 0; ret0 = 0;
                need_reset = false;
            }
            const bool seg = valid && lane >= first_lane;
            const bool drop = seg && act == 4;
            // ---- counters and the segment's end
            const int cs = min(cs0 + (lane - first_lane + 1), CNT_MAX);
            const unsigned long long dropm = __ballot(drop);
            const int cb = min(cb0 + (int)__popcll(dropm & le), CNT_MAX);
            const bool term = term_rule(drop, cb, tb, a.brick_gt);   // :117-126, before the time limit (the rules' pieces: snac_dev.h)
            const bool done = seg && (term || cs >= a.ts_done);
            const unsigned long long donem = __ballot(done);
            const int last = donem ? (__ffsll((long long)donem) - 1) : (nl - 1);     // the segment's last lane
            const bool in = seg && lane <= last;
            // ---- positions: two inclusive scans of x -> min(max(x + d, 3), 22) (clip_position :74-83; "up" is row + k, :100-103)
            int ra = 0, rlo = -4096, rhi = 4096, ca = 0, clo = -4096, chi = 4096;
            if (in) {
                ra = act == 2 ? k : (act == 3 ? -k : 0); rlo = 3; rhi = 22;
                ca = act == 1 ? k : (act == 0 ? -k : 0); clo = 3; chi = 22;
            }
            auto compose = [&](int pa, int plo, int phi, int& sa, int& slo, int& shi) {   // the earlier ticks first, then this lane's function
                const int nlo = min(max(plo + sa, slo), shi), nhi = min(max(phi + sa, slo), shi);
                sa += pa; slo = nlo; shi = nhi;
            };
#define SNAC_SCAN_STEP(CTRL, ROWS)                                                                                               \
            {                                                                                                                    \
                const int pa = dpp_from<CTRL, ROWS>(0, ra), plo = dpp_from<CTRL, ROWS>(-4096, rlo), phi = dpp_from<CTRL, ROWS>(4096, rhi); \
                const int qa = dpp_from<CTRL, ROWS>(0, ca), qlo = dpp_from<CTRL, ROWS>(-4096, clo), qhi = dpp_from<CTRL, ROWS>(4096, chi); \
                compose(pa, plo, phi, ra, rlo, rhi);                                                                             \
                compose(qa, qlo, qhi, ca, clo, chi);                                                                             \
            }
            SNAC_SCAN_STEP(0x111, 0xf) SNAC_SCAN_STEP(0x112, 0xf) SNAC_SCAN_STEP(0x114, 0xf) SNAC_SCAN_STEP(0x118, 0xf)
            SNAC_SCAN_STEP(0x142, 0xa) SNAC_SCAN_STEP(0x143, 0xc)
#undef SNAC_SCAN_STEP
            const int pr = min(max(r0 + ra, rlo), rhi), pc = min(max(c0 + ca, clo), chi);   // after the tick
            const int prv_r = dpp_from<0x138>(r0, pr), prv_c = dpp_from<0x138>(c0, pc);
            const int br = lane == first_lane ? r0 : prv_r, bc = lane == first_lane ? c0 : prv_c;   // before the tick: where a drop lands
            // ---- the window round the new position from the board as the chunk found it (k_step2d's encoding) ...
            uint32_t wr[7];
            uint32_t gdrop = G[min(max(br - 3, 0), GE - 1)], pdrop = P[min(max(br - 3, 0), GE - 1)];   // the drop's row: board and plan
            {
                const int sh = pc - 3;                               // first window column, bordered: 0..19
                constexpr uint32_t FRAME26 = 0x3800007u;             // frame columns 0-2 and 23-25 of an interior row
                const uint32_t frm = spread16((FRAME26 >> sh) & 0x7Fu) * 3u;
                uint32_t g[7];                                       // the seven rows in ONE round trip: left to the compiler each read
#pragma unroll                                                       // sinks into its row's `inb` branch and is waited for there
                for (int i = 0; i < 7; ++i) g[i] = G[min(max(pr - 6 + i, 0), GE - 1)];
                asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(gdrop), "+v"(pdrop));
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const int q = pr - 6 + i;                        // board row of window row i
                    const bool inb = (unsigned)q < (unsigned)GE;
                    wr[i] = inb ? (spread16(((g[i] << 3) >> sh) & 0x7Fu) | frm) : 0x3FFFu;
                }
            }
            // ... OR the bricks dropped earlier in this segment: every dropper in turn (wave-uniform), its cell against each later
            // lane's window, and against each later dropper's own cell ("was": the cell was taken by then)
            const int bcell = br * 32 + bc;                          // where this lane's drop lands
            bool was = ((gdrop >> (bc - 3)) & 1u) != 0u;
            const bool planned = ((pdrop >> (bc - 3)) & 1u) != 0u;
            unsigned long long dmask = 0ull;
            const unsigned long long inm = __ballot(in);
            for (unsigned long long m = dropm & inm; m; m &= m - 1) {
                const int L = __ffsll((long long)m) - 1;
                const int cellL = __builtin_amdgcn_readlane(bcell, L);
                const int di = (cellL >> 5) - (pr - 3), dj = (cellL & 31) - (pc - 3);
                if (lane >= L && (unsigned)di < 7u && (unsigned)dj < 7u) dmask |= 1ull << (di * 7 + dj);
                was = was || (lane > L && bcell == cellL);
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) wr[i] |= spread16((uint32_t)(dmask >> (7 * i)) & 0x7Fu);
            const int reward = reward2d(drop, term, was, planned);   // un-clamped cell vs plan (:129-133)
            const unsigned long long r5 = __ballot(in && reward != 0);
            const int ret = clamp16(ret0 + 5 * (int)__popcll(r5 & le));
            // ---- outputs of the segment's lanes: compact rows into the block's staging tile
            if (in) {
                const double q0v = (double)cb, q1v = (double)cs;
                double v0 = q0v, v1 = q1v;
                if (VAR ? (a.sc_norm != 0) : DYN) {                  // cb / tb, cs / T: correctly rounded (Roll3D, tests/native/recip_check.c)
                    const double q0 = q0v * rtb, q1 = q1v * rT;
                    v0 = tb > 0 ? __builtin_fma(__builtin_fma(-q0, dtb, q0v), rtb, q0) : q0v / dtb;
                    v1 = __builtin_fma(__builtin_fma(-q1, dT, q1v), rT, q1);
                }
                uint32_t* const o = stage + lane * TSTR + wv * RECW;
                const uint64_t b0 = (uint64_t)__double_as_longlong(v0), b1 = (uint64_t)__double_as_longlong(v1);
                *(uint4*)o = make_uint4(wr[0] | (wr[1] << 16), wr[2] | (wr[3] << 16), wr[4] | (wr[5] << 16), wr[6]);
                *(uint4*)(o + 4) = make_uint4((uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32));
                if constexpr (VAR) {                                 // SNAC_TAIL_RECORD's values (record_value), position, the plan row
                    *(uint4*)(o + 8) = make_uint4((uint32_t)reward, (lane == last && donem) ? 1u : 0u, (uint32_t)pr, (uint32_t)pc);
                    *(uint4*)(o + 12) = make_uint4((uint32_t)cb, (uint32_t)cs, (uint32_t)tb, (uint32_t)pidx);
                }
                sR[lane][wv] = (float)reward;
                sD[lane][wv] = (lane == last && donem) ? 1 : 0;
                if (a.actions_out) a.actions_out[row] = (int8_t)act;
                if (a.step_size_out) a.step_size_out[row] = (int8_t)k;
                if (a.plan_idx_out) a.plan_idx_out[row] = (int16_t)pidx;
                if (a.first_out) a.first_out[row] = cs == 1 ? 1 : 0;
                if (drop) atomicOr(&G[br - 3], 1u << (bc - 3));      // += 1 then clamp to 1 (:115, :134-135): the board takes the brick
            }
            // ---- the segment's end: the uniform state moves on
            r0 = __builtin_amdgcn_readlane(pr, last); c0 = __builtin_amdgcn_readlane(pc, last);     // `last` is uniform
            cb0 = __builtin_amdgcn_readlane(cb, last); cs0 = __builtin_amdgcn_readlane(cs, last); ret0 = __builtin_amdgcn_readlane(ret, last);
            flag_done = donem != 0ull;
            if (donem) {                                             // boolean IoU of the finished episode (script/DQN/2d/DQN_2d_dynamic.py:63-71), episodic sums
                asm volatile("" ::: "memory");                       // once per episode: stays a branch
                const uint32_t g = lane < GE ? G[lane] : 0u, p = lane < GE ? P[lane] : 0u;
                int inter = __popc(g & p), uni = __popc(g | p);
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) { inter += __shfl_xor(inter, off); uni += __shfl_xor(uni, off); }
                inter = __builtin_amdgcn_readfirstlane(inter); uni = __builtin_amdgcn_readfirstlane(uni);
                const double v = (double)inter / (double)uni;
                d_eps += 1; d_ret += ret0; d_iou += __double2ll_rn(v * FX40);
                need_reset = a.auto_reset != 0;
            }
            first_lane = last + 1;
        }
        }
        // ---- the chunk before leaves: per tick one run of the block's rows, expanded from the compact rows.  The ticks are a QUEUE
        // (a counter in LDS): the writer waves draw from it from the start, the stepper waves once their chunk is stepped -- from
        // 2048 envs on the expansion is the longer half of a chunk (writers alone 1.0e10 env-steps/s at N = 4096, steppers alone
        // 2.2e10), below it the stepping: whoever is free takes the next tick
        if (ch > 0) {
            const int t0 = (ch - 1) * 64;
            const int nl = min(64, a.T - t0);
            const uint32_t* const stage = stage2[(ch - 1) & 1];
            const float (*const sR)[EB + 1] = sR2[(ch - 1) & 1];
            const uint8_t (*const sD)[EB] = sD2[(ch - 1) & 1];
            unsigned int* const queue = &tickq[(ch - 1) & 1];
            int t0v = t0, wq = wv, lq = lane;
            asm volatile("" : "+s"(t0v), "+v"(wq), "+v"(lq));        // addresses from scratch every chunk (k_rollout1dt)
            const size_t row0 = tl ? ((size_t)(env0 >> 6) * (size_t)a.tiled_T + (size_t)(a.tiled_t0 + t0v)) * 64 + (size_t)(env0 & 63)
                                   : (size_t)t0v * (size_t)a.n + (size_t)env0;
            char* const ob = (char*)a.obs + row0 * RB;
            constexpr int TPW = 64 / EB;                             // ticks per writer wave (reward / done)
            // the next tick of the queue, wave-uniform -- in two halves, so that the counter's round trip can run beside the LDS reads
            // of the tick in hand (LDS answers in order: behind those reads the draw has arrived too)
            auto draw_issue = [&]() -> int {
                int v = 0;
                if (lq == 0) v = (int)atomicInc(queue, 0xffffffffu);   // (ds_inc_rtn_u32: the compiler's wave-aggregation
                return v;                                                           // of atomicAdd waits for its answer on the spot)
            };
            auto draw = [&]() -> int { return __builtin_amdgcn_readfirstlane(draw_issue()); };
            auto value = [&](const uint32_t* rec, int el) -> OT {    // element el of the compact row rec
                if (el < K::W) {
                    const int i = el / 7, j = el - 7 * i;
                    const uint32_t c = rec[i >> 1] >> ((i & 1) * 16);
                    return (OT)(((int)(c << (30 - 2 * j))) >> 30);  // signed 2-bit field: 0 / 1 / -1
                }
                return (OT)__longlong_as_double((long long)(((uint64_t)rec[5 + 2 * (el - K::W)] << 32) | rec[4 + 2 * (el - K::W)]));
            };
            if constexpr (VAR) {
                if (!stepper) {
                    char* const stg = vstg[wv];
                    uint32_t* const wP = vplan[wv];
                    const size_t rw0 = (size_t)t0v * (size_t)a.n + (size_t)env0;
                    for (int tk = draw(); tk < nl; tk = draw()) {
                        const uint32_t* const rec = stage + tk * TSTR;
                        int ll = lq;
                        asm volatile("" : "+v"(ll));                 // the lane's constants of the row assembly are worked out per tick: kept
                                                                     // across the loop they are live in the steppers' code too (128 registers)
                        if (a.tail & SNAC_TAIL_PLAN) {
                            const int pq = (int)rec[min(ll, EB - 1) * RECW + 15];                  // lane e: env e's plan row at this tick
#pragma unroll
                            for (int e = 0; e < EB; ++e) {
                                const int pe = __builtin_amdgcn_readlane(pq, e);                   // wave-uniform: the row comes through the scalar cache
                                if (e < nenv && pe != ptag[e]) {
                                    cmem_u32* const src = (cmem_u32*)(uintptr_t)a.plans + (size_t)pe * GE;
                                    uint32_t rw[GE];
#pragma unroll
                                    for (int q = 0; q < GE; ++q) rw[q] = src[q];
                                    if (lq == 0) {
#pragma unroll
                                        for (int q = 0; q < GE; ++q) wP[e * GE + q] = rw[q];
                                    }
                                    ptag[e] = pe;
                                }
                            }
                        }
                        emit_rows_lean<OT, VSTG>(stg, rec, wP, ob + (size_t)tk * ostr, ll, nenv, a.ld, a.tail, a.frame_val);
                        if (lq < nenv) {                             // four envs per tick: small stores beside rows of kilobytes
                            if (a.reward) a.reward[rw0 + (size_t)tk * (size_t)a.n + lq] = sR[tk][lq];
                            if (a.done) a.done[rw0 + (size_t)tk * (size_t)a.n + lq] = sD[tk][lq];
                        }
                    }
                }
            } else if (vec) {
                const int pt = nenv * ROWB / 16;
                int tk = draw();
                while (tk < nl) {
                    int pend = draw_issue();                         // the draw after this one travels with the tick's reads
                    const uint32_t* const trow = stage + tk * TSTR;
                    char* const orun = ob + (size_t)tk * ostr + lq * 16;
                    uint32_t lo[NQ][VP], hi[NQ][VP];                 // every LDS read of the tick first: one round trip per tick, not per piece
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int v = 0; v < VP; ++v) {
                            const uint32_t* const sp = trow + (fsrc[q][v] & 0xffff);
                            lo[q][v] = sp[0]; hi[q][v] = sp[1];
                        }
                    asm volatile("" : "+v"(pend) :: "memory");
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        OT val[VP];
#pragma unroll
                        for (int v = 0; v < VP; ++v) {
                            // cell or scalar slot by a BIT select on a per-lane mask (v_bfi_b32): written as `kind ? a : b` the compiler
                            // keeps the kinds as exec masks in spilled SGPRs and spends nine scalar instructions and a branch per value
                            const int f = fsrc[q][v];
                            const uint32_t m = (uint32_t)-(f >> 24);             // all ones: a scalar slot
                            const int cv = __builtin_amdgcn_sbfe((int)lo[q][v], (uint32_t)(f >> 16) & 0xffu, 2u);   // v_bfe_i32: 0 / 1 / -1
                            if constexpr (sizeof(OT) == 8) {       // a cell's double has a zero low word: one AND, one v_bfi_b32
                                const uint32_t ch = (uint32_t)((uint64_t)__double_as_longlong((double)cv) >> 32);
                                const uint32_t rl = m & lo[q][v], rh = bfi32(m, hi[q][v], ch);
                                val[v] = (OT)__longlong_as_double((long long)(((uint64_t)rh << 32) | rl));
                            } else {
                                const float sf = (float)__longlong_as_double((long long)(((uint64_t)hi[q][v] << 32) | lo[q][v]));
                                val[v] = (OT)__int_as_float((int)bfi32(m, (uint32_t)__float_as_int(sf), (uint32_t)__float_as_int((float)cv)));
                            }
                        }
                        if (lq + 64 * q < pt) {
                            if constexpr (VP == 2) { double2 o; o.x = val[0]; o.y = val[1]; *(double2*)(orun + q * 1024) = o; }
                            else { float4 o; o.x = val[0]; o.y = val[1]; o.z = val[2]; o.w = val[3]; *(float4*)(orun + q * 1024) = o; }
                        }
                    }
                    tk = __builtin_amdgcn_readfirstlane(pend);
                }
            } else {
                const int pe = nenv * D;                             // ragged or unaligned: element by element, still in runs
                for (int tk = draw(); tk < nl; tk = draw())
                    for (int gel = lq; gel < pe; gel += 64) {
                        const int e = gel / D;
                        ((OT*)(ob + (size_t)tk * ostr))[gel] = value(stage + tk * TSTR + e * RECW, gel - e * D);
                    }
            }
            if (!VAR && !stepper) {
                // reward / done: 64 / EB ticks x EB envs per writer wave, one instruction each
                const size_t rw0 = (size_t)t0v * (size_t)a.n + (size_t)env0;
                const int tk = wq * TPW + lq / EB, e = lq & (EB - 1);
                const bool mine = tk < nl && e < nenv;
                if (a.reward && mine) a.reward[rw0 + (size_t)tk * (size_t)a.n + e] = sR[tk][e];
                if (dvec) {
                    const int wt = tid - EB * 64;                    // the writers' thread index
                    if (wt < nl) *(uint4*)(a.done + rw0 + (size_t)wt * (size_t)a.n) = *(const uint4*)sD[wt];
                } else if (a.done && mine) a.done[rw0 + (size_t)tk * (size_t)a.n + e] = sD[tk][e];
            }
        }
        __syncthreads();
    }
    // ---- the env's record
    if (!own) return;
    if (lane < GE) ((uint32_t*)a.grid)[(size_t)env * GE + lane] = G[lane];
    if (lane == 0) {
        s.r = r0; s.c = c0; s.cb = cb0; s.cs = cs0; s.ep_ret = ret0; s.tb = tb; s.pidx = pidx; s.cross = 0;
        s.flags = flag_done ? SNAC_FLAG_NEED_RESET : 0;
        a.hdr[env] = s.pack();
        a.episode[env] = episode;
        if (d_eps) {
            a.stat_episodes[env] += d_eps;
            a.stat_return[env] += d_ret;
            a.stat_iou_fx[env] += d_iou;
        }
    }
}

template <bool DYN, typename OT, int EB>
void launch_roll2dt_e(const KArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.n + EB - 1) / EB)), block(2 * EB * 64);   // EB stepper waves + EB writer waves
    if (a.actions || a.step_size) hipLaunchKernelGGL((k_rollout2dt<DYN, OT, EB, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_rollout2dt<DYN, OT, EB, false>), grid, block, 0, s, a);
}
template <bool DYN, typename OT>
void launch_roll2dt_var(const KArgs& a, hipStream_t s) {
    // 4 steppers and 12 writers per block: the rows are what takes the time (with 4 writers in blocks of 8 waves: 5.7 instead of 6.0 TB/s
    // at 1024 envs and half the rate at 256)
    constexpr int EB = 4, WR = 12;
    const dim3 grid((unsigned)((a.n + EB - 1) / EB)), block((EB + WR) * 64);
    if (a.actions || a.step_size) hipLaunchKernelGGL((k_rollout2dt<DYN, OT, EB, true, true, WR>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_rollout2dt<DYN, OT, EB, false, true, WR>), grid, block, 0, s, a);
}
template <bool DYN, typename OT>
void launch_roll2dt_w(const KArgs& a, hipStream_t s) {
    if (a.variant) { launch_roll2dt_var<DYN, OT>(a, s); return; }
    const int emin = snac_detail::tune(snac_detail::TN_2D_TP_EB8);
    if (a.n >= emin) launch_roll2dt_e<DYN, OT, 8>(a, s);       // 8 envs per block: runs of 3264 / 1632 bytes per tick
    else launch_roll2dt_e<DYN, OT, 4>(a, s);                        // up to 1024 envs: a block per CU first (1536 envs: 0.097 against 0.086 ms)
}

}  // namespace

namespace snac_detail {

void launch_roll2dt(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll2dt_w<true, float>(a, s) : launch_roll2dt_w<true, double>(a, s);
    else f32 ? launch_roll2dt_w<false, float>(a, s) : launch_roll2dt_w<false, double>(a, s);
}

}  // namespace snac_detail
