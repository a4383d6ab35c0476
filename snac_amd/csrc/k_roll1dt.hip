// k_roll1dt.hip -- k_rollout1dt: time-parallel 1D rollouts
#include "snac_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------
// 1D fused rollout, TIME-parallel (round 3).  Every other rollout kernel walks the ticks one after the other and is, for 1D, bound
// by that chain: BASELINE config 2 (N = 4096, T = 750) writes 187 MB -- 30 us of HBM time -- in 0.29 ms.  But in 1D
// (DMP_Env_1D_static.py:85-136) the whole control of an episode depends on the ACTIONS alone: count_step counts ticks, count_brick
// counts drops, the position is a chain of clamped additions, and done follows from the two counters.  So one wavefront takes ONE
// env and 64 consecutive ticks, lane j = tick t0 + j:
//   counters   count_step = ticks since the segment began; count_brick = drops so far: popcount of the drop ballot below the lane;
//              done = the first lane whose counters say so -- the lanes up to it form a segment (an episode's end splits a chunk:
//              the reset happens in the wave's uniform state and the rest of the chunk is a second segment);
//   position   x -> min(max(x + d, 2), 31) composed with itself is again x -> min(max(x + a, lo), hi): an inclusive scan over the
//              lanes (six shuffle steps) gives every tick's position at once;
//   heights    a drop at tick j lands on the cell under the agent.  Every dropping lane ORs its bit into that cell's 64-bit mask
//              in LDS (ds_or_b64); the height of a cell as tick j sees it = its height at the segment's start + popcount(mask of
//              the cell & lanes <= j): the five window cells and the reward's comparison are five LDS reads and popcounts;
//   the rest   rewards by ballot / popcount prefix sums, the two observation scalars by one division per lane, IoU and the
//              episodic sums by the lane that ends a segment, the cells' new heights (+ popcount of their masks) once per segment.
// ~4 instructions per env-step instead of ~14, and nothing waits for the tick before.  The price: a lane writes its own 56-byte
// row (rows of one env are N x 56 bytes apart); neighbouring envs' rows are neighbouring waves' stores and meet in L2.
// Semantics are K1D::step's; counter-RNG or explicit inputs; SNAC_OBS_ALL / SNAC_OBS_TILED, the canonical layout.
// data-parallel primitives: lanes without a source (or outside ROWS) receive `idv`.  0x110 + n: row_shr n; 0x142 / 0x143: lane 15 / 31
// of the rows before to the whole next row(s); 0x138: the wave shifted up by one lane

// VAR: the layout variants of snac_env_desc (frame value, raw / normalised counters, position / plan / record tails: rows of a.ld <= 46
// values) -- a lane files its whole row, tails included (the plan tail from the segment's plan in LDS), the staging tile is sized
// for rows of up to VLD = 16 / 38 / 46 values (blocks of 4 envs: 34 / 79 / 95 KB with float64 rows), the runs leave in as many
// 16-byte pieces as they have.
// N % 4 = 0 and a 16-byte aligned output.
template <bool DYN, typename OT, int EB, bool EXPL, int VLD = 0>
__global__ __launch_bounds__(EB * 64, VLD ? 1 : 16 / EB) void k_rollout1dt(const KArgs a) {      // 16 waves per CU either way: <= 128 VGPRs (layout variants: what LDS allows)
    using K = K1D<DYN, 8>;
    constexpr bool VAR = VLD != 0;                                   // VLD: the longest row the staging tile holds: 16 (L-Net, record), 38 (PPO), 46 values
    constexpr int D = K::D;
    constexpr int ROWB = D * (int)sizeof(OT);                        // 56 / 28 bytes per row
    constexpr int LDMAX = VAR ? VLD : D;
    constexpr int TSTR = EB * LDMAX * (int)sizeof(OT) + 16;          // staging bytes per tick (+16: the lanes' row writes spread over the banks)
    static_assert(!VAR || EB == 4, "layout variants: blocks of four envs");
    __shared__ int sH[EB][32], sP[EB][32];
    __shared__ unsigned long long sM[EB][32];
    __shared__ __align__(16) char stage[64 * TSTR];                  // [tick][env of the block][D]: what 64 ticks of the block's envs write
    __shared__ float sR[64][EB + 1];
    __shared__ __align__(16) uint8_t sD[64][EB];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int env0 = (int)blockIdx.x * EB;
    const int nenv = min(EB, a.n - env0);                            // block-uniform; > 0 by the grid
    const bool own = wv < nenv;                                      // waves past the batch only keep the barriers company
    const int env = env0 + (own ? wv : 0);
    int* const H = sH[wv];                                           // heights of the 30 interior cells as the current segment found them
    int* const P = sP[wv];                                           // the env's plan
    unsigned long long* const M = sM[wv];                            // per cell: the lanes that dropped a brick on it in this segment
    Lane s;
    s.unpack(a.hdr[env]);
    int episode = a.episode[env];
    asm volatile("" : "+v"(episode));
    if (lane < 32) {
        H[lane] = lane < 30 ? (int)((const int16_t*)a.grid)[(size_t)env * K::GE + lane] : 0;
        P[lane] = lane < 30 ? (int)((const int16_t*)a.plans)[(size_t)s.pidx * K::GE + lane] : 0;
    }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const EnvKeys sk = env_keys(a.key_step, gid), pk = env_keys(a.key_plan, gid);
    // wave-uniform env state (every lane holds the same values)
    int pos0 = s.r, cb0 = s.cb, cs0 = s.cs, ret0 = s.ep_ret, tb = s.tb, pidx = s.pidx;
    asm volatile("" : "+v"(tb));                                     // the header has arrived HERE: no vector-memory wait inside the loop,
                                                                     // where it would also wait for the chunk before's stores
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
    const int tstr = VAR ? EB * RB + 16 : TSTR;                      // bytes per tick of the staging tile
    const size_t ostr = (tl ? (size_t)64 : (size_t)a.n) * RB;        // bytes from one tick's run to the next
    const bool vec = ((((uintptr_t)a.obs) | (uintptr_t)ostr | (uintptr_t)((size_t)nenv * RB)) & 15) == 0;
    const bool dvec = EB == 16 && a.done && nenv == EB && ((((uintptr_t)a.done) | (uintptr_t)a.n) & 15) == 0;
    for (int t0 = 0; t0 < a.T; t0 += 64) {
        const int nl = min(64, a.T - t0);
        if (own) {
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
            if (need_reset) {                                        // K1D::reset in the uniform state (rare: once per episode)
                episode += 1;
                const int np = pick_plan<K>(a, pk, episode, pidx);
                if (np != pidx) {
                    pidx = np; tb = (int)a.plan_tb[np];
                    asm volatile("" : "+v"(tb));
                    dtb = (double)tb; rtb = 1.0 / dtb;
                    if (lane < 32) P[lane] = lane < 30 ? (int)((const int16_t*)a.plans)[(size_t)np * K::GE + lane] : 0;
                }
                if (lane < 32) H[lane] = 0;
                pos0 = 2; cb0 = 0; cs0 = 0; ret0 = 0;
                need_reset = false;
            }
            const bool seg = valid && lane >= first_lane;
            const bool drop = seg && act == 2;
            // ---- counters and the segment's end
            const int cs = min(cs0 + (lane - first_lane + 1), CNT_MAX);
            const unsigned long long dropm = __ballot(drop);
            const int cb = min(cb0 + (int)__popcll(dropm & le), CNT_MAX);
            const bool term = term_rule(drop, cb, tb, a.brick_gt);   // :107-114, before the time limit (the rules' pieces: snac_dev.h)
            const bool done = seg && (term || cs >= a.ts_done);
            const unsigned long long donem = __ballot(done);
            const int last = donem ? (__ffsll((long long)donem) - 1) : (nl - 1);     // the segment's last lane
            const bool in = seg && lane <= last;
            // ---- positions: inclusive scan of x -> min(max(x + d, 2), 31)
            // the clamp pair (lo, hi) travels as two int16 in ONE register: a compose step is three packed instructions (v_pk_add / max / min_i16,
            // the scalar operand's half picked by op_sel) instead of six (round 6: 66 -> 36 vector instructions per segment; positions are
            // 2 .. 31, a chunk moves at most 192 cells, the identity is (-4096, 4096): everything fits 16 bits)
            int sa = 0;
            uint32_t lh = 0x1000F000u;                               // (lo, hi) = (-4096, 4096): the identity
            if (in) { sa = act == 0 ? -k : (act == 1 ? k : 0); lh = (31u << 16) | 2u; }
            auto compose = [&](int pa, uint32_t plh) {               // the earlier ticks first, then this lane's function
                uint32_t t;
                asm("v_pk_add_i16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(plh), "v"(sa));          // (plo + sa, phi + sa)
                asm("v_pk_max_i16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(t), "v"(lh));            // max(., slo)
                asm("v_pk_min_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "v"(t), "v"(lh));   // min(., shi)
                sa += pa; lh = t;
            };
            // Hillis-Steele inside the rows of 16 lanes (row_shr 1, 2, 4, 8), then the rows' last lanes to the rows behind them; a
            // lane without a source composes with the identity, so no step is conditional
            compose(dpp_from<0x111>(0, sa), (uint32_t)dpp_from<0x111>((int)0x1000F000u, (int)lh));
            compose(dpp_from<0x112>(0, sa), (uint32_t)dpp_from<0x112>((int)0x1000F000u, (int)lh));
            compose(dpp_from<0x114>(0, sa), (uint32_t)dpp_from<0x114>((int)0x1000F000u, (int)lh));
            compose(dpp_from<0x118>(0, sa), (uint32_t)dpp_from<0x118>((int)0x1000F000u, (int)lh));
            compose(dpp_from<0x142, 0xa>(0, sa), (uint32_t)dpp_from<0x142, 0xa>((int)0x1000F000u, (int)lh));
            compose(dpp_from<0x143, 0xc>(0, sa), (uint32_t)dpp_from<0x143, 0xc>((int)0x1000F000u, (int)lh));
            const int slo = (int)(int16_t)(lh & 0xFFFFu), shi = (int)(int16_t)(lh >> 16);
            const int pos = min(max(pos0 + sa, slo), shi);           // after the tick
            const int prev = dpp_from<0x138>(pos0, pos);
            const int posb = lane == first_lane ? pos0 : prev;       // before the tick: where a drop lands
            // ---- the drops as per-cell lane masks
            if (lane < 32) M[lane] = 0ull;
            if (in && drop) atomicOr(&M[posb - 2], 1ull << lane);
            // ---- the window round the new position as tick `lane` leaves it
            int win[K::W];
#pragma unroll
            for (int i = 0; i < K::W; ++i) {
                const int ci = pos - 4 + i;                          // interior cell index: -2 .. 31
                const int cc = min(max(ci, 0), 31);
                const int h = min(H[cc] + (int)__popcll(M[cc] & le), CNT_MAX);
                win[i] = (ci < 0 || ci > 29) ? -1 : h;
            }
            const int hnew = win[2];                                 // a drop does not move: the agent's cell after the brick
            const int pl = P[min(max(posb - 2, 0), 31)];
            const int reward = reward1d(drop, term, hnew, pl);       // :117-123
            // running return: rewards are -1 / 1 / 10, three ballots
            const unsigned long long inm = __ballot(in);
            const unsigned long long r10 = __ballot(in && reward == 10), r1 = __ballot(in && reward == 1), rm = __ballot(in && reward == -1);
            const int ret = clamp16(ret0 + 10 * (int)__popcll(r10 & le) + (int)__popcll(r1 & le) - (int)__popcll(rm & le));
            // ---- outputs of the segment's lanes: into the block's staging tile
            if (in) {
                const double c0 = (double)cb, c1 = (double)cs;
                double v0 = c0, v1 = c1;
                if (VAR ? (a.sc_norm != 0) : DYN) {                  // cb / tb, cs / T: correctly rounded (Roll3D, tests/native/recip_check.c)
                    const double q0 = c0 * rtb, q1 = c1 * rT;
                    v0 = tb > 0 ? __builtin_fma(__builtin_fma(-q0, dtb, c0), rtb, q0) : c0 / dtb;
                    v1 = __builtin_fma(__builtin_fma(-q1, dT, c1), rT, q1);
                }
                OT* const o = (OT*)(stage + lane * tstr + wv * RB);
#pragma unroll
                for (int i = 0; i < K::W; ++i) o[i] = (OT)(double)((VAR && win[i] < 0) ? a.frame_val : win[i]);
                o[K::W] = (OT)v0; o[K::W + 1] = (OT)v1;
                if constexpr (VAR) {                                 // the tails, in the descriptor's order
                    OT* q = o + D;
                    if (a.tail & SNAC_TAIL_POSITION) { q[0] = (OT)(double)pos; q += 1; }
                    if (a.tail & SNAC_TAIL_PLAN) {
#pragma unroll
                        for (int c = 0; c < 30; ++c) q[c] = (OT)(double)P[c];
                        q += 30;
                    }
                    if (a.tail & SNAC_TAIL_RECORD) {
                        const int rv[8] = {reward, (lane == last && donem) ? 1 : 0, pos, 0, cb, cs, tb, pidx};   // record_value
#pragma unroll
                        for (int j = 0; j < 8; ++j) q[j] = (OT)(double)rv[j];
                    }
                }
                sR[lane][wv] = (float)reward;
                sD[lane][wv] = (lane == last && donem) ? 1 : 0;
                if (a.actions_out) a.actions_out[row] = (int8_t)act;
                if (a.step_size_out) a.step_size_out[row] = (int8_t)k;
                if (a.plan_idx_out) a.plan_idx_out[row] = (int16_t)pidx;
                if (a.first_out) a.first_out[row] = cs == 1 ? 1 : 0;
            }
            // ---- the segment's end: the cells take their bricks, the uniform state moves on
            if (lane < 32) H[lane] = min(H[lane] + (int)__popcll(M[lane] & inm), CNT_MAX);
            pos0 = __builtin_amdgcn_readlane(pos, last); cb0 = __builtin_amdgcn_readlane(cb, last);     // `last` is uniform
            cs0 = __builtin_amdgcn_readlane(cs, last); ret0 = __builtin_amdgcn_readlane(ret, last);
            flag_done = donem != 0ull;
            if (donem) {                                             // iou :138-151 of the finished episode, episodic sums
                asm volatile("" ::: "memory");                       // once per episode: stays a branch (18 cross-lane steps otherwise run every segment)
                int g = lane < 30 ? H[lane] : 0, pp = lane < 30 ? P[lane] : 0, over = max(g - pp, 0);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { g += __shfl_xor(g, off); pp += __shfl_xor(pp, off); over += __shfl_xor(over, off); }
                const int cross = g - over;
                const double v = (double)cross / (double)(pp + g - cross);
                d_eps += 1; d_ret += ret0; d_iou += __double2ll_rn(v * FX40);
                need_reset = a.auto_reset != 0;
            }
            first_lane = last + 1;
        }
        }
        __syncthreads();
        // ---- the tile leaves: per tick one run of the block's rows, the threads of the block across the runs
        {
            int t0v = t0, wq = wv, lq = lane;
            asm volatile("" : "+s"(t0v), "+v"(wq), "+v"(lq));        // addresses from scratch every chunk: a dozen running 64-bit pointers
                                                                     // and offsets kept across the loop cost more registers than there are
            const size_t row0 = tl ? ((size_t)(env0 >> 6) * (size_t)a.tiled_T + (size_t)(a.tiled_t0 + t0v)) * 64 + (size_t)(env0 & 63)
                                   : (size_t)t0v * (size_t)a.n + (size_t)env0;
            char* const ob = (char*)a.obs + row0 * RB;
            constexpr int TPW = 64 / EB;                             // ticks per wave
            if constexpr (VAR) {
                const int pt = nenv * RB / 16;                       // 16-byte pieces of a tick's run (the dispatch sees to whole pieces)
                for (int i = 0; i < TPW; ++i) {
                    const int tk = wq * TPW + i;
                    if (tk < nl)
                        for (int pc = lq; pc < pt; pc += 64) *(uint4*)(ob + (size_t)tk * ostr + pc * 16) = *(const uint4*)(stage + tk * tstr + pc * 16);
                }
            } else if (vec) {
                // 16-byte pieces: a tick's run has pt <= LPT of them, LPT lanes per tick, 64 / LPT ticks per store instruction
                constexpr int PTMAX = EB * ROWB / 16, LPT = PTMAX > 32 ? 64 : (PTMAX > 16 ? 32 : (PTMAX > 8 ? 16 : 8)), TPI = 64 / LPT;
                const int pt = nenv * ROWB / 16, pc = lq & (LPT - 1);
#pragma unroll
                for (int i = 0; i < TPW / TPI; ++i) {
                    const int tk = wq * TPW + i * TPI + lq / LPT;
                    if (pc < pt && tk < nl) *(uint4*)(ob + (size_t)tk * ostr + pc * 16) = *(const uint4*)(stage + tk * TSTR + pc * 16);
                }
            } else {
                const int pe = nenv * D;                             // ragged or unaligned: element by element, still in runs
                for (int i = 0; i < TPW; ++i) {
                    const int tk = wq * TPW + i;
                    if (tk < nl)
                        for (int el = lq; el < pe; el += 64) ((OT*)(ob + (size_t)tk * ostr))[el] = ((const OT*)(stage + tk * TSTR))[el];
                }
            }
            // reward / done: 64 / EB ticks x EB envs per wave, one instruction each
            const size_t r0 = (size_t)t0v * (size_t)a.n + (size_t)env0;
            const int tk = wq * TPW + lq / EB, e = lq & (EB - 1);
            const bool mine = tk < nl && e < nenv;
            if (a.reward && mine) a.reward[r0 + (size_t)tk * (size_t)a.n + e] = sR[tk][e];
            if (dvec) {
                if (tid < nl) *(uint4*)(a.done + r0 + (size_t)tid * (size_t)a.n) = *(const uint4*)sD[tid];
            } else if (a.done && mine) a.done[r0 + (size_t)tk * (size_t)a.n + e] = sD[tk][e];
        }
        __syncthreads();
    }
    // ---- the env's record
    if (!own) return;
    if (lane < 32) ((int16_t*)a.grid)[(size_t)env * K::GE + lane] = lane < 30 ? (int16_t)H[lane] : (int16_t)0;
    if (lane == 0) {
        s.r = pos0; s.c = 0; s.cb = cb0; s.cs = cs0; s.ep_ret = ret0; s.tb = tb; s.pidx = pidx; s.cross = 0;
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
void launch_roll1dt_e(const KArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.n + EB - 1) / EB)), block(EB * 64);
    if (a.actions || a.step_size) hipLaunchKernelGGL((k_rollout1dt<DYN, OT, EB, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_rollout1dt<DYN, OT, EB, false>), grid, block, 0, s, a);
}
template <bool DYN, typename OT>
void launch_roll1dt_w(const KArgs& a, hipStream_t s) {
    if (a.variant) {
        const dim3 grid((unsigned)((a.n + 3) / 4)), block(256);
        const bool expl = a.actions || a.step_size;
        if (a.ld <= 16) {                                            // the smaller the staging tile, the more blocks share a CU
            if (expl) hipLaunchKernelGGL((k_rollout1dt<DYN, OT, 4, true, 16>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_rollout1dt<DYN, OT, 4, false, 16>), grid, block, 0, s, a);
        } else if (a.ld <= 38) {
            if (expl) hipLaunchKernelGGL((k_rollout1dt<DYN, OT, 4, true, 38>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_rollout1dt<DYN, OT, 4, false, 38>), grid, block, 0, s, a);
        } else {
            if (expl) hipLaunchKernelGGL((k_rollout1dt<DYN, OT, 4, true, 46>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_rollout1dt<DYN, OT, 4, false, 46>), grid, block, 0, s, a);
        }
        return;
    }
    const int emin = snac_detail::tune(snac_detail::TN_1D_TP_EB16), e8lo = snac_detail::tune(snac_detail::TN_1D_TP_EB8_MIN), e8hi = snac_detail::tune(snac_detail::TN_1D_TP_EB8_MAX);
    if (a.n >= e8lo && a.n <= e8hi) launch_roll1dt_e<DYN, OT, 8>(a, s);   // 8 envs per block: two blocks share a CU where 16-env blocks number one per CU
    else if (a.n >= emin) launch_roll1dt_e<DYN, OT, 16>(a, s); // 16 envs per block: runs of 896 / 448 bytes per tick (3072 envs: 0.048 against 0.041 ms; 3584: level)
    else launch_roll1dt_e<DYN, OT, 4>(a, s);                        // small batches: more blocks than CUs first
}

}  // namespace

namespace snac_detail {

void launch_roll1dt(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll1dt_w<true, float>(a, s) : launch_roll1dt_w<true, double>(a, s);
    else f32 ? launch_roll1dt_w<false, float>(a, s) : launch_roll1dt_w<false, double>(a, s);
}

}  // namespace snac_detail
