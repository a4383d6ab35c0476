// snac_hip.hip -- gfx950 (MI355X) kernels and the C ABI of include/snac_hip.h (trajectory memory: snac_traj.hip).
//
// Integer / indexing work only -- no MFMA; the bound is HBM (observation rows written) or, for small batches, the chain of ticks.
// DESIGN.md section 3 has the table of kernels with their measured times, launch() at the end of this file the dispatch.  In short:
//   k_rollout2d    the headline: 2D rollouts on tiles of 64 envs, lane = env in the transition AND in the observation rows, which are
//                  transposed through an LDS staging tile and leave 16 bytes per lane (emit_tile; layout variants: emit_rows_var);
//                  plan rows per wave in LDS, refilled through the scalar cache; no vector load in the loop (vmcnt retires in order);
//   k_rollout2dt   2D rollouts of small and middle batches, time-parallel: one wave per env, lane = tick; stepper and writer waves per
//                  block, the ticks to expand a queue between them; <.., VAR>: the layout variants (PPO rows of 451 values ...), rows
//                  assembled by twelve writer waves (emit_rows_lean);
//   k_rollout1dt   the same idea for 1D (counters by ballots, positions by DPP scans, heights by per-cell lane masks); <.., VLD>: its
//                  layout variants;
//   k_rollout3db   3D rollouts: one stepper wave (lane = env) and eight writer waves per 64 envs, one barrier per tick;
//   k_rollout3d    3D rollouts of small / odd batches: 8 envs per wave, software-pipelined round the store stream;
//   k_step2d / 3d  snac_step on identity rows: wide loads, rows through emit_tile;  k_edges3d: 3D tree edges, records through LDS;
//   k_transition2d / 3d, and the tile kernels k_rollout / k_transition / k_aux (rounds 1-2) behind all of them for everything else.
// The env records are read from HBM once per launch, kept on chip for all T steps, written back once.  Rollout outputs are [T][N][D]
// or, with SNAC_OBS_TILED, tile-major [N / 64][T][64][D].
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "snac_hip.h"

#include "snac_common.h"

namespace snac_detail {
thread_local char g_err[256] = "";
}
using snac_detail::fail;
using snac_detail::fail_hip;
using snac_detail::g_err;

namespace {

// ------------------------------------------------------------------------------------------------
// counter RNG (include/snac_hip.h)
__host__ __device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
inline uint32_t stream_key(uint64_t seed, uint32_t stream) {
    return mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9E3779B9u * (stream + 1u)));
}
struct EnvKeys { uint32_t e0, e1; };
__device__ inline EnvKeys env_keys(uint32_t key, uint64_t env) {
    const uint32_t elo = (uint32_t)env, ehi = (uint32_t)(env >> 32);
    EnvKeys k;
    k.e0 = mix32(key ^ mix32(elo + 0x85EBCA6Bu * ehi + 0x1B873593u));
    k.e1 = mix32((key + 0x27D4EB2Fu) ^ mix32((elo ^ 0x165667B1u) + 0xC2B2AE35u * ehi));
    return k;
}
__device__ inline uint32_t rng_word(EnvKeys k, uint32_t t) { return mix32(mix32(k.e0 ^ (0x9E3779B9u * t)) + k.e1); }

// ------------------------------------------------------------------------------------------------
struct KArgs {
    int32_t n, num_plans, static_plan, T, auto_reset, obs_mode;
    int32_t tiled_T, tiled_t0;  // SNAC_OBS_TILED: steps per tile region of the target ([..][tiled_T][64][LD]) and this launch's first step in it
    int32_t total_step;        // the env's time limit (snac_env_desc.total_step or the kind's default)
    int32_t ts_done;           // count_step >= ts_done ends the episode: total_step (+ 1 with SNAC_RULE_TIME_GT)
    int32_t brick_gt;          // 1: count_brick > total_brick ends the episode (SNAC_RULE_BRICK_GT), 0: >=
    uint32_t t0, key_step, key_plan;
    int64_t env_id_base;
    int4* hdr;                 // snac_env_hdr[N] as 16-byte words
    int32_t* episode;
    void* grid;
    const void* plans;
    const int16_t* plan_tb;
    int64_t* stat_episodes;
    int64_t* stat_return;
    int64_t* stat_iou_fx;
    const int8_t* actions;
    const int8_t* step_size;
    void* obs;
    float* reward;
    uint8_t* done;
    // optional per-step record (snac_rollout_rec): what a replay memory needs besides obs / reward / done
    int8_t* actions_out;
    int8_t* step_size_out;
    int16_t* plan_idx_out;
    uint8_t* first_out;
    // snac_transition only: the state arrays are a node pool of `pool` rows; n = number of transitions
    int32_t pool;
    int32_t stats_on;          // single-step kernel: update the episodic sums (snac_step) or not (snac_transition)
    const int32_t* src_index;  // row read by transition i (NULL: i)
    const int32_t* dst_index;  // row written by transition i (NULL: i)
    // aux kernel only
    int32_t aux_op;            // AUX_*
    const uint8_t* mask;
    const int16_t* plan_idx_in;
    double* out_f64;
    int32_t plan_scalar;       // snac_reset_scalar: plan row of every env (-1: unused)
    // observation-layout variants (snac_env_desc.frame_value / obs_scalars / obs_tail); variant != 0 selects the VAR kernels
    int32_t variant;
    int32_t ld;                // values per observation row: K::D + tail
    int32_t frame_val;         // value shown for frame cells
    int32_t sc_norm;           // 1: count_brick / total_brick, count_step / total_step; 0: raw counters
    int32_t tail;              // SNAC_TAIL_* bits
    // snac_step_scalar: one action / step size for every env, by value
    int32_t use_scalar, act_scalar, k_scalar;
};
enum { AUX_RESET = 0, AUX_OBSERVE = 1, AUX_IOU = 2 };

constexpr double FX40 = 1099511627776.0;  // 2^40

// state row of tile element i: identity for the env batch, a clamped gather / scatter index for snac_transition
__device__ __forceinline__ size_t row_of(const int32_t* idx, int pool, int i) {
    return idx ? (size_t)min(max(idx[i], 0), pool - 1) : (size_t)i;
}
// the same inside the cooperative tile loops: the rows of the tile's elements are staged in LDS once (rows[e]) instead of
// one more global load per element
__device__ __forceinline__ size_t tile_row(const int* rows, int env0, int e) {
    return rows ? (size_t)rows[e] : (size_t)(env0 + e);
}

// The header packs its counters as int16.  The reference never resets by itself and "keeps mutating" when stepped past
// done (Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:86: count_step is unbounded); here count_step,
// count_brick and the heights SATURATE at 32767 and the running return in [-32768, 32767] instead of wrapping.  Termination
// is unaffected (total_step <= 3000 and total_brick <= 32767 are reached long before), only the counters an observation
// shows stop growing.
constexpr int CNT_MAX = 32767;
__device__ __forceinline__ int clamp16(int v) { return min(max(v, -32768), 32767); }
// episodic sums by no-return atomics (nothing waits for them)
__device__ __forceinline__ void stat_add(int64_t* p, long long v) {
    (void)__hip_atomic_fetch_add((unsigned long long*)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// per-lane env scalars (one env per lane in phase 1)
struct Lane {
    int r, c, flags, cb, cs, tb, pidx, ep_ret, cross;
    __device__ void unpack(const int4 h) {
        r = (int)(int8_t)(h.x & 0xff); c = (int)(int8_t)((h.x >> 8) & 0xff); flags = (h.x >> 16) & 0xff;
        cb = (int)(int16_t)(h.y & 0xffff); cs = h.y >> 16;
        tb = (int)(int16_t)(h.z & 0xffff); pidx = h.z >> 16;
        ep_ret = (int)(int16_t)(h.w & 0xffff); cross = h.w >> 16;
    }
    __device__ int4 pack() const {
        int4 h;
        h.x = (r & 0xff) | ((c & 0xff) << 8) | ((flags & 0xff) << 16);
        h.y = (cb & 0xffff) | (cs << 16);
        h.z = (tb & 0xffff) | (pidx << 16);
        h.w = (ep_ret & 0xffff) | (cross << 16);
        return h;
    }
    __device__ void clear() { r = c = flags = cb = cs = tb = pidx = ep_ret = cross = 0; }
};

// ================================================================================================
// LDS images.  Every kind keeps the env's grid WITH its frame in LDS, so that neither the transition nor the
// observation window needs a bounds test: a window cell is one LDS read at (uniform base + lane constant).
// The HBM records stay compact (interior only); the frame is re-created when a tile is loaded.

// bit j of x (j < 16) -> bit 2j
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    return x;
}
// bit 2j of x -> bit j
__device__ __forceinline__ uint32_t squeeze16(uint32_t x) {
    x &= 0x55555555u; x = (x | (x >> 1)) & 0x33333333u; x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu; x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}

// ================================================================================================
// 2D: Env/2D/DMP_Env_2D_static.py, Env/2D/DMP_Env_2D_dynamic_usedata_plan.py
// LDS per wave: C[(row * RS + e)] 64-bit words, row = bordered row 0..25, RS = E + 1 (odd stride: the 7 rows of a
// window fall in different banks).  Cell k (bordered column 0..25) is the 2-bit field at bit 2k: 00 empty, 01 brick,
// 11 frame -- a signed 2-bit extract yields the reference's cell value 0 / 1 / -1 directly.  Then P[row * RS + e]:
// the env's plan rows as 1-bit boards (the step loop must not issue global loads: vmcnt is in-order, a load would
// wait for every observation store before it).
template <bool DYN_, int E_>
struct K2D {
    static constexpr bool DYN = DYN_;
    static constexpr int E = E_, D = 51, W = 49, A = 5, TS = 600, GE = 20, RS = E + 1;
    static constexpr int P_OFF = 52 * RS;                            // dwords
    static constexpr int SC_OFF = 72 * RS + ((72 * RS) & 1);
    static constexpr int LDS_WORDS = SC_OFF + 4 * E;
    __device__ static double* sc(uint32_t* lds) { return (double*)(lds + SC_OFF); }
    static constexpr uint32_t ROW_LO = 0x3Fu, ROW_HI = 0xFC000u;     // frame cells 0-2 and 23-25 of an interior row

    __device__ static uint64_t* cells(uint32_t* lds) { return (uint64_t*)lds; }
    __device__ static uint64_t encode_row(uint32_t bits) {           // 20 interior bits -> 26 two-bit cells
        const uint32_t lo = ROW_LO | (spread16(bits & 0x1FFFu) << 6), hi = ROW_HI | spread16(bits >> 13);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ static uint32_t decode_row(uint64_t w) {
        const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
        return squeeze16((lo >> 6) & 0x01555555u) | (squeeze16(hi & 0x1555u) << 13);
    }
    __device__ static void load_grid(uint32_t* lds, const KArgs& a, int env0, int nenv, int lane, const int* rows = nullptr) {
        uint64_t* c = cells(lds);
        for (int i = lane; i < 3 * RS; i += 64) { c[i] = 0x000FFFFFFFFFFFFFull; c[23 * RS + i] = 0x000FFFFFFFFFFFFFull; }
        const uint32_t* src = (const uint32_t*)a.grid;
        for (int i = lane; i < nenv * GE; i += 64) {
            const int e = i / GE, row = i - e * GE;
            c[(row + 3) * RS + e] = encode_row(src[tile_row(rows, env0, e) * GE + row]);
        }
    }
    __device__ static void store_grid(uint32_t* lds, const KArgs& a, int env0, int nenv, int lane, const int* rows = nullptr) {
        const uint64_t* c = cells(lds);
        uint32_t* dst = (uint32_t*)a.grid;
        for (int i = lane; i < nenv * GE; i += 64) {
            const int e = i / GE, row = i - e * GE;
            dst[tile_row(rows, env0, e) * GE + row] = decode_row(c[(row + 3) * RS + e]);
        }
    }
    __device__ static void load_plan(uint32_t* lds, const KArgs& a, int e, int pidx, int lane) {  // whole wave
        if (lane < GE) lds[P_OFF + lane * RS + e] = ((const uint32_t*)a.plans)[pidx * GE + lane];
    }
    // the one plan word a single step() can read: the agent's row (fetched early, placed once the tile is loaded)
    struct PlanCell { uint32_t v; };
    __device__ static PlanCell fetch_plan_cell(const KArgs& a, const Lane& s) {
        return PlanCell{((const uint32_t*)a.plans)[s.pidx * GE + (s.r - 3)]};
    }
    __device__ static void put_plan_cell(uint32_t* lds, const Lane& s, const PlanCell& pc, int lane) {
        lds[P_OFF + (s.r - 3) * RS + lane] = pc.v;
    }
    // reset: DMP_Env_2D_dynamic_usedata_plan.py:34-66 (the total_brick floor of 30 is folded into plan_tb)
    __device__ static void reset(const KArgs& a, Lane& s, int pidx) {
        if (pidx >= 0) { s.pidx = pidx; s.tb = a.plan_tb[pidx]; }    // pidx < 0: the env keeps its plan and total_brick
        s.r = 3; s.c = 3; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.cross = 0; s.flags = 0;
    }
    __device__ static void clear(uint32_t* lds, int e, int lane) {   // whole wave empties env e's interior
        if (lane < GE) cells(lds)[(lane + 3) * RS + e] = ((uint64_t)ROW_HI << 32) | ROW_LO;
    }
    // step: DMP_Env_2D_dynamic_usedata_plan.py:85-147
    __device__ static void step(uint32_t* lds, const KArgs& a, Lane& s, int act, int k, int ts, int bg, int lane, int& reward, bool& done) {
        uint64_t* cw = cells(lds) + s.r * RS + lane;
        const uint64_t w = *cw;
        const int off = 2 * s.c;
        const bool was = ((w >> off) & 1ull) != 0ull;
        const bool planned = ((lds[P_OFF + (s.r - 3) * RS + lane] >> (s.c - 3)) & 1u) != 0u;
        const bool drop = act == 4;
        s.cs = min(s.cs + 1, CNT_MAX);
        if (drop) {
            s.cb = min(s.cb + 1, CNT_MAX);
            *cw = w | (1ull << off);                                 // += 1 then clamp to 1 (:115, :134-135)
        }
        if (act == 0) s.c = max(s.c - k, 3);                         // clip_position :74-83
        if (act == 1) s.c = min(s.c + k, 22);
        if (act == 2) s.r = min(s.r + k, 22);                        // "up" is row + k (:100-103)
        if (act == 3) s.r = max(s.r - k, 3);
        const bool term = drop && s.cb >= s.tb + bg;                 // :117-126, tested before the time limit (bg: SNAC_RULE_BRICK_GT)
        done = term || s.cs >= ts;
        // un-clamped cell vs plan (:129-133): 5 iff the cell was empty and is planned
        reward = (drop && !term && !was && planned) ? 5 : 0;
    }
    // boolean IoU: script/DQN/2d/DQN_2d_dynamic.py:63-71
    __device__ static double iou(uint32_t* lds, const Lane& s, int lane) {
        int inter = 0, uni = 0;
        for (int row = 0; row < GE; ++row) {
            const uint32_t g = decode_row(cells(lds)[(row + 3) * RS + lane]);
            const uint32_t p = lds[P_OFF + row * RS + lane];
            inter += __popc(g & p); uni += __popc(g | p);
        }
        return (double)inter / (double)uni;
    }
    // phase-2 keys of the lane's env: byte offset of the window's first row, bit offset of its first column
    __device__ static int key0(const Lane& s) { return (s.r - 3) * RS * 8; }
    __device__ static int key1(const Lane& s) { return 2 * (s.c - 3); }
    // SNAC_TAIL_PLAN: input_plan cell (row-major 20x20) of plan row pidx, from the L2-resident table
    static constexpr int PLAN_CELLS = 400;
    __device__ static int plan_value(const KArgs& a, int pidx, int cell) {
        const int row = cell / 20, col = cell - row * 20;
        return (int)((((const uint32_t*)a.plans)[pidx * GE + row] >> col) & 1u);
    }
};

// ================================================================================================
// 3D: Env/3D/DMP_simulator_3d_static_circle.py, Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py
// LDS per wave: H[e * ES + r * 26 + c] int16, the bordered 26x26 height map (frame = -1), ES = 678 (odd dword
// stride); PL[e * PS + cell] the env's plan (20x20 interior), PS = 402.
template <bool DYN_, int E_>
struct K3D {
    static constexpr bool DYN = DYN_;
    static constexpr int E = E_, D = 51, W = 49, A = 8, TS = DYN_ ? 1000 : 1300, GE = 400, ES = 678;
    static constexpr int SC_OFF = E * ES / 2 + ((E * ES / 2) & 1);   // dwords
    static constexpr int LDS_WORDS = SC_OFF + 4 * E;
    __device__ static double* sc(uint32_t* lds) { return (double*)(lds + SC_OFF); }

    __device__ static int16_t* hmap(uint32_t* lds) { return (int16_t*)lds; }

    __device__ static void load_grid(uint32_t* lds, const KArgs& a, int env0, int nenv, int lane, const int* rows = nullptr) {
        for (int i = lane; i < E * ES / 2; i += 64) lds[i] = 0xFFFFFFFFu;            // everything frame (-1) ...
        const int16_t* src = (const int16_t*)a.grid;
        int16_t* h = hmap(lds);
        for (int i = lane; i < nenv * GE; i += 64) {                                  // ... then the interiors
            const int e = i / GE, cell = i - e * GE, r = cell / 20, c = cell - r * 20;
            h[e * ES + (r + 3) * 26 + c + 3] = src[tile_row(rows, env0, e) * GE + cell];
        }
    }
    __device__ static void store_grid(uint32_t* lds, const KArgs& a, int env0, int nenv, int lane, const int* rows = nullptr) {
        int16_t* dst = (int16_t*)a.grid;
        const int16_t* h = hmap(lds);
        for (int i = lane; i < nenv * GE; i += 64) {
            const int e = i / GE, cell = i - e * GE, r = cell / 20, c = cell - r * 20;
            dst[tile_row(rows, env0, e) * GE + cell] = h[e * ES + (r + 3) * 26 + c + 3];
        }
    }
    // The plan is NOT staged: a step needs at most one plan cell (the build target), fetched from the L2-resident table
    // inside step().  Leaving the 800-byte plan out of LDS is what lets 14 waves (instead of 9) share a CU.
    __device__ static void load_plan(uint32_t*, const KArgs&, int, int, int) {}
    struct PlanCell {};
    __device__ static PlanCell fetch_plan_cell(const KArgs&, const Lane&) { return PlanCell{}; }
    __device__ static void put_plan_cell(uint32_t*, const Lane&, const PlanCell&, int) {}
    // reset: DMP_simulator_3d_dynamic_triangle_usedata.py:45-75
    __device__ static void reset(const KArgs& a, Lane& s, int pidx) {
        if (pidx >= 0) { s.pidx = pidx; s.tb = a.plan_tb[pidx]; }    // pidx < 0: the env keeps its plan and total_brick
        s.r = 3; s.c = 3; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.cross = 0; s.flags = 0;
    }
    __device__ static void clear(uint32_t* lds, int e, int lane) {   // whole wave zeroes env e's interior
        int16_t* h = hmap(lds) + e * ES;
#pragma unroll
        for (int i = lane; i < GE; i += 64) { const int r = i / 20, c = i - r * 20; h[(r + 3) * 26 + c + 3] = 0; }
    }
    // step: DMP_simulator_3d_static_circle.py:153-230, DMP_simulator_3d_dynamic_triangle_usedata.py:142-231
    __device__ static void step(uint32_t* lds, const KArgs& a, Lane& s, int act, int k, int ts, int bg, int lane, int& reward, bool& done) {
        int16_t* h = hmap(lds) + lane * ES + s.r * 26 + s.c;         // the agent's cell
        s.cs = min(s.cs + 1, CNT_MAX);
        reward = 0;
        // check_sur (:88-102 / :77-91): left, right, "up" (row + 1), "down" (row - 1)
        const int n0 = h[-1], n1 = h[1], n2 = h[26], n3 = h[-26];
        const bool boxed_pre = n0 != 0 && n1 != 0 && n2 != 0 && n3 != 0;
        done = (s.cs >= ts) || (!DYN && boxed_pre);                  // bottom of step(): static :226, dynamic :226
        const int d = act & 3;
        const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
        const int dl = dr * 26 + dc;
        const int nd = d == 0 ? n0 : (d == 1 ? n1 : (d == 2 ? n2 : n3));
        const int c2 = h[2 * dl], c3 = h[3 * dl];                    // within the frame: |offset| <= 3 cells
        const bool valid = (unsigned)act < 8u;
        if (valid && act < 4) {
            if (nd == 0) {                                           // check[act] == 0
                // move_step (:104-134): consecutive free cells, at most k; clip_position is then a no-op
                int m = 1;
                if (k >= 2 && c2 == 0) { m = 2; if (k >= 3 && c3 == 0) m = 3; }
                s.r += dr * m; s.c += dc * m;
            }
        } else if (valid) {
            const bool built = nd != -1;                             // check[act] == 0 for act in 4..7
            const int newh = min(nd + 1, CNT_MAX);
            int pl = 0;
            if (built) {
                s.cb = min(s.cb + 1, CNT_MAX);
                h[dl] = (int16_t)newh;
                pl = ((const int16_t*)a.plans)[(size_t)s.pidx * GE + (s.r + dr - 3) * 20 + (s.c + dc - 3)];
                s.cross += newh <= pl ? 1 : 0;                       // running sum of min(height, plan) for iou()
            }
            bool fin = false;
            if (DYN) {
                // neighbours re-evaluated AFTER the build (:199-206)
                const bool boxed_post = built ? ((d == 0 || n0 != 0) && (d == 1 || n1 != 0) && (d == 2 || n2 != 0) && (d == 3 || n3 != 0))
                                              : boxed_pre;
                if (boxed_post) { reward = -100; done = true; fin = true; }
                else if (s.cb >= s.tb + bg) { reward = 0; done = true; fin = true; }     // :207-213
            } else {
                if (s.cb >= s.tb + bg || boxed_pre) { reward = 0; done = true; fin = true; }  // :210-215
            }
            if (!fin && built) {                                     // reward_check (:232-239); time limit NOT tested
                reward = newh > pl ? -1 : (newh == pl ? 10 : 1);
                done = false;
            }
        }
    }
    // iou (:257-276) = sum(min(g, plan)) / (tb + cb - sum); the sum is tracked incrementally in s.cross
    __device__ static double iou(uint32_t*, const Lane& s, int) {
        return (double)s.cross / (double)(s.tb + s.cb - s.cross);
    }
    __device__ static int key0(const Lane& s) { return ((s.r - 3) * 26 + (s.c - 3)) * 2; }   // byte offset of the window corner
    __device__ static int key1(const Lane&) { return 0; }
    static constexpr int PLAN_CELLS = 400;
    __device__ static int plan_value(const KArgs& a, int pidx, int cell) { return (int)((const int16_t*)a.plans)[(size_t)pidx * GE + cell]; }
};

// ================================================================================================
// 1D: Env/1D/DMP_Env_1D_static.py, Env/1D/DMP_Env_1D_dynamic_usedata_plan.py
// LDS per wave: H[e * ES + cell] int16, the bordered 34-cell row (frame = -1), ES = 34 (odd dword stride); PL the
// plan (30 cells), same stride; SC[e][2] float64 observation scalars; POS[e].
template <bool DYN_, int E_>
struct K1D {
    static constexpr bool DYN = DYN_;
    static constexpr int E = E_, D = 7, W = 5, A = 3, TS = 750, GE = 32, ES = 34;
    static constexpr int P_OFF = E * ES / 2;                         // dwords
    static constexpr int SC_OFF = E * ES + ((E * ES) & 1);
    static constexpr int LDS_WORDS = SC_OFF + 4 * E + E;

    __device__ static int16_t* hmap(uint32_t* lds) { return (int16_t*)lds; }
    __device__ static int16_t* plan(uint32_t* lds) { return (int16_t*)(lds + P_OFF); }
    __device__ static double* sc(uint32_t* lds) { return (double*)(lds + SC_OFF); }
    __device__ static int* pos(uint32_t* lds) { return (int*)(lds + SC_OFF + 4 * E); }

    __device__ static void load_grid(uint32_t* lds, const KArgs& a, int env0, int nenv, int lane, const int* rows = nullptr) {
        for (int i = lane; i < E * ES / 2; i += 64) lds[i] = 0xFFFFFFFFu;
        const int16_t* src = (const int16_t*)a.grid;
        int16_t* h = hmap(lds);
        for (int i = lane; i < nenv * GE; i += 64) {
            const int e = i / GE, cell = i - e * GE;
            if (cell < 30) h[e * ES + cell + 2] = src[tile_row(rows, env0, e) * GE + cell];
        }
    }
    __device__ static void store_grid(uint32_t* lds, const KArgs& a, int env0, int nenv, int lane, const int* rows = nullptr) {
        int16_t* dst = (int16_t*)a.grid;
        const int16_t* h = hmap(lds);
        for (int i = lane; i < nenv * GE; i += 64) {
            const int e = i / GE, cell = i - e * GE;
            dst[tile_row(rows, env0, e) * GE + cell] = cell < 30 ? h[e * ES + cell + 2] : (int16_t)0;
        }
    }
    __device__ static void load_plan(uint32_t* lds, const KArgs& a, int e, int pidx, int lane) {  // whole wave
        if (lane < GE / 2) lds[P_OFF + e * (ES / 2) + lane] = ((const uint32_t*)((const int16_t*)a.plans + (size_t)pidx * GE))[lane];
    }
    // the one plan cell a single step() can read: the agent's column (fetched early, placed once the tile is loaded)
    struct PlanCell { int16_t v; };
    __device__ static PlanCell fetch_plan_cell(const KArgs& a, const Lane& s) {
        return PlanCell{((const int16_t*)a.plans)[(size_t)s.pidx * GE + s.r - 2]};
    }
    __device__ static void put_plan_cell(uint32_t* lds, const Lane& s, const PlanCell& pc, int lane) {
        plan(lds)[lane * ES + s.r - 2] = pc.v;
    }
    // reset: DMP_Env_1D_static.py:66-83, DMP_Env_1D_dynamic_usedata_plan.py:40-70
    __device__ static void reset(const KArgs& a, Lane& s, int pidx) {
        if (pidx >= 0) { s.pidx = pidx; s.tb = a.plan_tb[pidx]; }    // pidx < 0: the env keeps its plan and total_brick
        s.r = 2; s.c = 0; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.cross = 0; s.flags = 0;
    }
    __device__ static void clear(uint32_t* lds, int e, int lane) {   // whole wave zeroes env e's interior
        if (lane < 30) hmap(lds)[e * ES + lane + 2] = 0;
    }
    // step: DMP_Env_1D_static.py:85-136
    __device__ static void step(uint32_t* lds, const KArgs& a, Lane& s, int act, int k, int ts, int bg, int lane, int& reward, bool& done) {
        int16_t* h = hmap(lds) + lane * ES + s.r;
        const int hnew = min((int)*h + 1, CNT_MAX);
        const int pl = plan(lds)[lane * ES + s.r - 2];
        const bool drop = act == 2;
        s.cs = min(s.cs + 1, CNT_MAX);
        if (drop) { s.cb = min(s.cb + 1, CNT_MAX); *h = (int16_t)hnew; }
        if (act == 0) s.r = max(s.r - k, 2);                         // clip_position :57-64
        if (act == 1) s.r = min(s.r + k, 31);
        const bool term = drop && s.cb >= s.tb + bg;                 // :107-114, before the time limit
        done = term || s.cs >= ts;
        reward = (drop && !term) ? (hnew > pl ? -1 : (hnew == pl ? 10 : 1)) : 0;   // :117-123
    }
    // iou: DMP_Env_1D_static.py:138-151
    __device__ static double iou(uint32_t* lds, const Lane& s, int lane) {
        const int16_t* h = hmap(lds) + lane * ES + 2;
        const int16_t* pl = plan(lds) + lane * ES;
        int a1 = 0, a2 = 0, kk = 0;
        for (int i = 0; i < 30; ++i) {
            const int g = h[i], p = pl[i];
            a1 += p; a2 += g; kk += max(g - p, 0);
        }
        const int cross = a2 - kk;
        return (double)cross / (double)(a1 + a2 - cross);
    }
    __device__ static int key0(const Lane& s) { return s.r; }
    __device__ static int key1(const Lane&) { return 0; }
    static constexpr int PLAN_CELLS = 30;
    __device__ static int plan_value(const KArgs& a, int pidx, int cell) { return (int)((const int16_t*)a.plans)[(size_t)pidx * GE + cell]; }
};

// ------------------------------------------------------------------------------------------------
// phase 2: write the observation rows of the tile's envs.  orow points at [env0][0] of the target step.
// k0 / k1: the per-lane phase-2 keys of the lane's env (K::key0 / key1).  FULL: the tile holds K::E envs.
// 2D / 3D: lanes 0..48 produce the 7x7 window of one env, lanes 49 / 50 its two scalar slots (staged in LDS by
// write_scalars), and the 51 values leave as ONE contiguous store.  (Writing the scalar slots with a separate
// per-lane store was measured: the partial-line writes cost 55 % -- 4.5 vs 2.9 ms per pass.)
// 1D: 7 values per env, flat, one element per lane.
// VAR: the layout variants of snac_env_desc (frame value, row length a.ld = K::D + tail, the tail itself); the canonical
// instantiation (VAR = false) carries none of it.
struct StepOut { int reward; int done; };                        // per lane: what SNAC_TAIL_RECORD reports besides the header

// one SNAC_TAIL_RECORD value: 0 reward, 1 done, 2 pos_r, 3 pos_c, 4 count_brick, 5 count_step, 6 total_brick, 7 plan_idx
__device__ __forceinline__ int record_value(int j, int reward, int done, int r, int c, int cb, int cs, int tb, int pidx) {
    return j == 0 ? reward : j == 1 ? done : j == 2 ? r : j == 3 ? c : j == 4 ? cb : j == 5 ? cs : j == 6 ? tb : pidx;
}

// LP: every env's whole plan row is in the wave's LDS (k_rollout loads and keeps it; the single-step kernels place one cell): the plan
// tail then comes from there.  From the table in memory it is a vector load in the middle of the row stores, and vmcnt retires in
// order -- every batch of 64 plan cells waited for the stores before it (451-value rows: 9 us per tick and wave).
template <class K, typename OT, bool FULL, bool VAR, bool LP = false>
__device__ __forceinline__ void write_obs(uint32_t* lds, OT* orow, int nenv, int k0, int k1, int lane, const KArgs& a,
                                          const Lane& s, const StepOut& so, const int16_t* plan3 = nullptr) {
    const int LD = VAR ? a.ld : K::D;
    if constexpr (K::D == 51) {
        constexpr int U = 8;                                         // envs per batch (K::E is a multiple of U)
        const int wl = lane < K::W ? lane : 0;
        const int wi = wl / 7, wj = wl - 7 * wi;
        const char* base = (const char*)lds;
        const double* scp = K::sc(lds) + (lane >= K::W ? min(lane - K::W, 1) : 0);
        const bool is_win = lane < K::W;
        OT* p = orow + lane;
        int lane_off;                                                // byte offset of this lane's cell / cell row
        if constexpr (K::A == 8) lane_off = (wi * 26 + wj) * 2;
        else lane_off = wi * K::RS * 8;
        for (int e0 = 0; e0 < (FULL ? K::E : nenv); e0 += U) {
            int v[U];
            double sv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u;                                // < K::E: LDS reads stay in range past nenv
                sv[u] = scp[2 * e];
                const int s0 = __builtin_amdgcn_readlane(k0, e);
                if constexpr (K::A == 8) {
                    v[u] = *(const int16_t*)(base + (e * K::ES * 2 + s0) + lane_off);
                } else {
                    const int off = __builtin_amdgcn_readlane(k1, e) + 2 * wj;
                    const uint64_t w = *(const uint64_t*)(base + (e * 8 + s0) + lane_off);
                    v[u] = ((int)((uint32_t)(w >> off) << 30)) >> 30;    // signed 2-bit field: 0 / 1 / -1
                }
                if constexpr (VAR) v[u] = v[u] < 0 ? a.frame_val : v[u];
            }
            // one fence per batch: every LDS read is in flight before the first store is built (otherwise the
            // compiler sinks each scalar read into its store's exec-masked block and serialises them)
            asm volatile("" ::"v"(sv[0]), "v"(sv[1]), "v"(sv[2]), "v"(sv[3]), "v"(sv[4]), "v"(sv[5]), "v"(sv[6]), "v"(sv[7]));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double val = is_win ? (double)v[u] : sv[u];
                if (lane < K::D && (FULL || e0 + u < nenv)) p[(size_t)(e0 + u) * LD] = (OT)val;
            }
        }
        if constexpr (VAR) {
            if (a.tail) {
                for (int e = 0; e < nenv; ++e) {                     // e is wave-uniform: readlane broadcasts env e's scalars
                    OT* q = orow + (size_t)e * LD + K::D;
                    const int r = __builtin_amdgcn_readlane(s.r, e), c = __builtin_amdgcn_readlane(s.c, e);
                    const int pidx = __builtin_amdgcn_readlane(s.pidx, e);
                    if (a.tail & SNAC_TAIL_POSITION) {
                        if (lane < 2) q[lane] = (OT)(double)(lane == 0 ? r : c);
                        q += 2;
                    }
                    if (a.tail & SNAC_TAIL_PLAN) {
                        constexpr int NP = (K::PLAN_CELLS + 63) / 64;                     // seven batches of 64 cells
                        int pv[NP];                                                       // every read of the env first: from the table in memory
#pragma unroll                                                                            // (LP = false) seven loads in flight, one wait
                        for (int i = 0; i < NP; ++i) {
                            const int cell = min(lane + 64 * i, K::PLAN_CELLS - 1);
                            if constexpr (LP && K::A != 8) { const int pr = cell / 20; pv[i] = (int)((lds[K::P_OFF + pr * K::RS + e] >> (cell - 20 * pr)) & 1u); }
                            else if constexpr (LP) pv[i] = (int)plan3[e * K::PLAN_CELLS + cell];   // 3D: k_rollout's own copy of the rows (the kind keeps no plan in LDS)
                            else pv[i] = K::plan_value(a, pidx, cell);
                        }
#pragma unroll
                        for (int i = 0; i < NP; ++i)
                            if (lane + 64 * i < K::PLAN_CELLS) q[lane + 64 * i] = (OT)(double)pv[i];
                        q += K::PLAN_CELLS;
                    }
                    if (a.tail & SNAC_TAIL_RECORD) {
                        const int val = record_value(lane, __builtin_amdgcn_readlane(so.reward, e), __builtin_amdgcn_readlane(so.done, e),
                                                     r, c, __builtin_amdgcn_readlane(s.cb, e), __builtin_amdgcn_readlane(s.cs, e),
                                                     __builtin_amdgcn_readlane(s.tb, e), pidx);
                        if (lane < 8) q[lane] = (OT)(double)val;
                    }
                }
            }
        }
    } else {
        // 1D: q = e * LD + el
        const int16_t* h = K::hmap(lds);
        const double* scp = K::sc(lds);
        const int* posp = K::pos(lds);
        const int total = nenv * LD;
        for (int q0 = 0; q0 < total; q0 += 64) {                     // uniform trip count: the tail's bpermutes need every lane
            const int q = min(q0 + lane, total - 1);
            const int e = q / LD, el = q - e * LD;
            int v = h[e * K::ES + posp[e] - 2 + min(el, K::W - 1)];
            if constexpr (VAR) v = v < 0 ? a.frame_val : v;
            double val = el < K::W ? (double)v : scp[2 * e + (el >= K::W + 1 ? 1 : 0)];
            if constexpr (VAR) {
                if (a.tail) {                                        // wave-uniform; e differs per lane -> lane e's scalars by bpermute
                    const int pos = posp[e], pidx = __shfl(s.pidx, e);
                    const int rw = __shfl(so.reward, e), dn = __shfl(so.done, e), cb = __shfl(s.cb, e), cs = __shfl(s.cs, e), tb = __shfl(s.tb, e);
                    if (el >= K::D) {
                        int ti = el - K::D, out = 0;
                        if (a.tail & SNAC_TAIL_POSITION) { if (ti == 0) out = pos; ti -= 1; }
                        if (a.tail & SNAC_TAIL_PLAN) {
                            if (ti >= 0 && ti < K::PLAN_CELLS) {
                                if constexpr (LP) out = (int)K::plan(lds)[e * K::ES + ti];
                                else out = K::plan_value(a, pidx, ti);
                            }
                            ti -= K::PLAN_CELLS;
                        }
                        if ((a.tail & SNAC_TAIL_RECORD) && ti >= 0) out = record_value(ti, rw, dn, pos, 0, cb, cs, tb, pidx);
                        val = (double)out;
                    }
                }
            }
            if (q0 + lane < total) orow[q] = (OT)val;
        }
    }
}

// the two scalar observation slots (count_brick, count_step or their normalised forms): one IEEE float64 division per
// lane (no fast-math), staged in LDS for phase 2.
template <class K, typename OT, bool VAR>
__device__ __forceinline__ void write_scalars(uint32_t* lds, const Lane& s, int ts, int lane, const KArgs& a) {
    const double num0 = (double)s.cb, num1 = (double)s.cs;
    const bool norm = VAR ? (a.sc_norm != 0) : K::DYN;
    const double v0 = norm ? num0 / (double)s.tb : num0;
    const double v1 = norm ? num1 / (double)ts : num1;
    if (lane < K::E) {
        double2 v; v.x = v0; v.y = v1;
        *(double2*)(K::sc(lds) + 2 * lane) = v;
        if constexpr (K::D == 7) K::pos(lds)[lane] = s.r;
    }
}

// plan row of a new episode: counter RNG stream 1 for the dataset classes; a static-plan env keeps `keep` -- its own row
// on auto-reset (per-env static plans, hindsight relabelling), desc->static_plan on an explicit reset without indices
template <class K>
__device__ __forceinline__ int pick_plan(const KArgs& a, EnvKeys pk, int episode, int keep) {
    if (K::DYN) return (int)__umulhi(rng_word(pk, (uint32_t)episode), (uint32_t)a.num_plans);
    return keep;
}

template <class K, int WPB>
__device__ __forceinline__ uint32_t* wave_lds() {
    __shared__ __attribute__((aligned(16))) uint32_t lds[WPB * K::LDS_WORDS];
    return lds + (threadIdx.x >> 6) * K::LDS_WORDS;
}

template <class K, typename OT, bool VAR, bool LP = false>
__device__ __forceinline__ void emit_obs(uint32_t* lds, OT* orow, int nenv, const Lane& s, const KArgs& a, int lane, const StepOut& so,
                                         const int16_t* plan3 = nullptr) {
    write_scalars<K, OT, VAR>(lds, s, a.total_step, lane, a);
    if (nenv == K::E) write_obs<K, OT, true, VAR, LP>(lds, orow, nenv, K::key0(s), K::key1(s), lane, a, s, so, plan3);
    else write_obs<K, OT, false, VAR, LP>(lds, orow, nenv, K::key0(s), K::key1(s), lane, a, s, so, plan3);
}

// T fused vector steps (T = 1: one step() call) for one tile of E envs per wave.
// EXPL: actions and / or step sizes come from the caller's arrays.  The counter-RNG instantiation (EXPL = false) has no
// global load in its loop at all: with the null tests at run time the compiler joins both paths behind one
// `s_waitcnt vmcnt(0)`, and vmcnt counts the observation stores too -- every tick would wait for the previous tick's rows.
template <class K, typename OT, int WPB, bool EXPL, bool VAR>
__global__ __launch_bounds__(WPB * 64) void k_rollout(const KArgs a) {
    constexpr int E = K::E;
    const int LD = VAR ? a.ld : K::D;
    const int lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x * WPB + (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(tile * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    uint32_t* lds = wave_lds<K, WPB>();
    // 3D keeps no plan in LDS (a step needs one cell); the layout variants' plan tail needs all 400 every tick: its rows get a copy of
    // their own here, refilled when an env starts over on another row
    constexpr bool P3 = VAR && K::A == 8;
    __shared__ int16_t plan3_all[P3 ? WPB * E * 400 : 1];
    int16_t* const plan3 = plan3_all + (P3 ? (int)(threadIdx.x >> 6) * E * 400 : 0);
    auto load_plan3 = [&](int e, int pidx) {
        if constexpr (P3) {
            if (a.tail & SNAC_TAIL_PLAN) {
                const uint32_t* const src = (const uint32_t*)((const int16_t*)a.plans + (size_t)pidx * 400);
                uint32_t* const dst = (uint32_t*)(plan3 + e * 400);
                for (int i = lane; i < 200; i += 64) dst[i] = src[i];
            }
        }
    };
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;                                                // idle lanes keep an in-range position
    int episode = 0;
    if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
    asm volatile("" : "+v"(episode));                                // waited for HERE, not in the loop's reset branch behind the row stores
    K::load_grid(lds, a, env0, nenv, lane);
    for (int e = 0; e < nenv; ++e) { K::load_plan(lds, a, e, __builtin_amdgcn_readlane(s.pidx, e), lane); load_plan3(e, __builtin_amdgcn_readlane(s.pidx, e)); }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const EnvKeys sk = env_keys(a.key_step, gid), pk = env_keys(a.key_plan, gid);
    int d_eps = 0, d_ret = 0;
    long long d_iou = 0;
    OT* const obs = (OT*)a.obs;
    for (int t = 0; t < a.T; ++t) {
        const size_t row = (size_t)t * (size_t)a.n + (size_t)env0;
        int reward = 0;
        bool done = false;
        const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
        if (__builtin_expect(__any(nr), 0)) {                        // rare paths out of line: a taken branch costs a lone wave ~30 cycles
            const int old_pidx = s.pidx, old_tb = s.tb;
            if (nr) {
                episode += 1;
                const int pidx = pick_plan<K>(a, pk, episode, old_pidx);
                K::reset(a, s, pidx == old_pidx ? -1 : pidx);   // -1: same plan again (static tables): keep tb, no load
                if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
            }
            for (unsigned long long m = __ballot(nr); m; m &= m - 1) {
                const int e = __ffsll(m) - 1;
                const int pe = __builtin_amdgcn_readlane(s.pidx, e);
                K::clear(lds, e, lane);
                // a vector load here waits for every observation store issued before it (vmcnt is in-order): skip it
                // when the env keeps its plan
                if (pe != __builtin_amdgcn_readlane(old_pidx, e)) { K::load_plan(lds, a, e, pe, lane); load_plan3(e, pe); }
            }
        }
        if (active) {
            const uint32_t w = rng_word(sk, a.t0 + (uint32_t)t);
            int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
            if constexpr (EXPL) {
                if (a.actions) act = (int)a.actions[row + lane];
                if (a.step_size) k = min(max((int)a.step_size[row + lane], 1), 3);
            }
            K::step(lds, a, s, act, k, a.ts_done, a.brick_gt, lane, reward, done);
            s.ep_ret = clamp16(s.ep_ret + reward);
            s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
            if (a.reward) a.reward[row + lane] = (float)reward;
            if (a.done) a.done[row + lane] = done ? 1 : 0;
            if (a.actions_out) a.actions_out[row + lane] = (int8_t)act;
            if (a.step_size_out) a.step_size_out[row + lane] = (int8_t)k;
            if (a.plan_idx_out) a.plan_idx_out[row + lane] = (int16_t)s.pidx;
            if (a.first_out) a.first_out[row + lane] = s.cs == 1 ? 1 : 0;   // first step of its episode
        }
        if (__builtin_expect(__any(done), 0)) {
            const double v = K::iou(lds, s, active ? lane : 0);      // idle lanes stay inside the wave's LDS slice
            if (done) { d_eps += 1; d_ret += s.ep_ret; d_iou += __double2ll_rn(v * FX40); }
        }
        if (a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED || (a.obs_mode == SNAC_OBS_LAST && t == a.T - 1)) {
            // SNAC_OBS_TILED: [ceil(N / 64)][T][64][LD] -- the tile's rows of step t follow its rows of step t - 1
            const size_t orow = a.obs_mode == SNAC_OBS_ALL ? row
                              : (a.obs_mode == SNAC_OBS_TILED ? ((size_t)(env0 >> 6) * (size_t)a.tiled_T + (size_t)(a.tiled_t0 + t)) * 64 + (size_t)(env0 & 63) : (size_t)env0);
            emit_obs<K, OT, VAR, true>(lds, obs + orow * LD, nenv, s, a, lane, StepOut{reward, done ? 1 : 0}, plan3);
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

// ------------------------------------------------------------------------------------------------
// Lane-per-env observation rows (round 3): lane l holds the 51 values of env l of a 64-env tile -- cell(el) for the 49 window
// cells, v0 / v1 the two scalar slots.  They are transposed through a staging tile in LDS ([env][51] values, odd dword stride:
// conflict-free) into the tile's contiguous piece of the output (64 x 51 values: 13 056 B of float32, 26 112 B of float64 in two
// halves of 32 envs), read back 16 bytes per lane and stored with global_store_dwordx4: 1 KiB per store instruction, 13 / 26 per
// tile instead of 64 row stores.  stg: TILE_STG_BYTES of 16-byte aligned LDS of this wave (every LDS read of a half is issued
// before its first store; reads of lanes past the tile's end fall into the pad).  g: the tile's first output byte, 16-byte aligned;
// nenv < 64: a ragged tile (nenv * 51 * sizeof(OT) must be a multiple of 16: the callers require N % 4 == 0).
constexpr int TILE_STG_BYTES = 13 * 1024;

template <typename OT, class F>
__device__ __forceinline__ void emit_tile(char* stg, char* g, int lane, int nenv, F cell, double v0, double v1) {
    constexpr int D = 51, W = 49, E = 64;
    constexpr int HALVES = sizeof(OT) == 8 ? 2 : 1, HE = E / HALVES;
    constexpr int STG_BYTES = HE * D * (int)sizeof(OT);              // 13 056 B either way
    constexpr int NF = (STG_BYTES + 1023) / 1024;
    static_assert(NF * 1024 <= TILE_STG_BYTES && STG_BYTES % 16 == 0, "staging tile");
    const bool full = nenv == E;
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
        if (HALVES == 1 || (lane >> 5) == h) {                       // transpose: lane -> row (lane - h * HE) of the staging tile
            OT* const S = (OT*)stg + (lane - h * HE) * D;
#pragma unroll
            for (int el = 0; el < W; ++el) S[el] = (OT)cell(el);
            S[W] = (OT)v0; S[W + 1] = (OT)v1;
        }
        uint4 fv[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) fv[i] = *(const uint4*)(stg + i * 1024 + lane * 16);
        char* const gh = g + (size_t)h * STG_BYTES + lane * 16;
        if (full) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if ((i + 1) * 1024 <= STG_BYTES || i * 1024 + lane * 16 < STG_BYTES) *(uint4*)(gh + i * 1024) = fv[i];
        } else {
            const int valid = min(max(nenv - h * HE, 0), HE) * D * (int)sizeof(OT);
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (i * 1024 + lane * 16 < valid) *(uint4*)(gh + i * 1024) = fv[i];
        }
    }
}

// The same for the layout variants of snac_env_desc (rows of LD = 51 + tail values: 451 for the script/PPO dataset copies, 59 with the
// record tail, ...).  A tile's rows of one step are still ONE contiguous run of 64 * LD values, 16-byte aligned as a whole (the callers
// require N % 4 == 0), so it is staged and flushed in GROUPS of G envs -- the largest power of two whose rows fit the staging tile (a
// multiple of 16 bytes for every LD: G >= 2 with float64, >= 4 with float32).  With 451-value rows a group is two or four envs, so
// nothing may be done "by the lanes of the group's envs" (a first version did, and repeated the window decode 32 times per tick: 3.4 ms
// per 60 ticks, 0.52 of the peak).  Instead every lane FILES what it holds once per tick in a compact record (cmp[lane][19]: the 7
// window row codes, the two scalar slots, reward / done / position / counters / plan row), and a row is assembled by the whole wave,
// lane = value: lanes 0..48 decode a window cell, 49 / 50 take the scalar slots, the next ones the position / record values, and for
// the plan tail lane = plan cell (+ 64 i), read from the wave's plan rows in LDS.  The group then leaves 16 bytes per lane, 1 KiB per
// store instruction.
constexpr int VAR_CMP_WORDS = 19;                                    // per env: 7 + 4 + 8 dwords (odd: conflict-free)

constexpr int VAR_STG_BYTES = 15 * 1024;                              // its staging tile: four 451-value float64 rows (14 432 B)

// 0 / 1 as OT without a conversion instruction: the value's bit pattern is a mask of the constant 1.0
template <typename OT>
__device__ __forceinline__ OT bit_as(uint32_t word, int bit) {
    const int m = ((int)(word << (31 - bit))) >> 31;                 // 0 or -1
    if constexpr (sizeof(OT) == 8) return (OT)__hiloint2double(m & 0x3FF00000, 0);
    else return (OT)__int_as_float(m & 0x3F800000);
}

template <typename OT, int STG_BYTES = VAR_STG_BYTES, int UMAX = 8, class PF>
__device__ __forceinline__ void emit_rows_var(char* stg, uint32_t* cmp, char* g, int lane, int nenv, int LD, int tail, int frame_val,
                                              const uint32_t (&wr)[7], double v0, double v1, const int (&recv)[8], PF plan) {
    constexpr int D = 51, W = 49;
    const int RB = LD * (int)sizeof(OT);
    int G = 64;
    while (G * RB > STG_BYTES) G >>= 1;
    const int pos_n = (tail & SNAC_TAIL_POSITION) ? 2 : 0, plan_n = (tail & SNAC_TAIL_PLAN) ? 400 : 0, rec_n = (tail & SNAC_TAIL_RECORD) ? 8 : 0;
    const int NE = D + pos_n + rec_n;                                // values of a row that come from the compact record (<= 61)
    if (lane < nenv) {                                               // (cmp holds nenv records: k_rollout2dt's writers pass 4)
        uint32_t* const mine = cmp + lane * VAR_CMP_WORDS;
#pragma unroll
        for (int i = 0; i < 7; ++i) mine[i] = wr[i];
        const uint64_t b0 = (uint64_t)__double_as_longlong(v0), b1 = (uint64_t)__double_as_longlong(v1);
        mine[7] = (uint32_t)b0; mine[8] = (uint32_t)(b0 >> 32); mine[9] = (uint32_t)b1; mine[10] = (uint32_t)(b1 >> 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) mine[11 + j] = (uint32_t)recv[j];
    }
    // what THIS lane contributes to every row: source dword in the compact record, kind (0 window cell, 1 scalar slot, 2 integer), place in the row
    int src, kind, sh = 0, dst = lane;
    if (lane < W) { src = lane / 7; sh = 30 - 2 * (lane - 7 * src); kind = 0; }
    else if (lane < D) { src = 7 + 2 * (lane - W); kind = 1; }
    else {
        int k = lane - D;
        kind = 2;
        if (k < pos_n) { src = 11 + 2 + k; dst = D + k; }            // position: record values 2, 3
        else { k -= pos_n; src = 11 + min(k, 7); dst = D + pos_n + plan_n + k; }
    }
    // the plan cells this lane fills in: cell lane + 64 i of every env -> plan row and bit (the same for every env and tick)
    int prow[7], pbit[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { const int pc = min(lane + 64 * i, 399); prow[i] = pc / 20; pbit[i] = pc - 20 * prow[i]; }
    // U envs at a time, every LDS read of the batch issued before its first write: the compiler cannot tell the staging rows from the
    // records and the plan rows, and one env per round trip made a tick latency-bound (64 round trips: 11 instead of 4.6 us per tick)
    const uint32_t m_sc = (uint32_t)-(int)(kind == 1), m_int = (uint32_t)-(int)(kind == 2);   // all ones: a scalar slot / an integer value
    auto batch = [&](auto uc, int e, int e0) {
        constexpr int U = decltype(uc)::value;
        uint32_t lo[U], hi[U], pw[U][7];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t* const c = cmp + (e + u) * VAR_CMP_WORDS + src;
            lo[u] = c[0]; hi[u] = c[1];                              // (src + 1 <= 18: inside the record)
            if (plan_n) {
#pragma unroll
                for (int i = 0; i < 7; ++i) pw[u][i] = plan(e + u, prow[i]);
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // which of the three forms a lane's value takes is a per-lane constant: bit selects on masks (written as `kind == ..` selects
            // the compiler keeps the kinds as exec masks and pays for them in scalar instructions and branches, cf. k_rollout2dt)
            const int cv = ((int)(lo[u] << sh)) >> 30;               // signed 2-bit field: 0 / 1 / -1
            const uint32_t iv = (m_int & lo[u]) | (~m_int & (uint32_t)(cv < 0 ? frame_val : cv));
            const uint64_t cb = (uint64_t)__double_as_longlong((double)(int)iv);
            const uint32_t rl = (m_sc & lo[u]) | (~m_sc & (uint32_t)cb), rh = (m_sc & hi[u]) | (~m_sc & (uint32_t)(cb >> 32));
            const double val = __longlong_as_double((long long)(((uint64_t)rh << 32) | rl));
            OT* const row = (OT*)stg + (e + u - e0) * LD;
            if (lane < NE) row[dst] = (OT)val;
            if (plan_n) {
                OT* const q = row + D + pos_n;
#pragma unroll
                for (int i = 0; i < 7; ++i)
                    if (i < 6 || lane < 16) q[lane + 64 * i] = bit_as<OT>(pw[u][i], pbit[i]);
            }
        }
    };
    for (int e0 = 0; e0 < nenv; e0 += G) {
        const int ge = min(G, nenv - e0);                            // a multiple of 2 (float64) / 4 (float32): N % 4 == 0
        if (G >= 16) {
            // short rows (no plan tail: 51 .. 61 values): a group is 16 or more envs, and the transposition of emit_tile is the cheaper
            // form -- the lanes of the group's envs write their own values (0.28 against 0.70 ms per 60 ticks for the 51-value L-Net rows)
            if (lane >= e0 && lane < e0 + ge) {
                OT* const S = (OT*)stg + (lane - e0) * LD;
#pragma unroll
                for (int el = 0; el < W; ++el) {
                    const int i = el / 7, j = el - 7 * i;
                    const int v = ((int)(wr[i] << (30 - 2 * j))) >> 30;
                    S[el] = (OT)(v < 0 ? frame_val : v);
                }
                S[W] = (OT)v0; S[W + 1] = (OT)v1;
                OT* q = S + D;
                if (pos_n) { q[0] = (OT)recv[2]; q[1] = (OT)recv[3]; q += 2; }
                if (rec_n) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) q[j] = (OT)recv[j];
                }
            }
        } else {
            int e = e0;
            if constexpr (UMAX >= 8) {
                for (; e + 8 <= e0 + ge; e += 8) batch(std::integral_constant<int, 8>{}, e, e0);
            } else if constexpr (UMAX >= 4) {                        // (callers with four envs at a time: no eight-env batch to hold registers for)
                for (; e + 4 <= e0 + ge; e += 4) batch(std::integral_constant<int, 4>{}, e, e0);
            }
            for (; e + 2 <= e0 + ge; e += 2) batch(std::integral_constant<int, 2>{}, e, e0);
        }
        const int valid = ge * RB;                                   // a multiple of 16
        char* const gh = g + (size_t)e0 * RB + lane * 16;
        const char* const sh = stg + lane * 16;
        const int full = valid >> 10, rest = valid & 1023;           // whole 1 KiB store instructions (wave-uniform), bytes of the last one
        int i = 0;
        for (; i + 4 <= full; i += 4) {                              // four at a time, their LDS reads issued first
            uint4 fv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) fv[k] = *(const uint4*)(sh + (i + k) * 1024);
#pragma unroll
            for (int k = 0; k < 4; ++k) *(uint4*)(gh + (i + k) * 1024) = fv[k];
        }
        {
            uint4 fv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) fv[k] = *(const uint4*)(stg + min((i + k) * 1024 + lane * 16, STG_BYTES - 16));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i + k < full) *(uint4*)(gh + (i + k) * 1024) = fv[k];
                else if (i + k == full && lane * 16 < rest) *(uint4*)(gh + (i + k) * 1024) = fv[k];
            }
        }
    }
}

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

typedef const uint32_t __attribute__((address_space(4))) cmem_u32;   // constant address space: uniform addresses become s_load

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
        const bool drop = act == 4;
        s.cs = min(s.cs + 1, CNT_MAX);
        if (drop) {
            s.cb = min(s.cb + 1, CNT_MAX);
            if (active) *cw = w | (1ull << off);                     // += 1 then clamp to 1 (:115, :134-135)
            gcnt += was ? 0 : 1;
            inter += (!was && planned) ? 1 : 0;
        }
        if (act == 0) s.c = max(s.c - k, 3);                         // clip_position :74-83
        if (act == 1) s.c = min(s.c + k, 22);
        if (act == 2) s.r = min(s.r + k, 22);                        // "up" is row + k (:100-103)
        if (act == 3) s.r = max(s.r - k, 3);
        const bool term = drop && s.cb >= s.tb + a.brick_gt;         // :117-126, tested before the time limit
        const bool done = active && (term || s.cs >= a.ts_done);
        const int reward = (drop && !term && !was && planned) ? 5 : 0;   // un-clamped cell vs plan (:129-133)
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
            emit_tile<OT>(stg, obs0 + (size_t)t * tstride, lane, nenv,
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
constexpr int TB_MAX = 2048;   // plan_tb rows staged in LDS per block

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
        const int rc = q_newh > q_pl ? -1 : (q_newh == q_pl ? 10 : 1);   // reward_check on the built cell
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
        const int nd = d == 0 ? n0 : (d == 1 ? n1 : (d == 2 ? n2 : n3));
        const bool valid = (unsigned)act < 8u, is_build = valid && act >= 4;
        const bool boxed_pre = n0 != 0 && n1 != 0 && n2 != 0 && n3 != 0;
        const bool first = s.cs == 0;
        s.cs = min(s.cs + 1, CNT_MAX);
        const bool can_move = valid && act < 4 && nd == 0;           // check[act] == 0
        const int m = (k >= 2 && c2 == 0) ? ((k >= 3 && c3 == 0) ? 3 : 2) : 1;   // move_step: consecutive free cells, <= k
        s.r += can_move ? dr * m : 0;
        s.c += can_move ? dc * m : 0;
        const bool built = is_build && nd != -1;                     // check[act] == 0 for act in 4..7
        const int newh = min(nd + 1, CNT_MAX);
        if (active && built) h[dl] = (int16_t)newh;
        s.cb = built ? min(s.cb + 1, CNT_MAX) : s.cb;
        const bool limit = s.cb >= s.tb + a.brick_gt;
        bool done = (s.cs >= a.ts_done) || (!DYN && boxed_pre);      // moves, blocked moves, blocked builds
        int reward0;                                                 // the part of the reward that does not need the plan
        bool sel;                                                    // reward = reward_check(built cell)
        if (DYN) {
            // neighbours re-evaluated AFTER the build: the built cell now blocks its direction
            const bool boxed_post = built ? ((d == 0 || n0 != 0) && (d == 1 || n1 != 0) && (d == 2 || n2 != 0) && (d == 3 || n3 != 0)) : boxed_pre;
            const bool fin = is_build && (boxed_post || limit);
            reward0 = (is_build && boxed_post) ? -100 : 0;
            sel = is_build && !fin && built;
            done = fin ? true : (sel ? false : done);
        } else {
            const bool fin = is_build && (limit || boxed_pre);
            reward0 = 0;
            sel = is_build && !fin && built;
            done = fin ? true : (sel ? false : done);
        }
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

template <bool DYN, typename OT, bool EXPL>
__global__ __launch_bounds__(576) void k_rollout3db(const KArgs a) {
    using K = K3D<DYN, 64>;
    constexpr int D = K::D, ROWB = D * (int)sizeof(OT), GE = K::GE, NT = 576;
    __shared__ __attribute__((aligned(16))) uint32_t hm[64 * K::ES / 2];      // 64 bordered height maps, 1356 bytes apart (odd dword stride)
    __shared__ __attribute__((aligned(16))) int16_t stg16[64 * 56];           // the tick's 64 windows as int16 cells: [env][window row][8], a row = one 16-byte write
    __shared__ double rtab[TB_MAX];                                  // 1 / total_brick per plan row: no division in the stepper
    __shared__ int16_t tbtab[TB_MAX];
    __shared__ double ssc[2][64][2];
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
    for (int i = tid; i < a.num_plans; i += NT) { const int tb = a.plan_tb[i]; tbtab[i] = (int16_t)tb; rtab[i] = 1.0 / (double)tb; }
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
        auto reset_scalars = [&]() {                                 // K3D::reset without the map
            episode += 1;
            const int np = pick_plan<K>(a, pk, episode, s.pidx);
            if (np != s.pidx) {                                      // K::reset: a new row brings its total_brick, the same row keeps the header's
                s.pidx = np; s.tb = tbtab[np];
                dtb = (double)s.tb; rtb = rtab[np];
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
        bool nr_prev = false;                                        // started one a tick ago: the map may not be cleared yet
        int pb_idx = -1, pb_h = 0;                                   // the cell built a tick ago: may not be in the map yet
        int act = 0, k = 1, act_n = 0, k_n = 1;
        if constexpr (EXPL) { load_inputs(0, act, k); load_inputs(1, act_n, k_n); }
        else inputs_of(0, act, k);
        int pl = (int)((const int16_t*)a.plans)[(size_t)s.pidx * GE + target_cell(act)];
        for (int t = 0; t < a.T; ++t) {
            const int par = t & 1;
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
            const int nd = d == 0 ? n0 : (d == 1 ? n1 : (d == 2 ? n2 : n3));
            const bool valid = (unsigned)act < 8u, is_build = valid && act >= 4;
            const bool boxed_pre = n0 != 0 && n1 != 0 && n2 != 0 && n3 != 0;
            const bool first = s.cs == 0;
            s.cs = min(s.cs + 1, CNT_MAX);
            const bool can_move = valid && act < 4 && nd == 0;
            const int m = (k >= 2 && c2 == 0) ? ((k >= 3 && c3 == 0) ? 3 : 2) : 1;
            s.r += can_move ? dr * m : 0;
            s.c += can_move ? dc * m : 0;
            const bool built = active && is_build && nd != -1;
            const int newh = min(nd + 1, CNT_MAX);
            s.cb = built ? min(s.cb + 1, CNT_MAX) : s.cb;
            s.cross += (built && newh <= pl) ? 1 : 0;                    // the tick's only vector-memory wait: pl, loaded a tick ago
            const bool limit = s.cb >= s.tb + a.brick_gt;
            bool done = (s.cs >= a.ts_done) || (!DYN && boxed_pre);
            int reward = 0;
            const int rcheck = newh > pl ? -1 : (newh == pl ? 10 : 1);
            if (DYN) {
                const bool boxed_post = built ? ((d == 0 || n0 != 0) && (d == 1 || n1 != 0) && (d == 2 || n2 != 0) && (d == 3 || n3 != 0)) : boxed_pre;
                const bool fin = is_build && (boxed_post || limit);
                reward = is_build ? (boxed_post ? -100 : ((!limit && built) ? rcheck : 0)) : 0;
                done = fin ? true : ((is_build && built) ? false : done);
            } else {
                const bool fin = is_build && (limit || boxed_pre);
                reward = (is_build && !fin && built) ? rcheck : 0;
                done = fin ? true : ((is_build && built) ? false : done);
            }
            done = done && active;
            s.ep_ret = clamp16(s.ep_ret + reward);
            s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
            pb_idx = built ? hidx + dl : -1; pb_h = newh;
            {   // the tick's outputs -> its half of the double buffer; the scalar slots by the exact-reciprocal quotients of Roll3D
                const double c0 = (double)s.cb, c1 = (double)s.cs;
                double v0 = c0, v1 = c1;
                if (DYN) {
                    const double q0 = c0 * rtb, q1 = c1 * rT;
                    v0 = s.tb > 0 ? __builtin_fma(__builtin_fma(-q0, dtb, c0), rtb, q0) : c0 / dtb;
                    v1 = __builtin_fma(__builtin_fma(-q1, dT, c1), rT, q1);
                }
                double2 sv; sv.x = v0; sv.y = v1;
                *(double2*)ssc[par][lane] = sv;
                spub[par][lane] = make_int2(s.r | (s.c << 8) | (nr ? 1 << 16 : 0), built ? ((hidx + dl) | (newh << 16)) : -1);
                srew[par][lane] = (float)reward;
                sdone[par][lane] = done ? 1 : 0;
                if (done) sfin[par][lane] = make_int4(s.cross, s.tb + s.cb - s.cross, s.ep_ret, 0);
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
    char* const obs0 = (char*)a.obs + ((tl ? ((size_t)(env0 >> 6) * (size_t)a.tiled_T + (size_t)a.tiled_t0) * 64 : (size_t)env0) + (size_t)e0) * ROWB;
    const size_t tstride = (tl ? (size_t)64 : (size_t)a.n) * ROWB;
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
    const int npieces = rows * ROWB / 16;
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
        {
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
// (m & a) | (~m & b) as the one instruction it is (left to itself the compiler hoists ~m out of a loop and issues two)
__device__ __forceinline__ uint32_t bfi32(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}

template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ int dpp_from(int idv, int v) { return __builtin_amdgcn_update_dpp(idv, v, CTRL, ROWS, 0xf, false); }

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
            const bool term = drop && cb >= tb + a.brick_gt;         // :107-114, before the time limit
            const bool done = seg && (term || cs >= a.ts_done);
            const unsigned long long donem = __ballot(done);
            const int last = donem ? (__ffsll((long long)donem) - 1) : (nl - 1);     // the segment's last lane
            const bool in = seg && lane <= last;
            // ---- positions: inclusive scan of x -> min(max(x + d, 2), 31)
            int sa = 0, slo = -4096, shi = 4096;
            if (in) { sa = act == 0 ? -k : (act == 1 ? k : 0); slo = 2; shi = 31; }
            // Hillis-Steele inside the rows of 16 lanes (row_shr 1, 2, 4, 8), then the rows' last lanes to the rows behind them; a
            // lane without a source composes with the identity (0, -4096, 4096), so no step is conditional
            auto compose = [&](int pa, int plo, int phi) {               // the earlier ticks first, then this lane's function
                const int nlo = min(max(plo + sa, slo), shi), nhi = min(max(phi + sa, slo), shi);
                sa += pa; slo = nlo; shi = nhi;
            };
            compose(dpp_from<0x111>(0, sa), dpp_from<0x111>(-4096, slo), dpp_from<0x111>(4096, shi));
            compose(dpp_from<0x112>(0, sa), dpp_from<0x112>(-4096, slo), dpp_from<0x112>(4096, shi));
            compose(dpp_from<0x114>(0, sa), dpp_from<0x114>(-4096, slo), dpp_from<0x114>(4096, shi));
            compose(dpp_from<0x118>(0, sa), dpp_from<0x118>(-4096, slo), dpp_from<0x118>(4096, shi));
            compose(dpp_from<0x142, 0xa>(0, sa), dpp_from<0x142, 0xa>(-4096, slo), dpp_from<0x142, 0xa>(4096, shi));
            compose(dpp_from<0x143, 0xc>(0, sa), dpp_from<0x143, 0xc>(-4096, slo), dpp_from<0x143, 0xc>(4096, shi));
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
            const int reward = (drop && !term) ? (hnew > pl ? -1 : (hnew == pl ? 10 : 1)) : 0;   // :117-123
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
            const bool term = drop && cb >= tb + a.brick_gt;         // :117-126, before the time limit
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
            const int reward = (drop && !term && !was && planned) ? 5 : 0;   // un-clamped cell vs plan (:129-133)
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

// transition(state, action) of the MCTS variants (Env/2D/DMP_ENV_2D_dynamic_MCTS.py:117-175 and the eight sibling files;
// caller: script/MCTS/utils/mcts_Qvalue_dynamic.py:88,118): ONE step of the same K::step on an explicit state, batched over
// a.n tree edges.  The state arrays are a node pool; edge i reads row src_index[i] and writes row dst_index[i] (out of
// place), the observation / reward / done rows are per edge.  snac_transition: no auto-reset, no episodic sums (a search
// is not an episode).  snac_step is the same kernel on the identity rows with both switched on.
template <class K, typename OT, int WPB, bool VAR>
__global__ __launch_bounds__(WPB * 64) void k_transition(const KArgs a) {
    constexpr int E = K::E;
    const int LD = VAR ? a.ld : K::D;
    const int lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x * WPB + (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(tile * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int edge = env0 + (active ? lane : 0);
    uint32_t* lds = wave_lds<K, WPB>();
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;
    int episode = 0;
    const size_t srow = row_of(a.src_index, a.pool, edge), drow = row_of(a.dst_index, a.pool, edge);
    if (active) { s.unpack(a.hdr[srow]); episode = a.episode[srow]; }
    const uint64_t gid = (uint64_t)(a.env_id_base + edge);
    // snac_step with auto_reset: an env whose previous step returned done starts a new episode first (as in k_rollout)
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    // one step reads one plan cell (3D: one of four): fetch it now, next to the tile's records, instead of staging plans
    const typename K::PlanCell pc = K::fetch_plan_cell(a, s);
    int* const rows = (int*)K::sc(lds);                          // free until write_scalars: the tile's gather / scatter rows
    if (a.src_index && active) rows[lane] = (int)srow;
    K::load_grid(lds, a, env0, nenv, lane, a.src_index ? rows : nullptr);
    for (unsigned long long m = __ballot(nr); m; m &= m - 1) K::clear(lds, __ffsll(m) - 1, lane);
    if (active) K::put_plan_cell(lds, s, pc, lane);
    int reward = 0;
    bool done = false;
    if (active) {
        const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
        int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
        if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }   // snac_step_scalar: by value, no input arrays
        if (a.actions) act = (int)a.actions[edge];
        if (a.step_size) k = (int)a.step_size[edge];
        k = min(max(k, 1), 3);
        K::step(lds, a, s, act, k, a.ts_done, a.brick_gt, lane, reward, done);
        s.ep_ret = clamp16(s.ep_ret + reward);
        s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
        if (a.reward) a.reward[edge] = (float)reward;
        if (a.done) a.done[edge] = done ? 1 : 0;
    }
    if (a.stats_on && __any(done)) {                             // snac_step: episodic sums (the IoU needs the whole plan)
        if constexpr (K::A != 8)
            for (unsigned long long m = __ballot(done); m; m &= m - 1) {
                const int e = __ffsll(m) - 1;
                K::load_plan(lds, a, e, __builtin_amdgcn_readlane(s.pidx, e), lane);
            }
        const double v = K::iou(lds, s, active ? lane : 0);
        if (done) {
            a.stat_episodes[drow] += 1;
            a.stat_return[drow] += s.ep_ret;
            a.stat_iou_fx[drow] += __double2ll_rn(v * FX40);
        }
    }
    if constexpr (VAR && K::A != 8) {
        // 1D / 2D: a row with the plan tail needs its env's whole plan: staged in LDS first (every load before the first row store: from
        // the table in memory each batch of 64 plan cells is a vector load behind the stores before it; 2D PPO rows at 16 384 envs:
        // 26.6 -> 19.3 us per tick).  3D keeps reading the table: its 800-byte plans cost more to stage than they save (130 -> 160 us
        // at 65 536 envs, six instead of nine waves per CU).
        if (a.obs && (a.tail & SNAC_TAIL_PLAN)) {
            for (int e = 0; e < nenv; ++e) K::load_plan(lds, a, e, __builtin_amdgcn_readlane(s.pidx, e), lane);
            emit_obs<K, OT, VAR, true>(lds, (OT*)a.obs + (size_t)env0 * LD, nenv, s, a, lane, StepOut{reward, done ? 1 : 0});
        } else if (a.obs) {
            emit_obs<K, OT, VAR>(lds, (OT*)a.obs + (size_t)env0 * LD, nenv, s, a, lane, StepOut{reward, done ? 1 : 0});
        }
    } else {
        if (a.obs) emit_obs<K, OT, VAR>(lds, (OT*)a.obs + (size_t)env0 * LD, nenv, s, a, lane, StepOut{reward, done ? 1 : 0});
    }
    if (a.dst_index && active) rows[lane] = (int)drow;           // the scalar slots have been written out by now
    K::store_grid(lds, a, env0, nenv, lane, a.dst_index ? rows : nullptr);
    if (active) { a.hdr[drow] = s.pack(); a.episode[drow] = episode; }
}

// ------------------------------------------------------------------------------------------------
// 3D single step / tree edge without the LDS image.  k_transition stages every 800-byte height map into the bordered LDS
// image and back with 2-byte accesses (profiles/r02_step_*: 524 288 edges in 468 us = 1.5 TB/s of HBM traffic, bound by
// ~25 narrow memory instructions per edge, not by HBM).  But one step changes ONE cell.  Here a wave takes 32 edges:
//   lane = edge   header, counter RNG or the caller's action, the six neighbour / path cells and the plan cell read straight
//                 from the source record (frame cells are -1 by their coordinates), K3D::step by selects;
//   per edge      the record is copied source -> destination in 16-byte lanes (50 lanes x 16 B), the built cell patched in
//                 the lane that holds it; lanes 0..48 gather the 7x7 window from the source record (patched the same way),
//                 lanes 49 / 50 take the scalar slots: one 408-byte row store.
// Four wide memory instructions per edge instead of ~25 narrow ones.  Semantics are K3D::step's (tests compare with the CPU
// restatement exactly as for k_transition); layout variants stay on the generic kernel.
template <bool DYN, typename OT, int WPB, bool INPLACE>
__global__ __launch_bounds__(WPB * 64) void k_transition3d(const KArgs a) {
    using K = K3D<DYN, 8>;
    constexpr int E = 32;
    __shared__ double sc_all[WPB][E][2];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int edge0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (edge0 >= a.n) return;
    const int nedge = min(E, a.n - edge0);
    const bool active = lane < nedge;
    const int edge = edge0 + (active ? lane : 0);
    const int srow = (int)row_of(a.src_index, a.pool, edge), drow = (int)row_of(a.dst_index, a.pool, edge);
    Lane s;
    s.unpack(a.hdr[srow]);
    int episode = a.episode[srow];
    const uint64_t gid = (uint64_t)(a.env_id_base + edge);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions) act = (int)a.actions[edge];
    if (a.step_size) k = (int)a.step_size[edge];
    k = min(max(k, 1), 3);
    const int16_t* const g16 = (const int16_t*)a.grid;
    const int16_t* const src = g16 + (size_t)srow * K::GE;
    // a cell of the source map in bordered coordinates: the frame is -1, a freshly reset env is empty
    auto cell = [&](int R, int C) -> int {
        const bool in = (unsigned)(R - 3) < 20u && (unsigned)(C - 3) < 20u;
        const int v = (in && !nr) ? (int)src[(R - 3) * 20 + (C - 3)] : 0;
        return in ? v : -1;
    };
    const int d = act & 3;
    const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
    const int n0 = cell(s.r, s.c - 1), n1 = cell(s.r, s.c + 1), n2 = cell(s.r + 1, s.c), n3 = cell(s.r - 1, s.c);
    const int c2 = cell(s.r + 2 * dr, s.c + 2 * dc), c3 = cell(s.r + 3 * dr, s.c + 3 * dc);
    const int tr = s.r + dr - 3, tc = s.c + dc - 3;
    const bool inside = (unsigned)tr < 20u && (unsigned)tc < 20u;
    const int tcell = inside ? tr * 20 + tc : 0;
    const int pl = ((const int16_t*)a.plans)[(size_t)s.pidx * K::GE + tcell];
    // K3D::step by selects (the same formulation as Roll3D::tick, without its deferral)
    const int nd = d == 0 ? n0 : (d == 1 ? n1 : (d == 2 ? n2 : n3));
    const bool valid = (unsigned)act < 8u, is_build = valid && act >= 4;
    const bool boxed_pre = n0 != 0 && n1 != 0 && n2 != 0 && n3 != 0;
    s.cs = min(s.cs + 1, CNT_MAX);
    const bool can_move = valid && act < 4 && nd == 0;
    const int m = (k >= 2 && c2 == 0) ? ((k >= 3 && c3 == 0) ? 3 : 2) : 1;
    s.r += can_move ? dr * m : 0;
    s.c += can_move ? dc * m : 0;
    const bool built = active && is_build && nd != -1;
    const int newh = min(nd + 1, CNT_MAX);
    s.cb = built ? min(s.cb + 1, CNT_MAX) : s.cb;
    s.cross += (built && newh <= pl) ? 1 : 0;
    const bool limit = s.cb >= s.tb + a.brick_gt;
    bool done = (s.cs >= a.ts_done) || (!DYN && boxed_pre);
    int reward = 0;
    const int rcheck = newh > pl ? -1 : (newh == pl ? 10 : 1);
    if (DYN) {
        const bool boxed_post = built ? ((d == 0 || n0 != 0) && (d == 1 || n1 != 0) && (d == 2 || n2 != 0) && (d == 3 || n3 != 0)) : boxed_pre;
        const bool fin = is_build && (boxed_post || limit);
        reward = is_build ? (boxed_post ? -100 : ((!limit && built) ? rcheck : 0)) : 0;
        done = fin ? true : ((is_build && built) ? false : done);
    } else {
        const bool fin = is_build && (limit || boxed_pre);
        reward = (is_build && !fin && built) ? rcheck : 0;
        done = fin ? true : ((is_build && built) ? false : done);
    }
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (active) {
        if (a.reward) a.reward[edge] = (float)reward;
        if (a.done) a.done[edge] = done ? 1 : 0;
        if (a.stats_on && done) {                                // snac_step: episodic sums
            const double v = K::iou(nullptr, s, 0);
            a.stat_episodes[drow] += 1;
            a.stat_return[drow] += s.ep_ret;
            a.stat_iou_fx[drow] += __double2ll_rn(v * FX40);
        }
    }
    // the two scalar observation slots of every edge -> LDS
    {
        const double c0 = (double)s.cb, c1 = (double)s.cs;
        double (*sc)[2] = sc_all[wv];
        if (lane < E) { sc[lane][0] = DYN ? c0 / (double)s.tb : c0; sc[lane][1] = DYN ? c1 / (double)a.total_step : c1; }
    }
    const int tpatch = built ? tcell : -1;                           // interior index of the cell this step changed
    const int key_r = s.r, key_c = s.c;
    const int wl = lane < K::W ? lane : 0, wi = wl / 7, wj = wl - 7 * wi;
    const uint4* const g4 = (const uint4*)a.grid;
    uint4* const g4w = (uint4*)a.grid;
    OT* const orow = a.obs ? (OT*)a.obs + (size_t)edge0 * K::D + lane : nullptr;
    // U edges at a time: every load of the group (record lanes and window cells, both from the SOURCE records) is issued before
    // the group's first store, so U records are in flight per wave instead of one -- the loop used to be a load -> store ->
    // load chain, the compiler may not move a load over a store into the same array.  Legal by the contract of
    // snac_transition (include/snac_hip.h): a destination row is never the source row of a different edge of the call.
    // INPLACE (identity rows: every snac_step): no record is copied at all -- a step writes its one changed cell, an auto-reset
    // writes the empty map -- so the whole tile's window gathers are issued up front (vmcnt retires in order: a later group's
    // loads would also wait for the row stores in front of them).
    constexpr int U = INPLACE ? 32 : 8;
    for (int e0 = 0; e0 < nedge; e0 += U) {                          // wave-uniform: readlane broadcasts an edge's scalars
        uint4 rec[INPLACE ? 1 : U];
        int wcell[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u, nedge - 1);                    // a short last group reloads its last edge (not stored)
            const int se = __builtin_amdgcn_readlane(srow, e);
            const int tp = __builtin_amdgcn_readlane(tpatch, e), nh = __builtin_amdgcn_readlane(newh, e);
            const bool fresh = __builtin_amdgcn_readlane((int)nr, e) != 0;
            // the 7x7 window around the NEW position, from the source record with the built cell patched in
            wcell[u] = -1;
            if (orow) {
                const int R = __builtin_amdgcn_readlane(key_r, e) - 3 + wi, C = __builtin_amdgcn_readlane(key_c, e) - 3 + wj;
                const bool in = (unsigned)(R - 3) < 20u && (unsigned)(C - 3) < 20u;
                const int idx = in ? (R - 3) * 20 + (C - 3) : 0;
                const int v = (in && !fresh && lane < K::W) ? (int)g16[(size_t)se * K::GE + idx] : 0;
                wcell[u] = in ? (idx == tp ? nh : v) : -1;
            }
            // a step in place (snac_step, or a tree edge onto its own row) changes ONE cell: no record copy
            if constexpr (!INPLACE) {
                const bool copy = fresh || se != __builtin_amdgcn_readlane(drow, e);
                rec[u] = (fresh || lane >= 50 || !copy) ? make_uint4(0u, 0u, 0u, 0u) : g4[(size_t)se * 50 + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u;
            if (e < nedge) {
                const int de = __builtin_amdgcn_readlane(drow, e);
                const int tp = __builtin_amdgcn_readlane(tpatch, e), nh = __builtin_amdgcn_readlane(newh, e);
                const bool copy = __builtin_amdgcn_readlane((int)nr, e) != 0 || (!INPLACE && de != __builtin_amdgcn_readlane(srow, e));
                if (!copy) {
                    if (tp >= 0 && lane == 0) ((int16_t*)a.grid)[(size_t)de * K::GE + tp] = (int16_t)nh;
                } else if (lane < 50) {
                    uint4 v = INPLACE ? make_uint4(0u, 0u, 0u, 0u) : rec[u];
                    if (tp >= 0 && (tp >> 3) == lane) {              // this lane's 8 cells hold the built one
                        const int hw = tp & 7, sh = (hw & 1) * 16;
                        const uint32_t keep = ~(0xFFFFu << sh), put = ((uint32_t)nh & 0xFFFFu) << sh;
                        if ((hw >> 1) == 0) v.x = (v.x & keep) | put;
                        else if ((hw >> 1) == 1) v.y = (v.y & keep) | put;
                        else if ((hw >> 1) == 2) v.z = (v.z & keep) | put;
                        else v.w = (v.w & keep) | put;
                    }
                    g4w[(size_t)de * 50 + lane] = v;
                }
                if (orow && lane < K::D) {
                    const double scal = sc_all[wv][e][lane >= K::W ? min(lane - K::W, 1) : 0];
                    orow[(size_t)e * K::D] = (OT)(lane < K::W ? (double)wcell[u] : scal);
                }
            }
        }
    }
    if (active) { a.hdr[drow] = s.pack(); a.episode[drow] = episode; }
}

// ------------------------------------------------------------------------------------------------
// 3D tree edges with gathered rows, round 4 (snac_transition: Env/3D/DMP_simulator_3d_dynamic_triangle_MCTS.py:195-277, one call per
// edge in script/MCTS/utils/mcts_Qvalue_dynamic.py:88,118).  k_transition3d issues, per edge, one record load, one record store, one
// 49-lane gather of 2-byte window cells and one 408-byte row store: half of its memory instructions are narrow (0.48 of the peak for
// its 2.06 KB per edge).  Here the records of a wave's 32 edges pass through LDS once and every memory instruction is wide:
//   in      the 32 source records (800 bytes each) arrive as 16-byte pieces, lane = piece of the group's 1600 (the owning edge's row by
//           ds_bpermute), 25 loads in flight, and lie in LDS as REC[edge][400 cells];
//   step    lane = edge: the six neighbour / path cells from its record in LDS, K3D::step by selects (k_transition3d's formulation),
//           the built cell patched into the record, the 7x7 window round the NEW position read back cell by cell (ds_read_i16);
//   out     the records leave again as 16-byte pieces (an edge onto its own row writes its one changed cell instead), and the 32
//           observation rows through emit_tile -- the staging tile reuses the records' LDS -- as 16-byte stores, 1 KiB per instruction.
// 25 + 25 + 13 wide memory instructions per 32 edges instead of 128.  Conditions: gathered / scattered rows (an index array given),
// canonical layout, observations wanted, m % 4 == 0 and a 16-byte aligned obs; everything else stays on k_transition3d.
template <bool DYN, typename OT, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_edges3d(const KArgs a) {
    using K = K3D<DYN, 8>;
    constexpr int E = 32, GE = K::GE, RECB = GE * 2;                 // 800 bytes per record
    constexpr int WAVE_BYTES = E * RECB > TILE_STG_BYTES ? E * RECB : TILE_STG_BYTES;
    __shared__ __attribute__((aligned(16))) char lds_all[WPB * WAVE_BYTES];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int edge0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (edge0 >= a.n) return;
    const int nedge = min(E, a.n - edge0);
    const bool active = lane < nedge;
    const int edge = edge0 + (active ? lane : 0);
    char* const rec = lds_all + wv * WAVE_BYTES;
    const int srow = (int)row_of(a.src_index, a.pool, edge), drow = (int)row_of(a.dst_index, a.pool, edge);
    Lane s;
    s.unpack(a.hdr[srow]);
    int episode = a.episode[srow];
    const uint64_t gid = (uint64_t)(a.env_id_base + edge);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    // ---- the records of the group: piece q = lane + 64 p of 1600, edge q / 50, 16-byte lane q % 50 of its record
    {
        const uint4* const g4 = (const uint4*)a.grid;
        uint4 pv[25];
#pragma unroll
        for (int p = 0; p < 25; ++p) {
            const int q = p * 64 + lane, e = q / 50, l = q - 50 * e;
            const int se = __shfl(srow, e);
            const bool fresh = __shfl((int)nr, e) != 0;
            pv[p] = (e < nedge && !fresh) ? g4[(size_t)se * 50 + l] : make_uint4(0u, 0u, 0u, 0u);   // a freshly reset env is empty
        }
#pragma unroll
        for (int p = 0; p < 25; ++p) *(uint4*)(rec + (p * 64 + lane) * 16) = pv[p];
    }
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions) act = (int)a.actions[edge];
    if (a.step_size) k = (int)a.step_size[edge];
    k = min(max(k, 1), 3);
    int16_t* const mine = (int16_t*)(rec + (lane & (E - 1)) * RECB);   // (lanes 32..63 shadow 0..31: nothing of theirs is stored)
    // a cell of the edge's map in bordered coordinates: the frame is -1
    auto cell = [&](int R, int C) -> int {
        const bool in = (unsigned)(R - 3) < 20u && (unsigned)(C - 3) < 20u;
        const int v = (int)mine[in ? (R - 3) * 20 + (C - 3) : 0];
        return in ? v : -1;
    };
    const int d = act & 3;
    const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
    const int tr = s.r + dr - 3, tc = s.c + dc - 3;
    const bool inside = (unsigned)tr < 20u && (unsigned)tc < 20u;
    const int tcell = inside ? tr * 20 + tc : 0;
    const int pl = ((const int16_t*)a.plans)[(size_t)s.pidx * GE + tcell];
    const int n0 = cell(s.r, s.c - 1), n1 = cell(s.r, s.c + 1), n2 = cell(s.r + 1, s.c), n3 = cell(s.r - 1, s.c);
    const int c2 = cell(s.r + 2 * dr, s.c + 2 * dc), c3 = cell(s.r + 3 * dr, s.c + 3 * dc);
    // K3D::step by selects (the formulation of k_transition3d / Roll3D::tick)
    const int nd = d == 0 ? n0 : (d == 1 ? n1 : (d == 2 ? n2 : n3));
    const bool valid = (unsigned)act < 8u, is_build = valid && act >= 4;
    const bool boxed_pre = n0 != 0 && n1 != 0 && n2 != 0 && n3 != 0;
    s.cs = min(s.cs + 1, CNT_MAX);
    const bool can_move = valid && act < 4 && nd == 0;
    const int m = (k >= 2 && c2 == 0) ? ((k >= 3 && c3 == 0) ? 3 : 2) : 1;
    s.r += can_move ? dr * m : 0;
    s.c += can_move ? dc * m : 0;
    const bool built = active && is_build && nd != -1;
    const int newh = min(nd + 1, CNT_MAX);
    s.cb = built ? min(s.cb + 1, CNT_MAX) : s.cb;
    s.cross += (built && newh <= pl) ? 1 : 0;
    const bool limit = s.cb >= s.tb + a.brick_gt;
    bool done = (s.cs >= a.ts_done) || (!DYN && boxed_pre);
    int reward = 0;
    const int rcheck = newh > pl ? -1 : (newh == pl ? 10 : 1);
    if (DYN) {
        const bool boxed_post = built ? ((d == 0 || n0 != 0) && (d == 1 || n1 != 0) && (d == 2 || n2 != 0) && (d == 3 || n3 != 0)) : boxed_pre;
        const bool fin = is_build && (boxed_post || limit);
        reward = is_build ? (boxed_post ? -100 : ((!limit && built) ? rcheck : 0)) : 0;
        done = fin ? true : ((is_build && built) ? false : done);
    } else {
        const bool fin = is_build && (limit || boxed_pre);
        reward = (is_build && !fin && built) ? rcheck : 0;
        done = fin ? true : ((is_build && built) ? false : done);
    }
    done = done && active;
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (built) mine[tcell] = (int16_t)newh;                          // the record and the window show the built cell
    if (active) {
        if (a.reward) a.reward[edge] = (float)reward;
        if (a.done) a.done[edge] = done ? 1 : 0;
        a.hdr[drow] = s.pack();
        a.episode[drow] = episode;
        if (a.stats_on && done) {                                    // (snac_step on gathered rows: episodic sums)
            const double v = K::iou(nullptr, s, 0);
            stat_add(a.stat_episodes + drow, 1);
            stat_add(a.stat_return + drow, s.ep_ret);
            stat_add(a.stat_iou_fx + drow, __double2ll_rn(v * FX40));
        }
    }
    // ---- the window round the new position, lane = edge
    int cellv[K::W];
#pragma unroll
    for (int el = 0; el < K::W; ++el) { const int i = el / 7, j = el - 7 * i; cellv[el] = cell(s.r - 3 + i, s.c - 3 + j); }
    // ---- the records leave: 16-byte pieces again; an edge onto its own row (not freshly reset) writes its one changed cell instead
    const bool copy = nr || drow != srow;
    if (active && !copy && built) ((int16_t*)a.grid)[(size_t)drow * GE + tcell] = (int16_t)newh;
    {
        uint4* const g4w = (uint4*)a.grid;
        uint4 pv[25];
#pragma unroll
        for (int p = 0; p < 25; ++p) pv[p] = *(const uint4*)(rec + (p * 64 + lane) * 16);
#pragma unroll
        for (int p = 0; p < 25; ++p) {
            const int q = p * 64 + lane, e = q / 50, l = q - 50 * e;
            const int de = __shfl(drow, e);
            const bool cp = __shfl((int)copy, e) != 0;
            if (e < nedge && cp) g4w[(size_t)de * 50 + l] = pv[p];
        }
    }
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const double v0 = DYN ? c0 / (double)s.tb : c0, v1 = DYN ? c1 / (double)a.total_step : c1;
    emit_tile<OT>(rec, (char*)a.obs + (size_t)edge0 * K::D * sizeof(OT), lane, nedge, [&](int el) { return cellv[el]; }, v0, v1);
}

// ------------------------------------------------------------------------------------------------
// 2D single step / tree edge without the LDS image.  k_transition expands every 80-byte bit-board into the bordered two-bit
// LDS image and squeezes it back (20 rows per edge, for a window that shows 7 of them and a step that changes one bit).
// Here a wave takes E edges and nothing is staged:
//   lane = edge   header, counter RNG or the caller's action, K2D::step on the agent's row word and the plan's row word;
//                 then the 7 row words around the NEW position, cut to the 7 window columns and re-coded as two-bit cells
//                 (00 empty / 01 brick / 11 frame, as in the LDS image): the whole 7x7 window is 98 bits = 4 registers;
//   per edge      four v_readlane broadcast those registers, lane l < 49 extracts the signed two-bit field at bit 2 l
//                 (0 / 1 / -1), lanes 49 / 50 take the broadcast scalar slots: one 408-byte row store, no load, no LDS;
//   the record    in place (snac_step): only the row word a brick changed is written back.  Gathered / scattered rows
//                 (snac_transition): copied source -> destination three records per instruction (lane = row word), the
//                 changed word patched on the way.
// Semantics are K2D::step's (tests compare with the CPU restatement exactly as for k_transition); layout variants stay on
// the generic kernel.
template <bool DYN, typename OT, int WPB, int E>
__global__ __launch_bounds__(WPB * 64) void k_transition2d(const KArgs a) {
    using K = K2D<DYN, 64>;
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int edge0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (edge0 >= a.n) return;
    const int nedge = min(E, a.n - edge0);
    const bool active = lane < nedge;
    const int edge = edge0 + (active ? lane : 0);
    const int srow = (int)row_of(a.src_index, a.pool, edge), drow = (int)row_of(a.dst_index, a.pool, edge);
    Lane s;
    s.unpack(a.hdr[srow]);
    int episode = a.episode[srow];
    const uint64_t gid = (uint64_t)(a.env_id_base + edge);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions) act = (int)a.actions[edge];
    if (a.step_size) k = (int)a.step_size[edge];
    k = min(max(k, 1), 3);
    const uint32_t* const g32 = (const uint32_t*)a.grid;
    const uint32_t* const src = g32 + (size_t)srow * K::GE;
    const uint32_t* const prow = (const uint32_t*)a.plans + (size_t)s.pidx * K::GE;
    // ---- K2D::step (DMP_Env_2D_dynamic_usedata_plan.py:85-147) on the agent's row word; a freshly reset board is empty
    const int q0 = min(max(s.r - 3, 0), K::GE - 1), bit = min(max(s.c - 3, 0), 19);
    const uint32_t row0 = nr ? 0u : src[q0];
    const bool was = ((row0 >> bit) & 1u) != 0u, planned = ((prow[q0] >> bit) & 1u) != 0u;
    const bool drop = active && act == 4;
    const uint32_t newrow = row0 | (1u << bit);                      // += 1 then clamp to 1 (:115, :134-135)
    const int patch = drop ? q0 : -1;                                // the board row this step changed
    s.cs = min(s.cs + 1, CNT_MAX);
    if (drop) s.cb = min(s.cb + 1, CNT_MAX);
    if (act == 0) s.c = max(s.c - k, 3);                             // clip_position :74-83
    if (act == 1) s.c = min(s.c + k, 22);
    if (act == 2) s.r = min(s.r + k, 22);                            // "up" is row + k (:100-103)
    if (act == 3) s.r = max(s.r - k, 3);
    const bool term = drop && s.cb >= s.tb + a.brick_gt;             // :117-126, tested before the time limit
    const bool done = term || s.cs >= a.ts_done;
    const int reward = (drop && !term && !was && planned) ? 5 : 0;   // un-clamped cell vs plan (:129-133)
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (active) {
        if (a.reward) a.reward[edge] = (float)reward;
        if (a.done) a.done[edge] = done ? 1 : 0;
        if (a.stats_on && done) {                                    // snac_step: episodic sums; the boolean IoU needs board and plan
            int inter = 0, uni = 0;
            uint32_t gq[K::GE], pq[K::GE];
#pragma unroll
            for (int q = 0; q < K::GE; ++q) { gq[q] = nr ? 0u : src[q]; pq[q] = prow[q]; }   // all 40 loads in flight together
#pragma unroll
            for (int q = 0; q < K::GE; ++q) {
                const uint32_t g = q == patch ? newrow : gq[q];
                inter += __popc(g & pq[q]); uni += __popc(g | pq[q]);
            }
            const double v = (double)inter / (double)uni;
            a.stat_episodes[drow] += 1;
            a.stat_return[drow] += s.ep_ret;
            a.stat_iou_fx[drow] += __double2ll_rn(v * FX40);
        }
    }
    // ---- the 7x7 window around the new position as 49 two-bit cells: window cell l = 7 i + j is the field at bit 2 l
    uint32_t win[4] = {0u, 0u, 0u, 0u};
    double sc0 = 0.0, sc1 = 0.0;
    if (a.obs) {
        const int sh = s.c - 3;                                      // first window column, bordered: 0..19
        constexpr uint32_t FRAME26 = 0x3800007u;                     // frame columns 0-2 and 23-25 of an interior row
        const uint32_t frm = spread16((FRAME26 >> sh) & 0x7Fu) * 3u; // 11 in every frame cell of the 7 columns
        uint32_t enc[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = s.r - 6 + i;                               // board row of window row i
            const bool in = (unsigned)q < (unsigned)K::GE;
            const int qc = in ? q : 0;
            uint32_t g = (in && !nr) ? src[qc] : 0u;
            g = qc == patch ? newrow : g;                            // a drop does not move: the changed row is window row 3
            enc[i] = in ? (spread16(((g << 3) >> sh) & 0x7Fu) | frm) : 0x3FFFu;
        }
        const uint64_t lo = (uint64_t)enc[0] | ((uint64_t)enc[1] << 14) | ((uint64_t)enc[2] << 28) | ((uint64_t)enc[3] << 42) | ((uint64_t)enc[4] << 56);
        const uint64_t hi = (uint64_t)(enc[4] >> 8) | ((uint64_t)enc[5] << 6) | ((uint64_t)enc[6] << 20);
        win[0] = (uint32_t)lo; win[1] = (uint32_t)(lo >> 32); win[2] = (uint32_t)hi; win[3] = (uint32_t)(hi >> 32);
        const double c0 = (double)s.cb, c1 = (double)s.cs;
        sc0 = DYN ? c0 / (double)s.tb : c0;
        sc1 = DYN ? c1 / (double)a.total_step : c1;
    }
    // ---- the record
    uint32_t* const g32w = (uint32_t*)a.grid;
    if (a.src_index || a.dst_index) {
        // three records per instruction: lane = (edge of the trio, row word)
        const int sub = lane / K::GE, q = lane - sub * K::GE;
        for (int e0 = 0; e0 < nedge; e0 += 3) {
            const int e = e0 + sub;
            const bool ok = sub < 3 && e < nedge;
            const int el = (ok ? e : e0) << 2;
            const int se = __builtin_amdgcn_ds_bpermute(el, srow), de = __builtin_amdgcn_ds_bpermute(el, drow);
            const int pe = __builtin_amdgcn_ds_bpermute(el, patch), fresh = __builtin_amdgcn_ds_bpermute(el, (int)nr);
            const uint32_t ne = (uint32_t)__builtin_amdgcn_ds_bpermute(el, (int)newrow);
            if (ok) {
                uint32_t v = fresh ? 0u : g32[(size_t)se * K::GE + q];
                v = q == pe ? ne : v;
                g32w[(size_t)de * K::GE + q] = v;
            }
        }
    } else if (active) {
        if (nr) for (int q = 0; q < K::GE; ++q) g32w[(size_t)drow * K::GE + q] = q == patch ? newrow : 0u;
        else if (drop) g32w[(size_t)drow * K::GE + q0] = newrow;
    }
    if (active) { a.hdr[drow] = s.pack(); a.episode[drow] = episode; }
    // ---- the observation rows: broadcast, extract, one store per edge
    if (a.obs) {
        const int wsel = min(lane >> 4, 3), wsh = 2 * (lane & 15);
        const int slo0 = (int)(uint32_t)__double_as_longlong(sc0), shi0 = (int)(uint32_t)(__double_as_longlong(sc0) >> 32);
        const int slo1 = (int)(uint32_t)__double_as_longlong(sc1), shi1 = (int)(uint32_t)(__double_as_longlong(sc1) >> 32);
        OT* const orow = (OT*)a.obs + (size_t)edge0 * K::D + lane;
        const bool is_win = lane < K::W;
        for (int e = 0; e < nedge; ++e) {                            // wave-uniform: readlane broadcasts edge e's registers
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)win[0], e), w1 = (uint32_t)__builtin_amdgcn_readlane((int)win[1], e);
            const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)win[2], e), w3 = (uint32_t)__builtin_amdgcn_readlane((int)win[3], e);
            const uint32_t ww = wsel == 0 ? w0 : (wsel == 1 ? w1 : (wsel == 2 ? w2 : w3));
            const int cellv = ((int)((ww >> wsh) << 30)) >> 30;      // signed 2-bit field: 0 / 1 / -1
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane(slo0, e), a1 = (uint32_t)__builtin_amdgcn_readlane(shi0, e);
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane(slo1, e), b1 = (uint32_t)__builtin_amdgcn_readlane(shi1, e);
            const double scal = __longlong_as_double((long long)(((uint64_t)(lane == K::W ? a1 : b1) << 32) | (lane == K::W ? a0 : b0)));
            if (lane < K::D) orow[(size_t)e * K::D] = (OT)(is_win ? (double)cellv : scal);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// snac_step on the identity rows, round 3.  k_transition2d / k_transition3d spend their time in the texture addresser, not in
// HBM (profiles/r02j_step_summary.txt: 43 % / 26 % of the wave cycles stalled at ISSUE, half of the HBM rate): they issue one
// narrow memory instruction per edge -- a 408-byte row store, in 3D also a 49-lane gather of 2-byte cells -- and seven scattered
// 4-byte / 2-byte loads per lane.  Here a wave takes a tile of 64 consecutive envs and every memory instruction is wide:
//   2D  the tile's 64 records (5 120 contiguous bytes) arrive as five 16-byte-per-lane loads and lie in LDS; lane l steps env l
//       on its row word, builds the 7 window rows as two-bit codes (k_transition2d's encoding) and hands them to emit_tile;
//   3D  lane l loads the 7 window rows of ITS env as seven 16-byte loads (8 cells from a column clamped into the record, 2-byte
//       aligned: the hardware takes unaligned global accesses) into a scratch row in LDS with -1 on either side, so that frame
//       cells, the neighbour / path cells of K3D::step and the window cells are all ds_read_i16 at (row, 4 + column - first
//       column); the built cell is patched there; an env that moved reloads its rows round the new position (mostly L2 hits);
//   both    the 51 values of an env leave through emit_tile (LDS transposition, 1 KiB stores); the staging tile reuses the
//       record / scratch LDS, whose values are in registers by then.  Episodic sums by no-return atomics (nothing waits for them).
// Write-back: the header, the episode counter of an env that was reset, the ONE changed row word / cell (a reset env: its record).
// Identity rows only (snac_step, snac_step_scalar), N % 4 == 0 and a 16-byte aligned obs; the canonical layout, in 2D also the layout
// variants of large batches (k_step2d<.., VAR>: from 45 056 / 32 769 / 24 576 envs, half-filled tiles for 24 577 .. 32 768; k_step3d<.., VAR>:
// from 24 576); everything else -- tree edges with gathered rows, the other layout variants, N = 1 of
// the single-env classes -- stays on k_transition2d / 3d / k_transition.

// VAR: the layout variants of snac_env_desc (rows of a.ld values: the 451-value rows of the PPO copies are what a trainer that steps
// tens of thousands of envs per tick reads): the rows leave through emit_rows_var (k_rollout2d's row assembly), the plan tail from the
// lanes' plan rows in LDS.  25 KB of LDS per wave, one block of four waves per CU -- 65 536 envs are exactly one round.
// TE = 32: half-filled tiles (lanes 32 .. 63 idle) -- twice the waves for batches that do not fill the CUs with 64 rows of kilobytes per wave.
template <bool DYN, typename OT, int WPB, bool VAR = false, int TE = 64>
__global__ __launch_bounds__(WPB * 64) void k_step2d(const KArgs a) {
    using K = K2D<DYN, 64>;
    constexpr int E = TE, GE = K::GE;
    static_assert(E * GE * 4 <= TILE_STG_BYTES, "the records fit the staging tile");
    constexpr int PLW = VAR ? GE * 65 : 0, CMPW = VAR ? 64 * VAR_CMP_WORDS : 0;           // the envs' plan rows [row][65], emit_rows_var's records
    constexpr int WAVE_WORDS = (VAR ? VAR_STG_BYTES : TILE_STG_BYTES) / 4 + PLW + CMPW;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[WPB * WAVE_WORDS];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    uint32_t* const rec = lds_all + wv * WAVE_WORDS;                 // [64][20] row words, then the staging tile
    // ---- every load that does not depend on another: the tile's records (16 bytes per lane), header, episode counter
    uint4 rv[5];
    {
        const uint4* const g4 = (const uint4*)a.grid + (size_t)env0 * 5;
        // nontemporal, like k_step3d's window rows: read once per tick, and kept out of the way of the row stores' lines in L2
        // (46.1 against 48.4 us per tick at N = 524 288, three runs each; the tree-edge kernels, whose parents are shared by
        // their children, lose 15-25 % with it and keep plain loads)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int g = i * 64 + lane;
            rv[i] = make_uint4(0u, 0u, 0u, 0u);
            if (g < nenv * 5) { const u32x4 t = __builtin_nontemporal_load((const u32x4*)(g4 + g)); rv[i] = make_uint4(t.x, t.y, t.z, t.w); }
        }
    }
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;
    int episode = 0;
    if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions && active) act = (int)a.actions[env];
    if (a.step_size && active) k = (int)a.step_size[env];
    k = min(max(k, 1), 3);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const uint32_t* const prow = (const uint32_t*)a.plans + (size_t)s.pidx * GE;
    const int q0 = min(max(s.r - 3, 0), GE - 1), bit = min(max(s.c - 3, 0), 19);
    const uint32_t pword = prow[q0];                                 // the one dependent load: the plan row under the agent (L2)
#pragma unroll
    for (int i = 0; i < 5; ++i) ((uint4*)rec)[i * 64 + lane] = rv[i];
    uint32_t* const mine = rec + lane * GE;
    if (nr) {                                                        // a freshly reset board is empty
#pragma unroll
        for (int q = 0; q < GE; ++q) mine[q] = 0u;
    }
    // ---- K2D::step (DMP_Env_2D_dynamic_usedata_plan.py:85-147) on the agent's row word
    const uint32_t row0 = mine[q0];
    const bool was = ((row0 >> bit) & 1u) != 0u, planned = ((pword >> bit) & 1u) != 0u;
    const bool drop = active && act == 4;
    const uint32_t newrow = row0 | (1u << bit);                      // += 1 then clamp to 1 (:115, :134-135)
    s.cs = min(s.cs + 1, CNT_MAX);
    if (drop) { s.cb = min(s.cb + 1, CNT_MAX); mine[q0] = newrow; }
    if (act == 0) s.c = max(s.c - k, 3);                             // clip_position :74-83
    if (act == 1) s.c = min(s.c + k, 22);
    if (act == 2) s.r = min(s.r + k, 22);                            // "up" is row + k (:100-103)
    if (act == 3) s.r = max(s.r - k, 3);
    const bool term = drop && s.cb >= s.tb + a.brick_gt;             // :117-126, tested before the time limit
    const bool done = active && (term || s.cs >= a.ts_done);
    const int reward = (drop && !term && !was && planned) ? 5 : 0;   // un-clamped cell vs plan (:129-133)
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (active) {
        if (a.reward) a.reward[env] = (float)reward;
        if (a.done) a.done[env] = done ? 1 : 0;
        a.hdr[env] = s.pack();
        if (nr) {
            a.episode[env] = episode;
            uint32_t* const gw = (uint32_t*)a.grid + (size_t)env * GE;
#pragma unroll
            for (int q = 0; q < GE; ++q) gw[q] = mine[q];
        } else if (drop) {
            ((uint32_t*)a.grid)[(size_t)env * GE + q0] = newrow;
        }
    }
    if (a.stats_on && __builtin_expect(__any(done), 0)) {            // snac_step: episodic sums; the boolean IoU needs board and plan
        if (done) {
            int inter = 0, uni = 0;
            for (int q = 0; q < GE; ++q) { const uint32_t g = mine[q], p = prow[q]; inter += __popc(g & p); uni += __popc(g | p); }
            const double v = (double)inter / (double)uni;
            stat_add(a.stat_episodes + env, 1);
            stat_add(a.stat_return + env, s.ep_ret);
            stat_add(a.stat_iou_fx + env, __double2ll_rn(v * FX40));
        }
    }
    if (!a.obs) return;
    // ---- the 7x7 window round the new position as two-bit codes (00 empty / 01 brick / 11 frame), 14 bits per row
    uint32_t wr[7];
    {
        const int sh = s.c - 3;                                      // first window column, bordered: 0..19
        constexpr uint32_t FRAME26 = 0x3800007u;                     // frame columns 0-2 and 23-25 of an interior row
        const uint32_t frm = spread16((FRAME26 >> sh) & 0x7Fu) * 3u;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = s.r - 6 + i;                               // board row of window row i
            const bool in = (unsigned)q < (unsigned)GE;
            const uint32_t g = mine[in ? q : 0];
            wr[i] = in ? (spread16(((g << 3) >> sh) & 0x7Fu) | frm) : 0x3FFFu;
        }
    }
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const bool norm = VAR ? (a.sc_norm != 0) : DYN;
    const double v0 = norm ? c0 / (double)s.tb : c0, v1 = norm ? c1 / (double)a.total_step : c1;
    if constexpr (VAR) {
        uint32_t* const pl = rec + VAR_STG_BYTES / 4;                // [20][65]: lane l's column holds its env's plan rows
        uint32_t* const cmp = pl + PLW;
        if (a.tail & SNAC_TAIL_PLAN) {
#pragma unroll
            for (int q = 0; q < GE; ++q) pl[q * 65 + lane] = prow[q];
        }
        const int recv[8] = {reward, done ? 1 : 0, s.r, s.c, s.cb, s.cs, s.tb, s.pidx};   // SNAC_TAIL_RECORD's values (record_value)
        emit_rows_var<OT>((char*)rec, cmp, (char*)a.obs + (size_t)env0 * (size_t)a.ld * sizeof(OT), lane, nenv, a.ld, a.tail, a.frame_val, wr, v0, v1,
                          recv, [&](int e, int row) { return pl[row * 65 + e]; });
    } else {
        emit_tile<OT>((char*)rec, (char*)a.obs + (size_t)env0 * K::D * sizeof(OT), lane, nenv,
                      [&](int el) { const int i = el / 7, j = el - 7 * i; return ((int)(wr[i] << (30 - 2 * j))) >> 30; }, v0, v1);
    }
}

typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));   // a 16-byte global access at a 2-byte aligned address

// VAR: the layout variants (rows of a.ld values: 451 with the plan tail of the PPO copies): the heads (window, scalar slots, position,
// record) leave in groups of 16 envs through the staging tile, lane = value; the plan tail of an env is its plan row itself -- 50
// lanes load it 16 bytes each, convert their 8 cells, and the 3200 (1600) bytes are turned through the staging tile into 16-byte
// pieces in row order; eight envs' loads are issued before the first of their stores (a load behind stores waits for them).
template <bool DYN, typename OT, int WPB, bool VAR = false>
__global__ __launch_bounds__(WPB * 64) void k_step3d(const KArgs a) {
    using K = K3D<DYN, 8>;
    constexpr int E = 64, GE = K::GE;
    // scratch per lane: 8 bytes of -1, 7 rows of 12 cells [2 x -1][8 loaded cells][2 x -1], 8 bytes of -1.  A window column may lie up
    // to 3 cells left or 4 right of the loaded block: what a row lacks in pads, its neighbour's pads (or the lane's own leading /
    // trailing 8 bytes) supply.  184 bytes per lane: 46 dwords, a 2-way bank pattern; the whole scratch is smaller than the staging tile.
    constexpr int LS = 184, RB = 24, R0 = 8;
    constexpr int WAVE_BYTES = E * LS > TILE_STG_BYTES ? E * LS : TILE_STG_BYTES;
    static_assert(WAVE_BYTES % 16 == 0, "16-byte aligned staging tiles");
    __shared__ __attribute__((aligned(16))) char lds_all[WPB * WAVE_BYTES];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    char* const scr = lds_all + wv * WAVE_BYTES;
    char* const mine = scr + lane * LS;
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;
    int episode = 0;
    if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
    *(uint64_t*)mine = ~0ull;                                        // the pads
    *(uint64_t*)(mine + R0 + 7 * RB) = ~0ull;
#pragma unroll
    for (int i = 0; i < 7; ++i) { *(uint32_t*)(mine + R0 + i * RB) = ~0u; *(uint32_t*)(mine + R0 + i * RB + 20) = ~0u; }
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const uint32_t w = rng_word(env_keys(a.key_step, gid), a.t0);
    int act = (int)(((w >> 16) * (uint32_t)K::A) >> 16), k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    if (a.use_scalar) { act = a.act_scalar; k = a.k_scalar; }
    if (a.actions && active) act = (int)a.actions[env];
    if (a.step_size && active) k = (int)a.step_size[env];
    k = min(max(k, 1), 3);
    const bool nr = active && a.auto_reset && (s.flags & SNAC_FLAG_NEED_RESET);
    if (nr) {
        const int old_pidx = s.pidx, old_tb = s.tb;
        episode += 1;
        const int pidx = pick_plan<K>(a, env_keys(a.key_plan, gid), episode, old_pidx);
        K::reset(a, s, pidx == old_pidx ? -1 : pidx);
        if (pidx == old_pidx) { s.pidx = old_pidx; s.tb = old_tb; }
    }
    const int16_t* const src = (const int16_t*)a.grid + (size_t)env * GE;
    // the 7 window rows round (r, c) -> scratch; returns the cell index of window column 0 in a scratch row.  Interior column
    // of window column j: c - 6 + j; 8 cells are loaded from `start` (clamped so that they lie inside the row), to cells 2..9.
    auto load_window = [&](int r, int c) -> int {
        const int start = min(max(c - 6, 0), 12);
        uint4 v[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = r - 6 + i;                                 // interior row of window row i
            const bool in = (unsigned)q < 20u;
            v[i] = make_uint4(~0u, ~0u, ~0u, ~0u);                   // a frame row
            if (in) {
                v[i] = make_uint4(0u, 0u, 0u, 0u);                   // a freshly reset env is empty
                if (active && !nr) {
                    // nontemporal: the rows are streamed once per tick (100.2 against 103.8 us per tick at N = 524 288, three runs each)
                    const u32x4_a2 t = __builtin_nontemporal_load((const u32x4_a2*)(src + q * 20 + start));
                    v[i] = make_uint4(t.x, t.y, t.z, t.w);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            uint32_t* const d = (uint32_t*)(mine + R0 + i * RB + 4);
            d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
        return 2 + (c - 6) - start;
    };
    const int d = act & 3;
    const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
    const int tr = s.r + dr - 3, tc = s.c + dc - 3;                  // the build target in plan coordinates
    const bool inside = (unsigned)tr < 20u && (unsigned)tc < 20u;
    const int tcell = inside ? tr * 20 + tc : 0;
    const int pl = ((const int16_t*)a.plans)[(size_t)s.pidx * GE + tcell];
    const int h0 = load_window(s.r, s.c);
    constexpr int RC = RB / 2;                                       // cells per scratch row
    const int16_t* const cen = (const int16_t*)(mine + R0) + 3 * RC + h0 + 3;   // the agent's cell
    // ---- K3D::step by selects (the formulation of k_transition3d / Roll3D::tick)
    const int n0 = cen[-1], n1 = cen[1], n2 = cen[RC], n3 = cen[-RC];    // check_sur: left, right, "up" (row + 1), "down"
    const int dl = dr * RC + dc;
    const int c2 = cen[2 * dl], c3 = cen[3 * dl];
    const int nd = d == 0 ? n0 : (d == 1 ? n1 : (d == 2 ? n2 : n3));
    const bool valid = (unsigned)act < 8u, is_build = valid && act >= 4;
    const bool boxed_pre = n0 != 0 && n1 != 0 && n2 != 0 && n3 != 0;
    s.cs = min(s.cs + 1, CNT_MAX);
    const bool can_move = valid && act < 4 && nd == 0;
    const int m = (k >= 2 && c2 == 0) ? ((k >= 3 && c3 == 0) ? 3 : 2) : 1;
    const int old_r = s.r, old_c = s.c;
    s.r += can_move ? dr * m : 0;
    s.c += can_move ? dc * m : 0;
    const bool built = active && is_build && nd != -1;
    const int newh = min(nd + 1, CNT_MAX);
    s.cb = built ? min(s.cb + 1, CNT_MAX) : s.cb;
    s.cross += (built && newh <= pl) ? 1 : 0;
    const bool limit = s.cb >= s.tb + a.brick_gt;
    bool done = (s.cs >= a.ts_done) || (!DYN && boxed_pre);
    int reward = 0;
    const int rcheck = newh > pl ? -1 : (newh == pl ? 10 : 1);
    if (DYN) {
        const bool boxed_post = built ? ((d == 0 || n0 != 0) && (d == 1 || n1 != 0) && (d == 2 || n2 != 0) && (d == 3 || n3 != 0)) : boxed_pre;
        const bool fin = is_build && (boxed_post || limit);
        reward = is_build ? (boxed_post ? -100 : ((!limit && built) ? rcheck : 0)) : 0;
        done = fin ? true : ((is_build && built) ? false : done);
    } else {
        const bool fin = is_build && (limit || boxed_pre);
        reward = (is_build && !fin && built) ? rcheck : 0;
        done = fin ? true : ((is_build && built) ? false : done);
    }
    done = done && active;
    s.ep_ret = clamp16(s.ep_ret + reward);
    s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
    if (built) ((int16_t*)(mine + R0))[(3 + dr) * RC + h0 + 3 + dc] = (int16_t)newh;   // the window shows the built cell
    if (active) {
        if (a.reward) a.reward[env] = (float)reward;
        if (a.done) a.done[env] = done ? 1 : 0;
        a.hdr[env] = s.pack();
        if (nr) a.episode[env] = episode;
        if (built && !nr) ((int16_t*)a.grid)[(size_t)env * GE + tcell] = (int16_t)newh;
        if (a.stats_on && done) {                                    // snac_step: episodic sums
            const double v = K::iou(nullptr, s, 0);
            stat_add(a.stat_episodes + env, 1);
            stat_add(a.stat_return + env, s.ep_ret);
            stat_add(a.stat_iou_fx + env, __double2ll_rn(v * FX40));
        }
    }
    for (unsigned long long mk = __ballot(nr); mk; mk &= mk - 1) {   // a reset env's record: empty, but for the cell it built
        const int e = __ffsll(mk) - 1;
        const int tp = __builtin_amdgcn_readlane(built ? tcell : -1, e), nh = __builtin_amdgcn_readlane(newh, e);
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
    // ---- the window round the NEW position: an env that moved reloads its rows (the neighbours' lines are in L2 by now)
    int h1 = h0;
    if (s.r != old_r || s.c != old_c) h1 = load_window(s.r, s.c);
    int cellv[K::W];
    {
        const int16_t* const wp = (const int16_t*)(mine + R0) + h1;
#pragma unroll
        for (int el = 0; el < K::W; ++el) { const int i = el / 7, j = el - 7 * i; cellv[el] = wp[i * RC + j]; }
    }
    const double c0 = (double)s.cb, c1 = (double)s.cs;
    const bool norm = VAR ? (a.sc_norm != 0) : DYN;
    const double v0 = norm ? c0 / (double)s.tb : c0, v1 = norm ? c1 / (double)a.total_step : c1;
    if constexpr (!VAR) {
        emit_tile<OT>(scr, (char*)a.obs + (size_t)env0 * K::D * sizeof(OT), lane, nenv, [&](int el) { return cellv[el]; }, v0, v1);
    } else {
        typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte global store at a 4-byte aligned address
        const int LD = a.ld;
        const int pos_n = (a.tail & SNAC_TAIL_POSITION) ? 2 : 0, plan_n = (a.tail & SNAC_TAIL_PLAN) ? 400 : 0, rec_n = (a.tail & SNAC_TAIL_RECORD) ? 8 : 0;
        const int NE = K::D + pos_n + rec_n;                         // values of a row beside the plan tail (<= 61)
        OT* const orow = (OT*)a.obs + (size_t)env0 * LD;
        // ---- heads: 16 envs at a time, each lane of the group files its NE values, then one env per store, lane = value
        const int rv[8] = {reward, done ? 1 : 0, s.r, s.c, s.cb, s.cs, s.tb, s.pidx};   // SNAC_TAIL_RECORD's values (record_value)
        const int dst = lane < K::D + pos_n ? lane : lane + plan_n;  // the record lies behind the plan tail
        for (int g0 = 0; g0 < nenv; g0 += 16) {
            if (lane >= g0 && lane < g0 + 16) {
                OT* const S = (OT*)scr + (lane - g0) * NE;
#pragma unroll
                for (int el = 0; el < K::W; ++el) S[el] = (OT)(double)cellv[el];
                S[K::W] = (OT)v0; S[K::W + 1] = (OT)v1;
                OT* q = S + K::D;
                if (pos_n) { q[0] = (OT)(double)rv[2]; q[1] = (OT)(double)rv[3]; q += 2; }
                if (rec_n) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) q[j] = (OT)(double)rv[j];
                }
            }
            const int ge = min(16, nenv - g0);
            OT hv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = ((const OT*)scr)[r * NE + min(lane, NE - 1)];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r < ge && lane < NE) orow[(size_t)(g0 + r) * LD + dst] = hv[r];
        }
        // ---- plan tails: eight envs' rows loaded, then each turned through the staging tile into pieces in row order
        if (plan_n) {
            constexpr int CP = 16 / (int)sizeof(OT);                 // cells per 16-byte piece of the output: 2 / 4
            constexpr int NPC = 400 / CP;                            // pieces per tail: 200 / 100
            for (int b0 = 0; b0 < nenv; b0 += 8) {
                uint4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int pe = __builtin_amdgcn_readlane(s.pidx, min(b0 + u, nenv - 1));
                    t[u] = ((const uint4*)((const int16_t*)a.plans + (size_t)pe * 400))[min(lane, 49)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (b0 + u < nenv) {                             // wave-uniform
                        if (lane < 50) {
                            const uint32_t w4[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
                            OT* const S = (OT*)scr + lane * 8;
#pragma unroll
                            for (int c = 0; c < 8; ++c) S[c] = (OT)(double)(int)(int16_t)(w4[c >> 1] >> ((c & 1) * 16));
                        }
                        char* const gq = (char*)(orow + (size_t)(b0 + u) * LD + K::D + pos_n);
                        uint4 pv[(NPC + 63) / 64];
#pragma unroll
                        for (int k = 0; k < (NPC + 63) / 64; ++k) pv[k] = ((const uint4*)scr)[min(lane + 64 * k, NPC - 1)];
#pragma unroll
                        for (int k = 0; k < (NPC + 63) / 64; ++k)
                            if (lane + 64 * k < NPC) {
                                u32x4_a4 o; o.x = pv[k].x; o.y = pv[k].y; o.z = pv[k].z; o.w = pv[k].w;
                                *(u32x4_a4*)(gq + (size_t)(lane + 64 * k) * 16) = o;
                            }
                    }
                }
            }
        }
    }
}

// reset(mask, plan_idx_in) / observe / iou on the same tile machinery
template <class K, typename OT, int WPB, bool VAR>
__global__ __launch_bounds__(WPB * 64) void k_aux(const KArgs a) {
    constexpr int E = K::E;
    const int LD = VAR ? a.ld : K::D;
    const int lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x * WPB + (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(tile * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    uint32_t* lds = wave_lds<K, WPB>();
    Lane s;
    s.clear();
    s.r = 3; s.c = 3;
    if (active) s.unpack(a.hdr[env]);
    K::load_grid(lds, a, env0, nenv, lane);
    if (a.aux_op == AUX_IOU)
        for (int e = 0; e < nenv; ++e) K::load_plan(lds, a, e, __builtin_amdgcn_readlane(s.pidx, e), lane);
    if (a.aux_op == AUX_RESET) {
        const bool doit = active && (a.mask ? a.mask[env] != 0 : true);
        for (unsigned long long m = __ballot(doit); m; m &= m - 1) K::clear(lds, __ffsll(m) - 1, lane);
        if (doit) {
            const int episode = a.episode[env] + 1;
            int pidx;
            if (a.plan_idx_in) pidx = a.plan_idx_in[env];
            else if (a.plan_scalar >= 0) pidx = a.plan_scalar;
            else pidx = pick_plan<K>(a, env_keys(a.key_plan, (uint64_t)(a.env_id_base + env)), episode, a.static_plan);
            pidx = min(max(pidx, 0), a.num_plans - 1);
            K::reset(a, s, pidx);
            a.hdr[env] = s.pack();
            a.episode[env] = episode;
        }
        K::store_grid(lds, a, env0, nenv, lane);
    }
    if (a.aux_op == AUX_IOU) {
        const double v = K::iou(lds, s, active ? lane : 0);      // 3D: from the running sum kept in the header
        if (active) a.out_f64[env] = v;
        return;
    }
    // SNAC_TAIL_RECORD outside a step: reward 0, done = the env's pending-reset flag
    if (a.obs) emit_obs<K, OT, VAR>(lds, (OT*)a.obs + (size_t)env0 * LD, nenv, s, a, lane, StepOut{0, (s.flags & SNAC_FLAG_NEED_RESET) ? 1 : 0});
}

// environment_memory with its -1 frame, float64 [N][H][W]; one thread per cell
template <int KIND>
__global__ void k_export(const KArgs a, long long total) {
    constexpr int H = KIND == 1 ? 1 : 26, Wd = KIND == 1 ? 34 : 26, HW = KIND == 1 ? 2 : 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long env = i / (H * Wd);
        const int cellidx = (int)(i - env * (H * Wd));
        const int r = cellidx / Wd, c = cellidx - r * Wd;
        int v = a.frame_val;
        if (KIND == 1) {
            if (c >= HW && c < Wd - HW) v = ((const int16_t*)a.grid)[env * 32 + (c - HW)];
        } else if (r >= HW && r < H - HW && c >= HW && c < Wd - HW) {
            if (KIND == 2) v = (((const uint32_t*)a.grid)[env * 20 + (r - HW)] >> (c - HW)) & 1u;
            else v = ((const int16_t*)a.grid)[env * 400 + (r - HW) * 20 + (c - HW)];
        }
        a.out_f64[i] = (double)v;
    }
}

// ------------------------------------------------------------------------------------------------
// states in the reference's own format -> packed records: the inverse of k_export plus the header.  The MCTS variants hand
// (position, environment_memory, count_brick, count_step) tuples around (Env/2D/DMP_ENV_2D_dynamic_MCTS.py:88-91); this is
// how such a tuple enters the node pool.  One wave per state.  Values are clamped into the ranges the step kernels index with.
struct IArgs {
    int32_t m, pool, num_plans;
    const int32_t* dst_index;
    const int32_t* pos;        // [m][2] (row, col); 1D: (position, ignored)
    const int32_t* cb;
    const int32_t* cs;
    const int32_t* plan_idx;   // NULL: the destination row keeps its plan
    const int32_t* tb;         // NULL: total_brick of the plan row (plan_tb)
    const double* mem;         // [m][H][W] environment_memory with its frame
    int4* hdr;
    int32_t* episode;
    void* grid;
    const void* plans;
    const int16_t* plan_tb;
};

template <int KIND>
__global__ __launch_bounds__(256) void k_import(const IArgs g) {
    constexpr int CELLS = KIND == 1 ? 34 : 676, LO = KIND == 1 ? 2 : 3, HI = KIND == 1 ? 31 : 22;
    const int lane = threadIdx.x & 63;
    const int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i >= g.m) return;
    const size_t drow = row_of(g.dst_index, g.pool, i);
    const double* src = g.mem + (size_t)i * CELLS;
    Lane s;
    s.unpack(g.hdr[drow]);
    s.pidx = min(max(g.plan_idx ? g.plan_idx[i] : s.pidx, 0), g.num_plans - 1);
    s.tb = g.tb ? min(max(g.tb[i], -32768), 32767) : (int)g.plan_tb[s.pidx];
    int cross = 0;
    if (KIND == 1) {
        if (lane < 32) ((int16_t*)g.grid)[drow * 32 + lane] = lane < 30 ? (int16_t)min(max(llrint(src[lane + 2]), 0ll), 32767ll) : (int16_t)0;
    } else if (KIND == 2) {
        for (int row = 0; row < 20; ++row) {
            const bool on = lane < 20 && src[(row + 3) * 26 + 3 + lane] > 0.0;
            const unsigned long long bits = __ballot(on);
            if (lane == 0) ((uint32_t*)g.grid)[drow * 20 + row] = (uint32_t)bits & 0xFFFFFu;
        }
    } else {
        const int16_t* pl = (const int16_t*)g.plans + (size_t)s.pidx * 400;
        for (int cell = lane; cell < 400; cell += 64) {
            const int r = cell / 20, c = cell - r * 20;
            const int v = (int)min(max(llrint(src[(r + 3) * 26 + c + 3]), 0ll), 32767ll);
            ((int16_t*)g.grid)[drow * 400 + cell] = (int16_t)v;
            cross += min(v, (int)pl[cell]);
        }
        for (int off = 32; off > 0; off >>= 1) cross += __shfl_xor(cross, off);
    }
    if (lane == 0) {
        s.r = min(max(g.pos[2 * i], LO), HI);
        s.c = KIND == 1 ? 0 : min(max(g.pos[2 * i + 1], LO), HI);
        s.flags = 0;
        s.cb = min(max(g.cb[i], 0), 32767);
        s.cs = min(max(g.cs[i], 0), 3000);
        s.ep_ret = 0;
        s.cross = min(cross, 32767);
        g.hdr[drow] = s.pack();
        if (g.episode[drow] < 0) g.episode[drow] = 0;
    }
}

// equality_operator(o1, o2) of the MCTS variants (np.array_equal on two observations,
// Env/2D/DMP_ENV_2D_dynamic_MCTS.py:254-258; used to recognise an already-expanded child,
// script/MCTS/utils/mcts_Qvalue_dynamic.py:100-106): out[i] = all(a[ia[i]] == b[ib[i]]).  One wave per pair.
template <typename OT>
__global__ __launch_bounds__(256) void k_equal(const OT* a, const int32_t* ia, int rows_a, const OT* b, const int32_t* ib, int rows_b,
                                               int m, int D, uint8_t* out) {
    const int lane = threadIdx.x & 63;
    const int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i >= m) return;
    const OT* pa = a + row_of(ia, rows_a, i) * D;
    const OT* pb = b + row_of(ib, rows_b, i) * D;
    bool differ = false;
    for (int j = lane; j < D; j += 64) differ = differ || !(pa[j] == pb[j]);
    const unsigned long long any = __ballot(differ);
    if (lane == 0) out[i] = any ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// plan generators on the device (include/snac_hip.h "Plan generators"): the reference draws a fresh random plan per reset in
// its hindsight classes -- random triangles (Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py:37-59, cv2.polylines /
// cv2.fillPoly, redraw until the area exceeds 50 dense / 20 sparse) and random sine curves
// (Env/1D/DMP_Env_1D_dynamic_hindsight_replay.py:29-42) -- and ships 400 + 50 + 50 of them per dataset.  Here a launch
// writes `count` rows of the plan table, one wave per plan, from counter-RNG stream 2 or from explicit vertices.
struct PArgs {
    int32_t first, count, sparse, use_vertices;
    uint32_t key;
    int64_t id_base;
    const int8_t* vertices;   // [count][6] x0 y0 x1 y1 x2 y2 (clamped into 0..19) or NULL
    void* plans;
    int16_t* plan_tb;
    int32_t* area_out;        // [count] or NULL: cells set by the (last) attempt
};

// the triangle rasteriser, restating what cv2 does for the reference's call (thickness 1, LINE_8, shift 0); lane = plan row
// (y), result = the 20-bit mask of its columns (x).
//   outline  cv2.polylines -> LineIterator(leftToRight): start at the LEFT end point, one pixel per step along the longer
//            axis, a diagonal step whenever the running error dx - 2 dy has gone negative (an exact tie stays on the row).
//            Every lane walks the same pixels and keeps those of its row.
//   fill     cv2.fillPoly -> FillEdgeCollection: each non-horizontal edge runs from its upper end in 16.16 fixed point with
//            slope ((x1 - x0) << 16) / (y1 - y0) truncated towards zero; scanline y in [y_min, y_max) fills
//            ceil(left) .. floor(right) between its two active edges (plus the outline above).
// With these two rules every one of the 1000 2D plans the reference ships (drawn by its authors with cv2) is reproduced
// bit for bit from its three vertices (tests/test_plan_generators.py).
__device__ __forceinline__ uint32_t tri_row(int row, const int* vx, const int* vy, bool fill) {
    uint32_t m = 0;
    for (int e = 0; e < 3; ++e) {
        int x1 = vx[(e + 2) % 3], y1 = vy[(e + 2) % 3], x2 = vx[e], y2 = vy[e];
        if (x2 < x1) { int t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
        int dx = x2 - x1, dy = y2 - y1;
        const int sy = dy < 0 ? -1 : 1;
        dy = abs(dy);
        const bool steep = dy > dx;
        if (steep) { const int t = dx; dx = dy; dy = t; }
        int err = dx - 2 * dy, x = x1, y = y1;
        for (int i = 0; i <= dx; ++i) {                              // at most 20 pixels per edge
            if (y == row) m |= 1u << x;
            if (err < 0) { err += 2 * dx; if (steep) x += 1; else y += sy; }
            err -= 2 * dy;
            if (steep) y += sy; else x += 1;
        }
    }
    if (fill) {
        long long xs[2];
        int k = 0, ymin = 99, ymax = -99;
        for (int e = 0; e < 3; ++e) {
            int ax = vx[(e + 2) % 3], ay = vy[(e + 2) % 3], bx = vx[e], by = vy[e];
            if (ay == by) continue;
            if (ay > by) { int t = ax; ax = bx; bx = t; t = ay; ay = by; by = t; }
            ymin = min(ymin, ay); ymax = max(ymax, by);
            if (ay <= row && row < by && k < 2) xs[k++] = ((long long)ax << 16) + (long long)(row - ay) * (((long long)(bx - ax) * 65536) / (by - ay));
        }
        if (k == 2 && row >= ymin && row < ymax) {
            const long long lo = xs[0] < xs[1] ? xs[0] : xs[1], hi = xs[0] < xs[1] ? xs[1] : xs[0];
            const int c0 = max((int)((lo + 65535) >> 16), 0), c1 = min((int)(hi >> 16), 19);
            if (c1 >= c0) m |= ((2u << c1) - 1u) & ~((1u << c0) - 1u);
        }
    }
    return m;
}

// sin(x) for the sine-curve plans, specified operation by operation so that the CPU restatement gives the same bits (device
// and host libm sines differ in the last place, and a plan height is a ROUNDED multiple of it): n = rint(x * 2/pi); two-step
// Cody-Waite reduction r = x - n * pi/2; the fdlibm kernel polynomials on |r| <= pi/4, every multiply-add a fused one.
__device__ __forceinline__ double spec_sin(double x) {
    const double n = __builtin_rint(x * 0.63661977236758134308);
    double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-n, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(z * r, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cs = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)n & 3;
    const double v = (q & 1) ? cs : sn;
    return (q & 2) ? -v : v;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_make_plans(const PArgs g) {
    const int lane = threadIdx.x & 63;
    const int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i >= g.count) return;
    const size_t rowi = (size_t)(g.first + i);
    const EnvKeys pk = env_keys(g.key, (uint64_t)(g.id_base + (int64_t)rowi));
    if (KIND == 1) {
        // y[x] = rint(k1 * sin(2 pi / 30 * (k2 x + phase)) + 20), k1 in [3, 12), k2 in {1, 2, 3}, phase in [-pi, pi)
        const double u1 = (double)rng_word(pk, 0) * 2.3283064365386963e-10, u2 = (double)rng_word(pk, 2) * 2.3283064365386963e-10;
        const double k1 = __builtin_fma(9.0, u1, 3.0), phase = __builtin_fma(2.0, u2, -1.0) * 3.14159265358979311600;
        const int k2 = 1 + (int)__umulhi(rng_word(pk, 1), 3u);
        const double arg = 0.20943951023931953 * __builtin_fma((double)k2, (double)min(lane, 29), phase);
        const int y = (int)__builtin_rint(__builtin_fma(k1, spec_sin(arg), 20.0));
        int sum = lane < 30 ? y : 0;
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if (lane < 32) ((int16_t*)g.plans)[rowi * 32 + lane] = lane < 30 ? (int16_t)y : (int16_t)0;
        if (lane == 0) { g.plan_tb[rowi] = (int16_t)sum; if (g.area_out) g.area_out[i] = sum; }
        return;
    }
    // 3D plans also have an upper bound: script/HumanPlayerGUI/env/Env3D.py:360-364 redraws while area <= min or area >= 110
    // (the 3D datasets hold 6 x [51, 109] bricks)
    const int thr = g.sparse ? 20 : 50, amax = KIND == 3 ? 110 : 401;
    uint32_t m = 0;
    int area = 0;
    bool accepted = false;
    for (int attempt = 0; attempt < 64; ++attempt) {                // the reference redraws without bound; P(64 rejections) ~ 0
        int vx[3], vy[3];
        for (int v = 0; v < 3; ++v) {
            if (g.use_vertices) {
                vx[v] = min(max((int)g.vertices[(size_t)i * 6 + 2 * v], 0), 19);
                vy[v] = min(max((int)g.vertices[(size_t)i * 6 + 2 * v + 1], 0), 19);
            } else {
                const uint32_t w = rng_word(pk, (uint32_t)(attempt * 4 + v));
                vx[v] = (int)(((w & 0xffffu) * 20u) >> 16);
                vy[v] = (int)(((w >> 16) * 20u) >> 16);
            }
        }
        m = lane < 20 ? tri_row(lane, vx, vy, !g.sparse) : 0u;
        area = __popc(m);
        for (int off = 32; off > 0; off >>= 1) area += __shfl_xor(area, off);
        if ((area > thr && area < amax) || g.use_vertices) { accepted = true; break; }
    }
    // 64 rejections in a row (P ~ 0): the last triangle stands -- with at least one brick, and area_out says so (-area)
    const int tb_floor = accepted ? 0 : 1;
    if (KIND == 2) {
        if (lane < 20) ((uint32_t*)g.plans)[rowi * 20 + lane] = m;
        if (lane == 0) g.plan_tb[rowi] = (int16_t)max(area, 30);     // the 2D total_brick floor (:45-46)
    } else {
        int16_t* dst = (int16_t*)g.plans + rowi * 400;
        for (int r = 0; r < 20; ++r) {
            const uint32_t mr = (uint32_t)__shfl((int)m, r);
            if (lane < 20) dst[r * 20 + lane] = (int16_t)(((mr >> lane) & 1u) * 6);   // plan * z
        }
        if (lane == 0) g.plan_tb[rowi] = (int16_t)max(area * 6, tb_floor);
    }
    if (lane == 0 && g.area_out) g.area_out[i] = accepted ? area : -area;
}

// ------------------------------------------------------------------------------------------------
// replay sampling (the step after the env path: script/DQN/2d/DQN_2d_dynamic.py:122-124,145-166 keeps
// (s, a, r, s', plan) tuples in a python deque and re-assembles float32 minibatches on the host).  The rollout output
// ring obs[cap][N][D] already holds every s' -- and s is the previous tick's row, or the constant reset observation when
// the step opened an episode -- so sampling is a gather: one wave per sample, float32 out, plan expanded from the table.
struct GArgs {
    int32_t n, cap, batch, num_plans;
    int32_t ld, frame_val;     // row length (K::D, + the position tail) and frame value of the ring's layout
    int32_t tiled;             // obs is [ceil(n / 64)][cap][64][ld] instead of [cap][n][ld]
    const void* obs;
    const uint8_t* first;
    const int16_t* plan_idx;
    const int32_t* tick;
    const int32_t* env;
    const void* plans;
    float* s;
    float* s_next;
    float* plan_out;
};

template <int KIND, typename OT>
__global__ __launch_bounds__(256) void k_gather(const GArgs g) {
    // S samples per wave: the index loads of all of them first (lane u = sample u), then every row load of the group in flight
    // before the first store -- one sample per wave was a chain of three dependent loads with a single row in flight.
    constexpr int D = KIND == 1 ? 7 : 51, W = KIND == 1 ? 5 : 49, PC = KIND == 1 ? 30 : 400, S = 4;
    const int lane = threadIdx.x & 63;
    const int b0 = ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * S;
    if (b0 >= g.batch) return;
    const int ns = min(S, g.batch - b0);
    const int LD = g.ld;                                         // D, or D + the position tail (1 / 2 values)
    const OT* o = (const OT*)g.obs;
    // lane u < ns: the sample's slot, env, first-step flag and plan row
    int t = 0, i = 0, first = 0, p = 0;
    if (lane < ns) {
        t = min(max(g.tick[b0 + lane], 0), g.cap - 1);
        i = min(max(g.env[b0 + lane], 0), g.n - 1);
        const size_t cur = (size_t)t * g.n + i;
        first = g.first[cur] != 0;
        if (g.plan_out) p = min(max((int)g.plan_idx[cur], 0), g.num_plans - 1);
    }
    OT vcur[S], vprev[S];
    int fst[S];
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int tu = __builtin_amdgcn_readlane(t, u), iu = __builtin_amdgcn_readlane(i, u);
        fst[u] = __builtin_amdgcn_readlane(first, u);
        const int tp = tu == 0 ? g.cap - 1 : tu - 1;
        const size_t ocur = g.tiled ? ((size_t)(iu >> 6) * g.cap + tu) * 64 + (iu & 63) : (size_t)tu * g.n + iu;
        const size_t oprev = g.tiled ? ((size_t)(iu >> 6) * g.cap + tp) * 64 + (iu & 63) : (size_t)tp * g.n + iu;
        vcur[u] = (OT)0; vprev[u] = (OT)0;
        if (u < ns && lane < LD) {
            vcur[u] = o[ocur * LD + lane];
            if (!fst[u]) vprev[u] = o[oprev * LD + lane];
        }
    }
#pragma unroll
    for (int u = 0; u < S; ++u) {
        if (u < ns && lane < LD) {
            const size_t b = (size_t)(b0 + u);
            g.s_next[b * LD + lane] = (float)vcur[u];
            float sv;
            if (fst[u]) {   // reset observation: window at the start position over an empty grid, both scalar slots 0
                const int wi = lane / 7, wj = lane - 7 * wi;
                const bool frame = KIND == 1 ? lane < 2 : (wi < 3 || wj < 3);
                sv = (lane < W && frame) ? (float)g.frame_val : 0.0f;
                if (lane >= D) sv = KIND == 1 ? 2.0f : 3.0f;         // position tail: the start position
            } else {
                sv = (float)vprev[u];
            }
            g.s[b * LD + lane] = sv;
        }
    }
    if (g.plan_out) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            if (u >= ns) break;
            const int pu = __builtin_amdgcn_readlane(p, u);
            float* po = g.plan_out + (size_t)(b0 + u) * PC;
            if (KIND == 1) {
                if (lane < PC) po[lane] = (float)((const int16_t*)g.plans)[pu * 32 + lane];
            } else {
                // four consecutive cells per lane (a row of 20 holds five such groups): one 16-byte store each, 100 lanes a plan
                for (int q = lane; q < PC / 4; q += 64) {
                    const int c = q * 4;
                    float4 v;
                    if (KIND == 2) {
                        const int row = c / 20, col = c - row * 20;
                        const uint32_t w = ((const uint32_t*)g.plans)[pu * 20 + row] >> col;
                        v = make_float4((float)(w & 1u), (float)((w >> 1) & 1u), (float)((w >> 2) & 1u), (float)((w >> 3) & 1u));
                    } else {
                        const short4 h = *(const short4*)((const int16_t*)g.plans + (size_t)pu * 400 + c);
                        v = make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
                    }
                    *(float4*)(po + c) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// the layout flags of the descriptor (include/snac_hip.h "Observation-layout variants")
int check_layout(const snac_env_desc* d) {
    if (d->frame_value != 0 && d->frame_value != -1 && d->frame_value != 2) return fail(SNAC_ERR_ARG, "frame_value must be -1 (or 0) or 2");
    if (d->frame_value == 2 && d->kind == SNAC_ENV_3D) return fail(SNAC_ERR_UNSUPPORTED, "frame_value 2 is a 1D / 2D layout (the 3D rules test the frame for -1)");
    if (d->obs_scalars < SNAC_SCALARS_DEFAULT || d->obs_scalars > SNAC_SCALARS_NORM) return fail(SNAC_ERR_ARG, "unknown obs_scalars");
    if (d->obs_tail & ~(SNAC_TAIL_POSITION | SNAC_TAIL_PLAN | SNAC_TAIL_RECORD)) return fail(SNAC_ERR_ARG, "unknown bits in obs_tail");
    if (d->reserved != 0) return fail(SNAC_ERR_ARG, "snac_env_desc.reserved must be 0");
    return SNAC_OK;
}
int base_obs_dim(int kind) { return kind == SNAC_ENV_1D ? 7 : 51; }
int tail_len(int kind, int tail) {
    return ((tail & SNAC_TAIL_POSITION) ? (kind == SNAC_ENV_1D ? 1 : 2) : 0) + ((tail & SNAC_TAIL_PLAN) ? (kind == SNAC_ENV_1D ? 30 : 400) : 0) +
           ((tail & SNAC_TAIL_RECORD) ? 8 : 0);
}

int check_common(const snac_env_desc* d, const snac_state* st) {
    if (!d || !st) return fail(SNAC_ERR_ARG, "null desc/state");
    if (d->kind < SNAC_ENV_1D || d->kind > SNAC_ENV_3D) return fail(SNAC_ERR_ARG, "unknown env kind");
    if (d->num_envs <= 0) return fail(SNAC_ERR_ARG, "num_envs must be positive");
    if (d->num_plans <= 0 || d->num_plans > 32767) return fail(SNAC_ERR_ARG, "num_plans out of range");
    if (d->static_plan < 0 || d->static_plan >= d->num_plans) return fail(SNAC_ERR_ARG, "static_plan out of range");
    if (d->obs_dtype != SNAC_OBS_F64 && d->obs_dtype != SNAC_OBS_F32) return fail(SNAC_ERR_ARG, "unknown obs_dtype");
    // the running return is an int16 and a step pays at most 10: 3000 steps cannot overflow it (the reference: <= 1300)
    if (d->total_step < 0 || d->total_step > 3000) return fail(SNAC_ERR_ARG, "total_step out of range (0..3000)");
    if (d->rules & ~(SNAC_RULE_BRICK_GT | SNAC_RULE_TIME_GT)) return fail(SNAC_ERR_ARG, "unknown bits in rules");
    if (int rc = check_layout(d)) return rc;
    if (!st->hdr || !st->episode || !st->grid || !st->plans || !st->plan_tb || !st->stat_episodes || !st->stat_return ||
        !st->stat_iou_fx)
        return fail(SNAC_ERR_ARG, "null pointer in snac_state");
    return SNAC_OK;
}

KArgs make_args(const snac_env_desc* d, const snac_state* st) {
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = d->num_envs; a.num_plans = d->num_plans; a.static_plan = d->static_plan;
    a.total_step = d->total_step > 0 ? d->total_step
                                     : (d->kind == SNAC_ENV_1D ? 750 : (d->kind == SNAC_ENV_2D ? 600 : (d->dynamic ? 1000 : 1300)));
    a.brick_gt = (d->rules & SNAC_RULE_BRICK_GT) ? 1 : 0;
    a.ts_done = a.total_step + ((d->rules & SNAC_RULE_TIME_GT) ? 1 : 0);
    a.key_step = stream_key(d->seed, 0); a.key_plan = stream_key(d->seed, 1);
    a.env_id_base = d->env_id_base;
    a.hdr = (int4*)st->hdr; a.episode = st->episode; a.grid = st->grid; a.plans = st->plans; a.plan_tb = st->plan_tb;
    a.stat_episodes = st->stat_episodes; a.stat_return = st->stat_return; a.stat_iou_fx = st->stat_iou_fx;
    a.plan_scalar = -1;
    a.frame_val = d->frame_value == 2 ? 2 : -1;
    a.sc_norm = d->obs_scalars == SNAC_SCALARS_DEFAULT ? (d->dynamic ? 1 : 0) : (d->obs_scalars == SNAC_SCALARS_NORM ? 1 : 0);
    a.tail = d->obs_tail;
    a.ld = base_obs_dim(d->kind) + tail_len(d->kind, d->obs_tail);
    a.variant = (a.frame_val != -1 || a.sc_norm != (d->dynamic ? 1 : 0) || a.tail != 0) ? 1 : 0;
    return a;
}

enum Op { OP_ROLLOUT, OP_AUX, OP_TRANSITION };

// which kernel the calling thread's last launch went to (snac_last_kernel(): bench.py and the tests name the kernel they measured
// from here instead of restating the dispatch conditions)
thread_local const char* g_kernel = "";

template <class K, typename OT, int WPB>
void launch_k(Op op, const KArgs& a, hipStream_t s) {
    const int tiles = (a.n + K::E - 1) / K::E;
    const dim3 grid((unsigned)((tiles + WPB - 1) / WPB)), block(WPB * 64);
    if (a.variant) {                                             // layout variants: their own instantiations, the canonical ones stay lean
        if (op == OP_ROLLOUT) hipLaunchKernelGGL((k_rollout<K, OT, WPB, true, true>), grid, block, 0, s, a);
        else if (op == OP_TRANSITION) hipLaunchKernelGGL((k_transition<K, OT, WPB, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_aux<K, OT, WPB, true>), grid, block, 0, s, a);
        return;
    }
    if (op == OP_ROLLOUT) {
        if (a.actions || a.step_size) hipLaunchKernelGGL((k_rollout<K, OT, WPB, true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_rollout<K, OT, WPB, false, false>), grid, block, 0, s, a);
    }
    else if (op == OP_TRANSITION) hipLaunchKernelGGL((k_transition<K, OT, WPB, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_aux<K, OT, WPB, false>), grid, block, 0, s, a);
}

template <template <bool, int> class KT, bool DYN, int E, int WPB>
void launch_dt(Op op, int obs_dtype, const KArgs& a, hipStream_t s) {
    if (obs_dtype == SNAC_OBS_F32) launch_k<KT<DYN, E>, float, WPB>(op, a, s);
    else launch_k<KT<DYN, E>, double, WPB>(op, a, s);
}

// tile size: enough tiles to give every SIMD of the 256 CUs a few waves; SNAC_TILE overrides (tuning)
int pick_tile(int kind, int n) {
    static const int forced = [] { const char* e = std::getenv("SNAC_TILE"); return e ? std::atoi(e) : 0; }();
    if (kind == SNAC_ENV_3D) return forced == 16 ? 16 : 8;   // 8: 17 KB of LDS per wave, 9 waves per CU; measured +10-15 % over 16
    if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;
    // measured per kind (tools/ab_time.py sweeps): 2D wants large tiles early (E x 408-byte store runs),
    // 1D's 56-byte rows do not care and prefer more, smaller waves
    const int shift = kind == SNAC_ENV_1D ? 1 : 0;
    if (n >= (64 * 1024) << shift) return 64;   // 2D: >= one wave per SIMD on 256 CUs; best at N = 65536 (profiles/)
    if (n >= (32 * 1024) << shift) return 32;
    if (n >= (16 * 1024) << shift) return 16;
    return 8;                        // small batches: one-wave blocks of 8 envs, so that 4096 envs still reach every CU
}

// SNAC_3D_PIPELINE=0 keeps every launch on the generic tile kernels (A/B timing, tests of both paths)
bool pipeline_off() {
    static const bool off = [] { const char* e = std::getenv("SNAC_3D_PIPELINE"); return e && e[0] == '0'; }();
    return off;
}

template <bool DYN, typename OT, int WPB>
void launch_roll3d_w(const KArgs& a, hipStream_t s) {
    const int tiles = (a.n + 7) / 8, blocks = (tiles + WPB - 1) / WPB;
    const dim3 grid((unsigned)(((blocks + 7) / 8) * 8)), block(WPB * 64);   // a multiple of 8: the XCD remap covers every tile
    if (a.actions || a.step_size) hipLaunchKernelGGL((k_rollout3d<DYN, OT, WPB, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_rollout3d<DYN, OT, WPB, false>), grid, block, 0, s, a);
}
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

// 3D rollouts by blocks of 64 envs; SNAC_3D_BLOCK=0 keeps them on k_rollout3d (A/B timing, tests of both paths)
bool roll3db_ok(const KArgs& a, bool f32) {
    static const bool off = [] { const char* e = std::getenv("SNAC_3D_BLOCK"); return e && e[0] == '0'; }();
    static const int nmin = [] { const char* e = std::getenv("SNAC_3D_BLOCK_MIN"); return e ? std::atoi(e) : -1; }();   // (tuning)
    // where k_rollout3d's one-wave blocks stop being faster: float64 rows 4096 envs 1.02 against 1.04 ms, 6144 level, 8192 1.08 against
    // 1.04; float32 rows 4096 envs 1.02 against 0.97 already
    const int lim = nmin >= 0 ? nmin : (f32 ? 4096 : 6144);
    return !off && a.n >= lim && !a.variant && (a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) && a.num_plans <= TB_MAX && (a.n & 3) == 0 &&
           (((uintptr_t)a.obs) & 15) == 0 && !pipeline_off();
}
template <bool DYN, typename OT>
void launch_roll3db_w(const KArgs& a, hipStream_t s) {
    const int blocks = (a.n + 63) / 64;
    const dim3 grid((unsigned)(((blocks + 7) / 8) * 8)), block(576);   // a multiple of 8: the XCD remap covers every block
    if (a.actions || a.step_size) hipLaunchKernelGGL((k_rollout3db<DYN, OT, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_rollout3db<DYN, OT, false>), grid, block, 0, s, a);
}
void launch_roll3db(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll3db_w<true, float>(a, s) : launch_roll3db_w<true, double>(a, s);
    else f32 ? launch_roll3db_w<false, float>(a, s) : launch_roll3db_w<false, double>(a, s);
}

// SNAC_2D_STAGE=0 keeps 2D rollouts on the tile kernel (A/B timing, tests of both paths)
bool stage2d_off() {
    static const bool off = [] { const char* e = std::getenv("SNAC_2D_STAGE"); return e && e[0] == '0'; }();
    return off;
}
bool roll2d_ok(const KArgs& a, int E) {
    return E == 64 && (a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) &&
           (a.n & 3) == 0 && ((uintptr_t)a.obs & 15) == 0 && !pipeline_off() && !stage2d_off();
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
void launch_roll2d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll2d_w<true, float>(a, s) : launch_roll2d_w<true, double>(a, s);
    else f32 ? launch_roll2d_w<false, float>(a, s) : launch_roll2d_w<false, double>(a, s);
}

// time-parallel 2D rollouts (one wave per env, lane = tick): small and middle batches, where the lane-per-env kernels are bound by the
// chain of their ticks (0.5-0.7 ms per 600 ticks at every N <= 16 384) or leave CUs empty (one wave of 64 envs per CU at N = 16 384).
// Where it stops paying was measured on trajectory memory (profiles/r04_2d_midrange.txt, part 3): float64 rows up to 19 456 envs --
// except just below 16 384, where the tile kernel's 256 waves fill the chip exactly (5.8 against 5.55 TB/s) --, float32 rows up to
// 30 719 (16 384 envs: 4.2 against 2.9 TB/s); batches whose per-tick runs are not 16-byte pieces (odd N) only up to 8192.
// SNAC_2D_TP=0 keeps every 2D rollout off this kernel (A/B timing, tests of both paths), SNAC_2D_TP_MAX=n replaces the limits by n
bool roll2dt_ok(const KArgs& a, bool f32) {
    static const bool off = [] { const char* e = std::getenv("SNAC_2D_TP"); return e && e[0] == '0'; }();
    static const int nmax = [] { const char* e = std::getenv("SNAC_2D_TP_MAX"); return e ? std::atoi(e) : 0; }();   // (tuning)
    if (off || !(a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) || pipeline_off()) return false;
    if (a.variant) {
        // the layout variants (rows of a.ld values: k_rollout2dt<.., VAR>): whole groups of four envs and 16-byte pieces only.  Where
        // the lane-per-env kernels take over again was measured with the 451-value rows of the PPO copies (profiles/r04_2d_layouts.txt)
        // (profiles/r04_2d_layouts.txt, part 3): 6.0-6.4 TB/s from 1024 envs on against k_rollout2d's 5.45 at 49 152 envs and 7.2 at 65 536;
        // short rows (no plan tail: 53 .. 61 values) level off at 5.8e9 env-steps/s and hand over near 6144 envs
        static const int vmax = [] { const char* e = std::getenv("SNAC_2D_TP_VAR_MAX"); return e ? std::atoi(e) : 0; }();   // (tuning)
        const int lim = vmax ? vmax : ((a.tail & SNAC_TAIL_PLAN) ? 49152 : 6144);
        return (a.n & 3) == 0 && (((uintptr_t)a.obs) & 15) == 0 && a.n <= lim;
    }
    if (nmax) return a.n <= nmax;
    const size_t rowb = (size_t)K2D<true, 64>::D * (f32 ? 4 : 8);
    const bool pieces = (((uintptr_t)a.obs) & 15) == 0 && (a.obs_mode == SNAC_OBS_TILED || (((size_t)a.n * rowb) & 15) == 0);
    if (!pieces) return a.n <= 8192;
    if (f32) return a.n < 30720;
    return a.n <= 15872 || (a.n > 16384 && a.n <= 19456);
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
    static const int emin = [] { const char* e = std::getenv("SNAC_2D_TP_EB8"); return e ? std::atoi(e) : 1025; }();   // (tuning)
    if (a.n >= emin) launch_roll2dt_e<DYN, OT, 8>(a, s);       // 8 envs per block: runs of 3264 / 1632 bytes per tick
    else launch_roll2dt_e<DYN, OT, 4>(a, s);                        // up to 1024 envs: a block per CU first (1536 envs: 0.097 against 0.086 ms)
}
void launch_roll2dt(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll2dt_w<true, float>(a, s) : launch_roll2dt_w<true, double>(a, s);
    else f32 ? launch_roll2dt_w<false, float>(a, s) : launch_roll2dt_w<false, double>(a, s);
}

// time-parallel 1D rollouts (one wave per env, lane = tick).  Its rate levels off at 6-7e10 env-steps/s (instruction issue: ~9 per
// env-step), the tile kernel's keeps growing with the batch: float64 rows 49 152 envs 0.54 against 0.68 ms per 750 ticks, 65 536
// 0.72-0.87 against 0.72; float32 rows 65 536 envs 0.65 against 0.73, 131 072 1.35 against 0.94 (profiles/r03_1d_time_parallel.txt).
// SNAC_1D_TP=0 keeps every 1D rollout on the tile kernel (A/B timing)
bool roll1dt_ok(const KArgs& a, bool f32) {
    static const bool off = [] { const char* e = std::getenv("SNAC_1D_TP"); return e && e[0] == '0'; }();
    static const int nmax = [] { const char* e = std::getenv("SNAC_1D_TP_MAX"); return e ? std::atoi(e) : 0; }();   // (tuning)
    const int lim = nmax ? nmax : (f32 ? 65536 : 49152);
    if (off || !(a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) || pipeline_off()) return false;
    if (a.variant) {
        // the layout variants (k_rollout1dt<.., VAR>: blocks of four envs, one per CU): whole groups of four envs and 16-byte pieces;
        // it levels off at 1.2e10 env-steps/s with the 37-value PPO rows (the tile kernel: 8.6e9 at 65 536 envs) and at 4.4-5.1e10 with the
        // 8-value L-Net rows (the tile kernel: 3.1e10 at 65 536 envs) (profiles/r04_1d_layouts.txt)
        static const int vmax = [] { const char* e = std::getenv("SNAC_1D_TP_VAR_MAX"); return e ? std::atoi(e) : 65536; }();   // (tuning)
        return (a.n & 3) == 0 && (((uintptr_t)a.obs) & 15) == 0 && a.n <= vmax;
    }
    return a.n <= lim;
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
    static const int emin = [] { const char* e = std::getenv("SNAC_1D_TP_EB16"); return e ? std::atoi(e) : 3584; }();   // (tuning)
    if (a.n >= emin) launch_roll1dt_e<DYN, OT, 16>(a, s);      // 16 envs per block: runs of 896 / 448 bytes per tick (3072 envs: 0.048 against 0.041 ms; 3584: level)
    else launch_roll1dt_e<DYN, OT, 4>(a, s);                        // small batches: more blocks than CUs first
}
void launch_roll1dt(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll1dt_w<true, float>(a, s) : launch_roll1dt_w<true, double>(a, s);
    else f32 ? launch_roll1dt_w<false, float>(a, s) : launch_roll1dt_w<false, double>(a, s);
}

// SNAC_STEP_STAGE=0 keeps snac_step on k_transition2d / k_transition3d (A/B timing, tests of both paths)
bool step_stage_ok(const KArgs& a) {
    static const bool off = [] { const char* e = std::getenv("SNAC_STEP_STAGE"); return e && e[0] == '0'; }();
    return !off && !a.src_index && !a.dst_index && (a.n & 3) == 0 && ((uintptr_t)a.obs & 15) == 0;
}
// the layout variants on k_step2d<.., VAR>: 64 envs per wave are 64 rows of kilobytes per wave -- batches large enough to fill the
// CUs that way; below, the 8-env tiles of k_transition spread the rows over more waves.  PPO rows (451 values), us per tick at
// 40 960 / 49 152 / 65 536 envs: 36.7 / 36.8 / 37.7 against 35.4 / 40.8 / 88 (float32 rows at 32 768: 26.7 against 32.0); L-Net rows
// 24 576 / 32 768 / 65 536: 8.3 / 8.8 / 9.9 against 8.5 / 10.5 / 20.2 (profiles/r04_step_layouts.txt).  SNAC_STEP_VAR_MIN=n replaces
// the limits.
bool step_var_ok(const KArgs& a, bool f32) {
    static const int nmin = [] { const char* e = std::getenv("SNAC_STEP_VAR_MIN"); return e ? std::atoi(e) : 0; }();   // (tuning)
    if (nmin) return a.n >= nmin;
    if (!(a.tail & SNAC_TAIL_PLAN)) return a.n >= 24576;
    // rows with the plan tail: half-filled tiles for 24 577 .. 32 768 envs (one round of 1024 waves: 22.7 us at 32 768 envs against
    // k_transition's 28.8 and the full tiles' 36.6; float32 18.2 / 23.7 / 26.7), full tiles from 45 056 (float32: above 32 768)
    return (a.n > 24576 && a.n <= 32768) || a.n >= (f32 ? 32769 : 45056);
}
// half-filled tiles for rows with the plan tail up to 32 768 envs (SNAC_STEP_VAR_HALF=0 / 1 forces)
bool step_var_half(const KArgs& a) {
    static const int force = [] { const char* e = std::getenv("SNAC_STEP_VAR_HALF"); return e ? std::atoi(e) : -1; }();   // (tuning)
    if (force >= 0) return force != 0;
    return (a.tail & SNAC_TAIL_PLAN) && a.n <= 32768;
}
// 3D: k_step3d<.., VAR>.  PPO rows, us per tick at 16 384 / 65 536 envs: 37.8 / 42.8 against k_transition's 34.2 / 112.6 (float32 rows
// at 524 288 envs: 403 against 1032) (profiles/r04_step_layouts.txt).  SNAC_STEP_VAR3_MIN=n moves the limit.
bool step_var3_ok(const KArgs& a) {
    static const int nmin = [] { const char* e = std::getenv("SNAC_STEP_VAR3_MIN"); return e ? std::atoi(e) : 24576; }();   // (tuning)
    return a.n >= nmin && a.frame_val == -1;
}
template <int KIND>
void launch_step_tile(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    const int tiles = (a.n + 63) / 64;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (KIND == 2 && a.variant && step_var_half(a)) {
        const dim3 grid2((unsigned)(((a.n + 31) / 32 + 3) / 4));
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step2d<true, float, 4, true, 32>), grid2, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<true, double, 4, true, 32>), grid2, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step2d<false, float, 4, true, 32>), grid2, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<false, double, 4, true, 32>), grid2, block, 0, s, a); }
    } else if (KIND == 2 && a.variant) {
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step2d<true, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<true, double, 4, true>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step2d<false, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<false, double, 4, true>), grid, block, 0, s, a); }
    } else if (KIND == 2) {
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step2d<true, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<true, double, 4>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step2d<false, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step2d<false, double, 4>), grid, block, 0, s, a); }
    } else if (a.variant) {
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step3d<true, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3d<true, double, 4, true>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step3d<false, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3d<false, double, 4, true>), grid, block, 0, s, a); }
    } else {
        if (dyn) { if (f32) hipLaunchKernelGGL((k_step3d<true, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3d<true, double, 4>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_step3d<false, float, 4>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_step3d<false, double, 4>), grid, block, 0, s, a); }
    }
}

void launch_trans3d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    const int tiles = (a.n + 31) / 32;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (!a.src_index && !a.dst_index) {   // identity rows (snac_step, or a transition on rows i -> i)
        if (dyn) { if (f32) hipLaunchKernelGGL((k_transition3d<true, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_transition3d<true, double, 4, true>), grid, block, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_transition3d<false, float, 4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_transition3d<false, double, 4, true>), grid, block, 0, s, a); }
        return;
    }
    // gathered / scattered rows (tree edges): the records through LDS, every memory instruction wide (k_edges3d); SNAC_EDGES3D=0 keeps
    // them on k_transition3d (A/B timing, tests of both paths)
    static const bool edges_off = [] { const char* e = std::getenv("SNAC_EDGES3D"); return e && e[0] == '0'; }();
    if (!edges_off && a.obs && (a.n & 3) == 0 && ((uintptr_t)a.obs & 15) == 0) {
        g_kernel = "k_edges3d";
        const dim3 g2((unsigned)((tiles + 1) / 2)), b2(128);     // two waves per block: 51 KB of LDS, three blocks per CU
        if (dyn) { if (f32) hipLaunchKernelGGL((k_edges3d<true, float, 2>), g2, b2, 0, s, a); else hipLaunchKernelGGL((k_edges3d<true, double, 2>), g2, b2, 0, s, a); }
        else { if (f32) hipLaunchKernelGGL((k_edges3d<false, float, 2>), g2, b2, 0, s, a); else hipLaunchKernelGGL((k_edges3d<false, double, 2>), g2, b2, 0, s, a); }
        return;
    }
    if (dyn) { if (f32) hipLaunchKernelGGL((k_transition3d<true, float, 4, false>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_transition3d<true, double, 4, false>), grid, block, 0, s, a); }
    else { if (f32) hipLaunchKernelGGL((k_transition3d<false, float, 4, false>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_transition3d<false, double, 4, false>), grid, block, 0, s, a); }
}

// E edges per wave.  The kernel is bound by HBM traffic from N = 2^19 down to where the launch itself dominates; 32 edges per
// wave were 4 % ahead of 64 there (two rounds of waves: the second round's loads run under the first round's row stores),
// small batches take 16 so that a step() on 4096 envs is still 256 waves.  SNAC_T2D_E overrides (tuning).
template <bool DYN, typename OT>
void launch_trans2d_e(const KArgs& a, hipStream_t s) {
    static const int forced = [] { const char* e = std::getenv("SNAC_T2D_E"); return e ? std::atoi(e) : 0; }();
    const int E = (forced == 16 || forced == 32 || forced == 64) ? forced : (a.n >= 65536 ? 32 : 16);
    const int tiles = (a.n + E - 1) / E;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (E == 64) hipLaunchKernelGGL((k_transition2d<DYN, OT, 4, 64>), grid, block, 0, s, a);
    else if (E == 32) hipLaunchKernelGGL((k_transition2d<DYN, OT, 4, 32>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_transition2d<DYN, OT, 4, 16>), grid, block, 0, s, a);
}
void launch_trans2d(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_trans2d_e<true, float>(a, s) : launch_trans2d_e<true, double>(a, s);
    else f32 ? launch_trans2d_e<false, float>(a, s) : launch_trans2d_e<false, double>(a, s);
}

template <template <bool, int> class KT, int WPB>
void launch_tile(Op op, bool dyn, int E, int obs_dtype, const KArgs& a, hipStream_t s) {
    if (E == 64) dyn ? launch_dt<KT, true, 64, WPB>(op, obs_dtype, a, s) : launch_dt<KT, false, 64, WPB>(op, obs_dtype, a, s);
    else if (E == 32) dyn ? launch_dt<KT, true, 32, WPB>(op, obs_dtype, a, s) : launch_dt<KT, false, 32, WPB>(op, obs_dtype, a, s);
    else if (E == 16) dyn ? launch_dt<KT, true, 16, WPB>(op, obs_dtype, a, s) : launch_dt<KT, false, 16, WPB>(op, obs_dtype, a, s);
    else dyn ? launch_dt<KT, true, 8, 1>(op, obs_dtype, a, s) : launch_dt<KT, false, 8, 1>(op, obs_dtype, a, s);
}

int launch(Op op, const snac_env_desc* d, const KArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool dyn = d->dynamic != 0;
    const int E = pick_tile(d->kind, a.n);
    const char* const tile_name = op == OP_ROLLOUT ? "k_rollout" : (op == OP_TRANSITION ? "k_transition" : "k_aux");
    g_kernel = tile_name;
    switch (d->kind) {
        case SNAC_ENV_1D:
            // rollouts that write every row: the time-parallel kernel while its rate beats the tile kernel's (lane-per-env transition)
            if (op == OP_ROLLOUT && roll1dt_ok(a, d->obs_dtype == SNAC_OBS_F32)) { g_kernel = "k_rollout1dt"; launch_roll1dt(d, a, s); break; }
            launch_tile<K1D, 4>(op, dyn, E, d->obs_dtype, a, s); break;
        case SNAC_ENV_2D:
            if (op == OP_TRANSITION && !pipeline_off() && step_stage_ok(a) && (!a.variant || step_var_ok(a, d->obs_dtype == SNAC_OBS_F32))) { g_kernel = "k_step2d"; launch_step_tile<2>(d, a, s); break; }
            if (op == OP_TRANSITION && !a.variant && !pipeline_off()) { g_kernel = "k_transition2d"; launch_trans2d(d, a, s); break; }
            // float32 rows from N = 32 768: 512 staged waves (1.05 -> 0.74 ms per 600 ticks); float64 rows there are level (1.26-1.60 ms
            // by box for either kernel) and stay on 32-env tiles
            if (op == OP_ROLLOUT && roll2dt_ok(a, d->obs_dtype == SNAC_OBS_F32)) { g_kernel = "k_rollout2dt"; launch_roll2dt(d, a, s); break; }
            {
                // float64 rows from 32 769 envs (40 960: 1.56 against 1.99 ms per 600 ticks for the tile kernel's 32-env tiles, 49 152: 1.75 / 2.08,
                // 57 344: 2.05 / 2.26; at 32 768 and below the tile kernel's 1024 waves of 32 envs are ahead: 1.25 against 1.43), float32
                // rows from 32 768 (0.81 against 1.06); half-filled tiles -- 32 envs per wave on twice the waves -- were tried for
                // 24 576 .. 32 768 envs and lose on trajectory memory (profiles/r04_2d_midrange.txt).  SNAC_2D_STAGE_MIN=n moves the limit.
                static const int stage_min = [] { const char* e = std::getenv("SNAC_2D_STAGE_MIN"); return e ? std::atoi(e) : 0; }();
                const int from = stage_min ? stage_min : (d->obs_dtype == SNAC_OBS_F32 ? 32768 : 32769);
                if (op == OP_ROLLOUT && roll2d_ok(a, a.n >= from ? 64 : E)) { g_kernel = "k_rollout2d"; launch_roll2d(d, a, s); break; }
            }
            launch_tile<K2D, 4>(op, dyn, E, d->obs_dtype, a, s); break;
        default:   // 3D: 2.1 KB of LDS per env -> tiles of 16 (or 8 for small batches: two waves per SIMD sooner)
            if (op == OP_ROLLOUT && roll3db_ok(a, d->obs_dtype == SNAC_OBS_F32)) { g_kernel = "k_rollout3db"; launch_roll3db(d, a, s); break; }
            if (op == OP_ROLLOUT && E == 8 && !a.variant && (a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) && a.num_plans <= TB_MAX && !pipeline_off()) { g_kernel = "k_rollout3d"; launch_roll3d(d, a, s); break; }
            if (op == OP_TRANSITION && !pipeline_off() && step_stage_ok(a) && (!a.variant || step_var3_ok(a))) { g_kernel = "k_step3d"; launch_step_tile<3>(d, a, s); break; }
            if (op == OP_TRANSITION && !a.variant && !pipeline_off()) { g_kernel = "k_transition3d"; launch_trans3d(d, a, s); break; }
            if (E == 8 && a.n < 8192) dyn ? launch_dt<K3D, true, 8, 1>(op, d->obs_dtype, a, s) : launch_dt<K3D, false, 8, 1>(op, d->obs_dtype, a, s);
            else if (E == 8) dyn ? launch_dt<K3D, true, 8, 4>(op, d->obs_dtype, a, s) : launch_dt<K3D, false, 8, 4>(op, d->obs_dtype, a, s);
            else dyn ? launch_dt<K3D, true, 16, 2>(op, d->obs_dtype, a, s) : launch_dt<K3D, false, 16, 2>(op, d->obs_dtype, a, s);
            break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return SNAC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

int snac_version(void) { return SNAC_ABI_VERSION; }

const char* snac_last_error(void) { return g_err; }

const char* snac_last_kernel(void) { return g_kernel; }

int snac_stream_sync(void* stream) {
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? SNAC_OK : fail_hip(e, "hipStreamSynchronize");
}

int snac_env_sizes(int kind, int dynamic, snac_sizes* o) {
    if (!o) return fail(SNAC_ERR_ARG, "null out");
    std::memset(o, 0, sizeof(*o));
    if (kind == SNAC_ENV_1D) {
        *o = snac_sizes{7, 3, 750, 2, 1, 34, 1, 30, 32, 2, 32, 2};
    } else if (kind == SNAC_ENV_2D) {
        *o = snac_sizes{51, 5, 600, 3, 26, 26, 20, 20, 20, 4, 20, 4};
    } else if (kind == SNAC_ENV_3D) {
        *o = snac_sizes{51, 8, dynamic ? 1000 : 1300, 3, 26, 26, 20, 20, 400, 2, 400, 2};
    } else {
        return fail(SNAC_ERR_ARG, "unknown env kind");
    }
    return SNAC_OK;
}

int snac_obs_dim(const snac_env_desc* d) {
    if (!d) return fail(SNAC_ERR_ARG, "null desc");
    if (d->kind < SNAC_ENV_1D || d->kind > SNAC_ENV_3D) return fail(SNAC_ERR_ARG, "unknown env kind");
    if (int rc = check_layout(d)) return rc;
    return base_obs_dim(d->kind) + tail_len(d->kind, d->obs_tail);
}

int snac_reset_scalar(const snac_env_desc* d, const snac_state* st, int32_t plan_idx, void* obs, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (plan_idx < 0 || plan_idx >= d->num_plans) return fail(SNAC_ERR_ARG, "plan_idx out of range");
    KArgs a = make_args(d, st);
    a.aux_op = AUX_RESET; a.plan_scalar = plan_idx; a.obs = obs;
    return launch(OP_AUX, d, a, stream);
}

int snac_step_scalar(const snac_env_desc* d, const snac_state* st, uint32_t t, int32_t action, int32_t step_size, int auto_reset,
                     void* obs, float* reward, uint8_t* done, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    KArgs a = make_args(d, st);
    a.pool = d->num_envs; a.stats_on = 1;
    a.T = 1; a.t0 = t; a.auto_reset = auto_reset ? 1 : 0; a.obs_mode = obs ? SNAC_OBS_ALL : SNAC_OBS_NONE;
    a.use_scalar = 1; a.act_scalar = action; a.k_scalar = step_size;
    a.obs = obs; a.reward = reward; a.done = done;
    return launch(OP_TRANSITION, d, a, stream);
}

int snac_reset(const snac_env_desc* d, const snac_state* st, const uint8_t* mask, const int16_t* plan_idx_in, void* obs,
               void* stream) {
    if (int rc = check_common(d, st)) return rc;
    KArgs a = make_args(d, st);
    a.aux_op = AUX_RESET; a.mask = mask; a.plan_idx_in = plan_idx_in; a.obs = obs;
    return launch(OP_AUX, d, a, stream);
}

int snac_rollout_rec(const snac_env_desc* d, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                     const int8_t* step_size, int obs_mode, void* obs, float* reward, uint8_t* done,
                     const snac_rollout_record* rec, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (T < 0) return fail(SNAC_ERR_ARG, "T must be >= 0");
    if (obs_mode < SNAC_OBS_NONE || obs_mode > SNAC_OBS_TILED) return fail(SNAC_ERR_ARG, "unknown obs_mode");
    if (obs_mode != SNAC_OBS_NONE && !obs) return fail(SNAC_ERR_ARG, "obs_mode set but obs is null");
    if (T == 0) return SNAC_OK;
    KArgs a = make_args(d, st);
    a.T = T; a.t0 = t0; a.auto_reset = 1; a.obs_mode = obs_mode;
    a.tiled_T = T; a.tiled_t0 = 0;
    a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    if (rec) {
        a.actions_out = rec->actions; a.step_size_out = rec->step_size; a.plan_idx_out = rec->plan_idx; a.first_out = rec->first;
    }
    return launch(OP_ROLLOUT, d, a, stream);
}

int snac_rollout_tiled(const snac_env_desc* d, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                       const int8_t* step_size, int32_t ring_ticks, int32_t first_tick, void* obs, float* reward, uint8_t* done,
                       const snac_rollout_record* rec, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (T < 0 || ring_ticks < 1 || first_tick < 0 || (long long)first_tick + T > ring_ticks) return fail(SNAC_ERR_ARG, "steps outside the ring");
    if (!obs) return fail(SNAC_ERR_ARG, "obs is null");
    if (T == 0) return SNAC_OK;
    KArgs a = make_args(d, st);
    a.T = T; a.t0 = t0; a.auto_reset = 1; a.obs_mode = SNAC_OBS_TILED;
    a.tiled_T = ring_ticks; a.tiled_t0 = first_tick;
    a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    if (rec) {
        a.actions_out = rec->actions; a.step_size_out = rec->step_size; a.plan_idx_out = rec->plan_idx; a.first_out = rec->first;
    }
    return launch(OP_ROLLOUT, d, a, stream);
}

int snac_rollout(const snac_env_desc* d, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                 const int8_t* step_size, int obs_mode, void* obs, float* reward, uint8_t* done, void* stream) {
    return snac_rollout_rec(d, st, T, t0, actions, step_size, obs_mode, obs, reward, done, nullptr, stream);
}

static int replay_gather(const snac_env_desc* d, const snac_state* st, int32_t cap, const void* obs_ring,
                         const uint8_t* first_ring, const int16_t* plan_idx_ring, const int32_t* tick_idx,
                         const int32_t* env_idx, int32_t batch, float* s_out, float* s_next_out, float* plan_out,
                         void* stream, int tiled) {
    if (int rc = check_common(d, st)) return rc;
    if (cap < 2 || batch < 0) return fail(SNAC_ERR_ARG, "cap must be >= 2 and batch >= 0");
    if (!obs_ring || !first_ring || !tick_idx || !env_idx || !s_out || !s_next_out) return fail(SNAC_ERR_ARG, "null pointer");
    if (plan_out && !plan_idx_ring) return fail(SNAC_ERR_ARG, "plan_out needs plan_idx_ring");
    if (plan_out && d->kind != SNAC_ENV_1D && ((uintptr_t)plan_out & 15)) return fail(SNAC_ERR_ARG, "plan_out must be 16-byte aligned");
    if (d->obs_tail & ~SNAC_TAIL_POSITION) return fail(SNAC_ERR_UNSUPPORTED, "replay gather supports the position tail only");
    if (batch == 0) return SNAC_OK;
    GArgs g;
    g.ld = base_obs_dim(d->kind) + tail_len(d->kind, d->obs_tail); g.frame_val = d->frame_value == 2 ? 2 : -1;
    g.n = d->num_envs; g.cap = cap; g.batch = batch; g.num_plans = d->num_plans; g.tiled = tiled;
    g.obs = obs_ring; g.first = first_ring; g.plan_idx = plan_idx_ring; g.tick = tick_idx; g.env = env_idx;
    g.plans = st->plans; g.s = s_out; g.s_next = s_next_out; g.plan_out = plan_out;
    hipStream_t s = (hipStream_t)stream;
    const bool f32 = d->obs_dtype == SNAC_OBS_F32;
    // (round 4: a variant that takes whole groups of 16 samples with 16-byte stores -- lane = piece of the group's consecutive rows --
    // was built and measured: 0.0382 against 0.0352 ms per 65 536 samples for this kernel, which already runs at 5.3 TB/s = 0.66 of the
    // peak; what rounds 2 and 3 reported as "0.23-0.30" was the Python wrapper's own index kernels.  Not kept; tools/gather_time.py)
    const dim3 grid((unsigned)((batch + 15) / 16)), block(256);   // 4 waves x 4 samples (S of k_gather; 8 were no faster)
    void (*kern)(const GArgs);
    if (d->kind == SNAC_ENV_1D) kern = f32 ? k_gather<1, float> : k_gather<1, double>;
    else if (d->kind == SNAC_ENV_2D) kern = f32 ? k_gather<2, float> : k_gather<2, double>;
    else kern = f32 ? k_gather<3, float> : k_gather<3, double>;
    hipLaunchKernelGGL(kern, grid, block, 0, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "gather launch");
    return SNAC_OK;
}

int snac_replay_gather(const snac_env_desc* d, const snac_state* st, int32_t cap, const void* obs_ring,
                       const uint8_t* first_ring, const int16_t* plan_idx_ring, const int32_t* tick_idx,
                       const int32_t* env_idx, int32_t batch, float* s_out, float* s_next_out, float* plan_out,
                       void* stream) {
    return replay_gather(d, st, cap, obs_ring, first_ring, plan_idx_ring, tick_idx, env_idx, batch, s_out, s_next_out, plan_out, stream, 0);
}

int snac_replay_gather_tiled(const snac_env_desc* d, const snac_state* st, int32_t cap, const void* obs_ring,
                             const uint8_t* first_ring, const int16_t* plan_idx_ring, const int32_t* tick_idx,
                             const int32_t* env_idx, int32_t batch, float* s_out, float* s_next_out, float* plan_out,
                             void* stream) {
    return replay_gather(d, st, cap, obs_ring, first_ring, plan_idx_ring, tick_idx, env_idx, batch, s_out, s_next_out, plan_out, stream, 1);
}

int snac_step(const snac_env_desc* d, const snac_state* st, uint32_t t, const int8_t* actions, const int8_t* step_size,
              int auto_reset, void* obs, float* reward, uint8_t* done, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    KArgs a = make_args(d, st);
    a.pool = d->num_envs; a.stats_on = 1;                        // the single-step kernel on the identity rows
    a.T = 1; a.t0 = t; a.auto_reset = auto_reset ? 1 : 0; a.obs_mode = obs ? SNAC_OBS_ALL : SNAC_OBS_NONE;
    a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    return launch(OP_TRANSITION, d, a, stream);
}

int snac_transition(const snac_env_desc* d, const snac_state* st, int32_t m, const int32_t* src_index, const int32_t* dst_index,
                    uint32_t t, const int8_t* actions, const int8_t* step_size, void* obs, float* reward, uint8_t* done,
                    void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (m < 0) return fail(SNAC_ERR_ARG, "m must be >= 0");
    // an absent index array means "row i": with either one absent, edge i touches pool row i, so m is bounded by the pool
    if ((!src_index || !dst_index) && m > d->num_envs) return fail(SNAC_ERR_ARG, "m exceeds the pool (num_envs)");
    if (m == 0) return SNAC_OK;
    KArgs a = make_args(d, st);
    a.pool = d->num_envs; a.n = m; a.src_index = src_index; a.dst_index = dst_index;
    a.T = 1; a.t0 = t; a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    return launch(OP_TRANSITION, d, a, stream);
}

int snac_import_state(const snac_env_desc* d, const snac_state* st, int32_t m, const int32_t* dst_index, const int32_t* position,
                      const int32_t* count_brick, const int32_t* count_step, const int32_t* plan_idx, const int32_t* total_brick,
                      const double* environment_memory, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (m < 0) return fail(SNAC_ERR_ARG, "m must be >= 0");
    if (!dst_index && m > d->num_envs) return fail(SNAC_ERR_ARG, "m exceeds the pool (num_envs)");
    if (!position || !count_brick || !count_step || !environment_memory) return fail(SNAC_ERR_ARG, "null pointer");
    if (m == 0) return SNAC_OK;
    IArgs g;
    g.m = m; g.pool = d->num_envs; g.num_plans = d->num_plans; g.dst_index = dst_index; g.pos = position; g.cb = count_brick;
    g.cs = count_step; g.plan_idx = plan_idx; g.tb = total_brick; g.mem = environment_memory; g.hdr = (int4*)st->hdr; g.episode = st->episode;
    g.grid = st->grid; g.plans = st->plans; g.plan_tb = st->plan_tb;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((m + 3) / 4)), block(256);
    if (d->kind == SNAC_ENV_1D) hipLaunchKernelGGL((k_import<1>), grid, block, 0, s, g);
    else if (d->kind == SNAC_ENV_2D) hipLaunchKernelGGL((k_import<2>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((k_import<3>), grid, block, 0, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "import launch");
    return SNAC_OK;
}

int snac_obs_equal(const snac_env_desc* d, const void* obs_a, const int32_t* idx_a, int32_t rows_a, const void* obs_b,
                   const int32_t* idx_b, int32_t rows_b, int32_t m, uint8_t* out, void* stream) {
    if (!d) return fail(SNAC_ERR_ARG, "null desc");
    if (d->kind < SNAC_ENV_1D || d->kind > SNAC_ENV_3D) return fail(SNAC_ERR_ARG, "unknown env kind");
    if (d->obs_dtype != SNAC_OBS_F64 && d->obs_dtype != SNAC_OBS_F32) return fail(SNAC_ERR_ARG, "unknown obs_dtype");
    if (m < 0 || rows_a <= 0 || rows_b <= 0) return fail(SNAC_ERR_ARG, "m must be >= 0 and rows_a / rows_b positive");
    if ((!idx_a && m > rows_a) || (!idx_b && m > rows_b)) return fail(SNAC_ERR_ARG, "m exceeds the number of rows");
    if (!obs_a || !obs_b || !out) return fail(SNAC_ERR_ARG, "null pointer");
    if (m == 0) return SNAC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = check_layout(d)) return rc;
    const int D = base_obs_dim(d->kind) + tail_len(d->kind, d->obs_tail);   // <= 459 values: the wave strides over the row
    const dim3 grid((unsigned)((m + 3) / 4)), block(256);
    if (d->obs_dtype == SNAC_OBS_F32)
        hipLaunchKernelGGL((k_equal<float>), grid, block, 0, s, (const float*)obs_a, idx_a, rows_a, (const float*)obs_b, idx_b, rows_b, m, D, out);
    else
        hipLaunchKernelGGL((k_equal<double>), grid, block, 0, s, (const double*)obs_a, idx_a, rows_a, (const double*)obs_b, idx_b, rows_b, m, D, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "equal launch");
    return SNAC_OK;
}

int snac_make_plans(const snac_env_desc* d, const snac_state* st, int32_t first, int32_t count, int32_t sparse, uint64_t seed,
                    int64_t plan_id_base, const int8_t* vertices, int32_t* area_out, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (first < 0 || count < 0 || (int64_t)first + count > d->num_plans) return fail(SNAC_ERR_ARG, "plan rows out of range");
    if (vertices && d->kind == SNAC_ENV_1D) return fail(SNAC_ERR_ARG, "vertices are a 2D / 3D input");
    if (count == 0) return SNAC_OK;
    PArgs g;
    g.first = first; g.count = count; g.sparse = sparse ? 1 : 0; g.use_vertices = vertices ? 1 : 0;
    g.key = stream_key(seed, 2); g.id_base = plan_id_base; g.vertices = vertices;
    g.plans = const_cast<void*>(st->plans); g.plan_tb = const_cast<int16_t*>(st->plan_tb); g.area_out = area_out;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((count + 3) / 4)), block(256);
    if (d->kind == SNAC_ENV_1D) hipLaunchKernelGGL((k_make_plans<1>), grid, block, 0, s, g);
    else if (d->kind == SNAC_ENV_2D) hipLaunchKernelGGL((k_make_plans<2>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((k_make_plans<3>), grid, block, 0, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "make_plans launch");
    return SNAC_OK;
}

int snac_observe(const snac_env_desc* d, const snac_state* st, void* obs, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (!obs) return fail(SNAC_ERR_ARG, "null obs");
    KArgs a = make_args(d, st);
    a.aux_op = AUX_OBSERVE; a.obs = obs;
    return launch(OP_AUX, d, a, stream);
}

int snac_iou(const snac_env_desc* d, const snac_state* st, double* out, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (!out) return fail(SNAC_ERR_ARG, "null out");
    KArgs a = make_args(d, st);
    a.aux_op = AUX_IOU; a.out_f64 = out;
    return launch(OP_AUX, d, a, stream);
}

int snac_export_grid(const snac_env_desc* d, const snac_state* st, double* out, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (!out) return fail(SNAC_ERR_ARG, "null out");
    KArgs a = make_args(d, st);
    a.out_f64 = out;
    const long long total = (long long)d->num_envs * (d->kind == SNAC_ENV_1D ? 34 : 676);
    const int block = 256;
    long long want = (total + block - 1) / block;
    const unsigned grid = (unsigned)(want > 8192 ? 8192 : want);
    hipStream_t s = (hipStream_t)stream;
    if (d->kind == SNAC_ENV_1D) hipLaunchKernelGGL((k_export<1>), dim3(grid), dim3(block), 0, s, a, total);
    else if (d->kind == SNAC_ENV_2D) hipLaunchKernelGGL((k_export<2>), dim3(grid), dim3(block), 0, s, a, total);
    else hipLaunchKernelGGL((k_export<3>), dim3(grid), dim3(block), 0, s, a, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "export launch");
    return SNAC_OK;
}


}  // extern "C"
