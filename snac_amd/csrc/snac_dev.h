// snac_dev.h -- what every translation unit of libsnac_hip.so shares (internal, not installed): the kernel argument block, the counter
// RNG, the packed env header, the three env kinds with their LDS images and transition rules (K1D / K2D / K3D), the observation-row
// writers (write_obs for the tile kernels; emit_tile / emit_rows_var: rows transposed through an LDS staging tile, 16 bytes per lane),
// and the declarations of the per-family launch functions that the dispatch in snac_hip.hip calls.
//
// Integer / indexing work only -- no MFMA; the bound is HBM (observation rows written) or, for small batches, the chain of ticks.
// DESIGN.md section 3 has the table of kernels with their measured times.  Translation units (one object each, built in parallel):
//   k_roll2d.hip    k_rollout2d   the headline: 2D rollouts on tiles of 64 envs, lane = env in the transition AND in the observation rows;
//                                 plan rows per wave in LDS, refilled through the scalar cache; no vector load in the loop
//   k_roll2dt.hip   k_rollout2dt  2D rollouts of small and middle batches, time-parallel: one wave per env, lane = tick
//   k_roll1dt.hip   k_rollout1dt  the same idea for 1D
//   k_roll1dl.hip   k_rollout1dl  1D rollouts of large batches: lane = env, the headline kernel's shape (rows1d.h: the 1D rows of a wave as one run)
//   k_roll3db.hip   k_rollout3db  3D rollouts: one stepper wave (lane = env) and eight writer waves per 64 envs, one barrier per tick
//   k_roll3d.hip    k_rollout3d   3D rollouts of small / odd batches: 8 envs per wave, software-pipelined round the store stream
//   k_step.hip      k_step2d / 3d snac_step on identity rows: wide loads, rows through emit_tile; AUX forms: masked resets and observe without a step
//   k_step3dq.hip   k_step3dq     the canonical 3D snac_step: 16 envs per wave, four lanes per env
//   k_step1d.hip    k_step1d, k_edges1d   the 1D snac_step (canonical rows and the layout variants) and 1D tree edges
//   k_reset.hip     k_reset, k_iou        whole-batch resets without reading the old state; snac_iou lane-per-env
//   k_nodes2d.hip   k_edges2dp    2D tree edges on node pools of one 128-byte record per node
//   k_mailbox.hip   k_mailbox     the resident stepper behind the drop-in classes (mailbox_host.h: the host half of its protocol)
//   k_trans.hip     k_transition2d / 3d, k_edges3d: single steps and tree edges with gathered rows
//   k_tile{1,2,3}d.hip  the tile kernels k_rollout / k_transition / k_aux (rounds 1-2) behind all of them (templates: k_tile.inc)
//   k_misc.hip      export / import / equality / plan generators / replay gather, with their entry points
//   snac_hip.hip    the C ABI of include/snac_hip.h and the dispatch (launch(): one table of thresholds)
//   snac_traj.hip   trajectory memory
// The env records are read from HBM once per launch, kept on chip for all T steps, written back once.  Rollout outputs are [T][N][D]
// or, with SNAC_OBS_TILED, tile-major [N / 64][T][64][D].
#pragma once
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

#include "snac_tune.h"   // the ids of the dispatch table's entries (kept out of this header: adding a knob does not touch the kernels' source)

namespace snac_detail {
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

enum Op { OP_ROLLOUT, OP_AUX, OP_TRANSITION };

// which kernel the calling thread's last launch went to (snac_last_kernel())
extern thread_local const char* g_kernel;

// ---- host side, defined in snac_hip.hip
int check_layout(const snac_env_desc* d);
int base_obs_dim(int kind);
int tail_len(int kind, int tail);
int check_common(const snac_env_desc* d, const snac_state* st);
KArgs make_args(const snac_env_desc* d, const snac_state* st);
int launch(Op op, const snac_env_desc* d, const KArgs& a, void* stream);

// ---- the launch functions of the kernel families (each defined beside its kernels; the dispatch decides, they only launch)
void launch_tile1d(Op op, bool dyn, int E, int obs_dtype, const KArgs& a, hipStream_t s);      // k_tile1d.hip
void launch_tile2d(Op op, bool dyn, int E, int obs_dtype, const KArgs& a, hipStream_t s);      // k_tile2d.hip
void launch_tile3d(Op op, bool dyn, int E, int obs_dtype, const KArgs& a, hipStream_t s);      // k_tile3d.hip
void launch_roll2d(const snac_env_desc* d, const KArgs& a, hipStream_t s);                     // k_roll2d.hip
void launch_roll2dt(const snac_env_desc* d, const KArgs& a, hipStream_t s);                    // k_roll2dt.hip
void launch_roll1dt(const snac_env_desc* d, const KArgs& a, hipStream_t s);                    // k_roll1dt.hip
void launch_roll1dl(const snac_env_desc* d, const KArgs& a, hipStream_t s);                    // k_roll1dl.hip
void launch_roll3d(const snac_env_desc* d, const KArgs& a, hipStream_t s);                     // k_roll3d.hip
void launch_roll3db(const snac_env_desc* d, const KArgs& a, hipStream_t s);                    // k_roll3db.hip
void launch_step2d(const snac_env_desc* d, const KArgs& a, bool half, hipStream_t s);          // k_step.hip
void launch_step3d(const snac_env_desc* d, const KArgs& a, bool span, hipStream_t s);          // k_step.hip (span: k_step3ds)
void launch_trans2d(const snac_env_desc* d, const KArgs& a, hipStream_t s);                    // k_trans.hip
void launch_trans3d(const snac_env_desc* d, const KArgs& a, hipStream_t s);                    // k_trans.hip (k_edges3d for gathered rows)
}  // namespace snac_detail
using snac_detail::fail;
using snac_detail::fail_hip;
using snac_detail::g_err;
using snac_detail::KArgs;
using snac_detail::Op;
using snac_detail::OP_ROLLOUT;
using snac_detail::OP_AUX;
using snac_detail::OP_TRANSITION;

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

// ------------------------------------------------------------------------------------------------
// The RULES of the 2D step (Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:85-147, static: DMP_Env_2D_static.py:95-154), ONCE, for every
// kernel that steps a 2D env lane-per-env (K2D::step of the tile kernels and the mailbox, k_rollout2d, k_rollout2db, k_step2d,
// k_transition2d, k_edges2d, k_edges2dp): the kernels differ in where a board lives (two-bit cells in LDS, row words in LDS or
// registers, node records) and hand in the two facts the rules need about the cell under the agent -- `was`: it holds a brick,
// `planned`: the plan wants one; what a step does to the counters, the position, the reward and `done` is decided here and nowhere
// else (round 5 had seven copies).  When `drop` comes back the caller sets the cell's bit (+= 1, then clamped to 1: :115, :134-135).
// k_rollout2dt is time-parallel (counters by ballots, positions by scans of composed clamps): the same rules in another formulation.
// The pieces of the rules that the TIME-PARALLEL kernels (k_rollout1dt, k_rollout2dt: a lane per tick, counters by ballots, positions by
// scans of composed clamps) share with the lane-per-env formulation below: when a drop ends the episode, and what a drop earns.
__device__ __forceinline__ bool term_rule(bool drop, int cb, int tb, int brick_gt) { return drop && cb >= tb + brick_gt; }   // 1D :107-114, 2D :117-126 (brick_gt: SNAC_RULE_BRICK_GT)
__device__ __forceinline__ int reward1d(bool drop, bool term, int hnew, int pl) {                                           // DMP_Env_1D_static.py:117-123
    return (drop && !term) ? (hnew > pl ? -1 : (hnew == pl ? 10 : 1)) : 0;
}
__device__ __forceinline__ int reward2d(bool drop, bool term, bool was, bool planned) {   // the un-clamped cell against the plan (:129-133): 5 iff it was empty and is planned
    return (drop && !term && !was && planned) ? 5 : 0;
}
// The RULES of the 1D step (Env/1D/DMP_Env_1D_static.py:85-136, dynamic: DMP_Env_1D_dynamic_usedata_plan.py), ONCE, for the lane-per-env
// kernels (K1D::step of the tile kernels and the mailbox, k_rollout1dl): the caller hands in the height under the agent (`hold`) and the
// plan's height there (`pl`); when `drop` comes back it stores `hnew` in that cell.  s.cs, s.cb and s.r are updated here.
struct Rule1D { bool drop, done; int hnew, reward; };
__device__ __forceinline__ Rule1D rules1d(Lane& s, int act, int k, int hold, int pl, int ts_done, int brick_gt) {
    Rule1D o;
    o.hnew = min(hold + 1, CNT_MAX);
    o.drop = act == 2;
    s.cs = min(s.cs + 1, CNT_MAX);
    if (o.drop) s.cb = min(s.cb + 1, CNT_MAX);
    if (act == 0) s.r = max(s.r - k, 2);                             // clip_position :57-64
    if (act == 1) s.r = min(s.r + k, 31);
    const bool term = term_rule(o.drop, s.cb, s.tb, brick_gt);       // :107-114, before the time limit
    o.done = term || s.cs >= ts_done;
    o.reward = reward1d(o.drop, term, o.hnew, pl);
    return o;
}

struct Rule2D { bool drop, term, done; int reward; };
__device__ __forceinline__ Rule2D rules2d(Lane& s, int act, int k, bool was, bool planned, int ts_done, int brick_gt) {
    Rule2D o;
    o.drop = act == 4;
    s.cs = min(s.cs + 1, CNT_MAX);
    if (o.drop) s.cb = min(s.cb + 1, CNT_MAX);
    if (act == 0) s.c = max(s.c - k, 3);                             // clip_position :74-83
    if (act == 1) s.c = min(s.c + k, 22);
    if (act == 2) s.r = min(s.r + k, 22);                            // "up" is row + k (:100-103)
    if (act == 3) s.r = max(s.r - k, 3);
    o.term = term_rule(o.drop, s.cb, s.tb, brick_gt);                // :117-126, tested before the time limit
    o.done = o.term || s.cs >= ts_done;
    o.reward = reward2d(o.drop, o.term, was, planned);
    return o;
}

// The RULES of the 3D step (Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py:142-231, static: DMP_simulator_3d_static_circle.py:153-230),
// ONCE, for K3D::step (tile kernels, mailbox), k_rollout3d, k_rollout3db, k_step3d / 3ds / 3dq, k_transition3d and k_edges3d: a kernel
// hands in the six cells the rules look at -- n0 .. n3: the four neighbours as check_sur sees them (left, right, "up" = row + 1, "down";
// -1 frame, > 0 built), c2 / c3: the cells two and three ahead in the action's direction -- and gets back what the step does:
//   built        a brick goes on the neighbour in direction act & 3 (the caller writes newh there and adds to s.cross / count arrays)
//   sel          the reward is reward_check3d(newh, plan cell) (:233-240); otherwise it is reward0 (-100 boxed in, else 0)
//   done         SURVEY 8a-Q7 .. Q9: static tests the neighbours BEFORE the build, dynamic AFTER it; a successful non-terminal build
//                does not test the time limit
// s.cs, s.r, s.c, s.cb are updated here.  The plan is not needed (k_rollout3d defers the plan-dependent part by a tick).
struct Rule3D { bool built, sel, done; int newh, reward0; };
__device__ __forceinline__ int reward_check3d(int newh, int pl) { return newh > pl ? -1 : (newh == pl ? 10 : 1); }
template <bool DYN>
__device__ __forceinline__ Rule3D rules3d(Lane& s, int act, int k, int n0, int n1, int n2, int n3, int c2, int c3, bool active, int ts_done, int brick_gt) {
    Rule3D o;
    const int d = act & 3;
    const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
    const int nd = d == 0 ? n0 : (d == 1 ? n1 : (d == 2 ? n2 : n3));
    const bool valid = (unsigned)act < 8u, is_build = valid && act >= 4;
    const bool boxed_pre = n0 != 0 && n1 != 0 && n2 != 0 && n3 != 0;
    s.cs = min(s.cs + 1, CNT_MAX);
    const bool can_move = valid && act < 4 && nd == 0;               // check[act] == 0
    const int m = (k >= 2 && c2 == 0) ? ((k >= 3 && c3 == 0) ? 3 : 2) : 1;   // move_step (:93-123): consecutive free cells, at most k
    s.r += can_move ? dr * m : 0;
    s.c += can_move ? dc * m : 0;
    o.built = active && is_build && nd != -1;                        // check[act] == 0 for act in 4 .. 7: only the frame refuses a brick
    o.newh = min(nd + 1, CNT_MAX);
    s.cb = o.built ? min(s.cb + 1, CNT_MAX) : s.cb;
    const bool limit = s.cb >= s.tb + brick_gt;
    const bool bottom = (s.cs >= ts_done) || (!DYN && boxed_pre);    // moves, blocked moves, blocked builds (static :226, dynamic :226)
    bool fin;
    if (DYN) {
        // the neighbours re-evaluated AFTER the build (:199-206): the built cell now blocks its direction
        const bool boxed_post = o.built ? ((d == 0 || n0 != 0) && (d == 1 || n1 != 0) && (d == 2 || n2 != 0) && (d == 3 || n3 != 0)) : boxed_pre;
        fin = is_build && (boxed_post || limit);                     // -100 (:199-206), then the brick limit with reward 0 (:207-213)
        o.reward0 = (is_build && boxed_post) ? -100 : 0;
    } else {
        fin = is_build && (limit || boxed_pre);                      // :210-215, reward 0
        o.reward0 = 0;
    }
    o.sel = is_build && !fin && o.built;
    o.done = fin ? true : (o.sel ? false : bottom);
    return o;
}

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
        const Rule2D u = rules2d(s, act, k, was, planned, ts, bg);
        if (u.drop) *cw = w | (1ull << off);
        done = u.done;
        reward = u.reward;
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
        const int d = act & 3;
        const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
        const int dl = dr * 26 + dc;
        // check_sur (:88-102 / :77-91): left, right, "up" (row + 1), "down" (row - 1); the cells two and three ahead (within the frame)
        const int n0 = h[-1], n1 = h[1], n2 = h[26], n3 = h[-26];
        const int c2 = h[2 * dl], c3 = h[3 * dl];
        const int tcell = (s.r + dr - 3) * 20 + (s.c + dc - 3);      // the build target in plan coordinates (a build does not move)
        const Rule3D u = rules3d<DYN>(s, act, k, n0, n1, n2, n3, c2, c3, true, ts, bg);   // the rules, once for every 3D kernel
        reward = u.reward0;
        done = u.done;
        if (u.built) {                                               // (the frame refuses a brick: the target lies inside)
            h[dl] = (int16_t)u.newh;
            const int pl = ((const int16_t*)a.plans)[(size_t)s.pidx * GE + tcell];
            s.cross += u.newh <= pl ? 1 : 0;                         // running sum of min(height, plan) for iou()
            if (u.sel) reward = reward_check3d(u.newh, pl);          // reward_check (:232-239)
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
        const Rule1D u = rules1d(s, act, k, (int)*h, (int)plan(lds)[lane * ES + s.r - 2], ts, bg);   // the rules: above
        if (u.drop) *h = (int16_t)u.hnew;
        reward = u.reward; done = u.done;
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

// ------------------------------------------------------------------------------------------------
// Lane-per-env observation rows (round 3): lane l holds the 51 values of env l of a 64-env tile -- cell(el) for the 49 window
// cells, v0 / v1 the two scalar slots.  They are transposed through a staging tile in LDS ([env][51] values, odd dword stride:
// conflict-free) into the tile's contiguous piece of the output (64 x 51 values: 13 056 B of float32, 26 112 B of float64 in two
// halves of 32 envs), read back 16 bytes per lane and stored with global_store_dwordx4: 1 KiB per store instruction, 13 / 26 per
// tile instead of 64 row stores.  stg: TILE_STG_BYTES of 16-byte aligned LDS of this wave (every LDS read of a half is issued
// before its first store; reads of lanes past the tile's end fall into the pad).  g: the tile's first output byte, 16-byte aligned;
// nenv < 64: a ragged tile (nenv * 51 * sizeof(OT) must be a multiple of 16: the callers require N % 4 == 0).
constexpr int TILE_STG_BYTES = 13 * 1024;

// a load that is, or is not, non-temporal: state that a step kernel reads once per tick should not stay in the caches when the batch is
// large (k_step3dq at 524 288 envs: 70 us with non-temporal span loads, 101 with plain ones) and SHOULD when the whole state fits
// them (65 536 envs: 12.8 us non-temporal, 11.5 plain) -- the launch picks by batch size (SNAC_STEP3D_NTLOAD_MIN, SNAC_STEP2D_PLAIN_LO / _HI; profiles/r06_step_loads.txt)
template <bool NT, typename V>
__device__ __forceinline__ V load_nt_if(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// a 16-byte store; NT: non-temporal (streamed rows that nobody reads back soon: they do not displace what a kernel's gathered reads find
// in L2 / the Infinity Cache -- k_edges2dp's node records)
template <bool NT>
__device__ __forceinline__ void store16(char* p, const uint4& v) {
    if constexpr (NT) {
        typedef uint32_t u32x4_st __attribute__((ext_vector_type(4)));
        u32x4_st t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, (u32x4_st*)p);
    } else {
        *(uint4*)p = v;
    }
}

// Which kernels' rows leave non-temporal is a BUILD constant (a run-time flag does not survive the compiler: two stores that differ in
// the hint alone are merged into one plain store): bits 1 = snac_step (k_step2d / k_step3d / k_step3dq), 2 = snac_transition with gathered
// rows (k_edges2d / k_edges3d); k_edges2dp has both forms (SNAC_NODES2D_NT).  A/B builds: make CXXFLAGS+=-DSNAC_ROWS_NT=3 OBJDIR=.obj_nt ...
// Measured (profiles/r06_edges.txt, 524 288 edges / envs): edges on batch rows 80.4 -> 68.1 us (k_edges2d), 229.9 -> 223.0 (k_edges3d) -- the
// gathered records stay in the Infinity Cache; snac_step 2D 46.2 -> 53.9 us, 3D 69.7 -> 78.4 per tick (worse: its reads are a stream of their
// own, 65 536 envs: 8.5 -> 8.1) -- so: edges yes, steps no.
#ifndef SNAC_ROWS_NT
#define SNAC_ROWS_NT 2
#endif
constexpr bool ROWS_NT_STEP = (SNAC_ROWS_NT & 1) != 0, ROWS_NT_EDGES = (SNAC_ROWS_NT & 2) != 0;

template <typename OT, bool NT = false, class F>
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
        // The lanes exchange their values through LDS without a barrier (one wave: LDS operations complete in order) -- which the
        // COMPILER does not know: a lane that wrote nothing in this half may be handed the values it read in the half before.  The build
        // with non-temporal stores did exactly that (round 6: the second half's ds_read_b128 sunk into the writers' branch, rows 32-63 of
        // every tile stale; tests/test_gpu_nodes2d.py caught it -- the plain form happened not to).  What was tried: an empty asm with a
        // memory clobber (a vmcnt(0) per tile: 2.3 -> 5.1 ms for the headline pass); fences of wavefront scope (no instruction, but they
        // end before the machine passes: the reads sink into the ragged path's store blocks, one register quad, a vmcnt(0) per store).
        // VOLATILE reads are reads the compiler must perform where they stand, and cost nothing (headline 2.300-2.309 against
        // 2.311-2.313 ms, k_step2d 45.6-46.1 against 45.3-45.4 us: level).  The LDS address space is spelled out: a volatile access
        // through the generic pointer becomes a flat_load and waits on vmcnt.
        uint4 fv[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            typedef uint32_t u32x4_ld __attribute__((ext_vector_type(4)));
            const u32x4_ld t = *(const volatile __attribute__((address_space(3))) u32x4_ld*)(stg + i * 1024 + lane * 16);
            fv[i] = make_uint4(t.x, t.y, t.z, t.w);
        }
        char* const gh = g + (size_t)h * STG_BYTES + lane * 16;
        if (full) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if ((i + 1) * 1024 <= STG_BYTES || i * 1024 + lane * 16 < STG_BYTES) store16<NT>(gh + i * 1024, fv[i]);
        } else {
            const int valid = min(max(nenv - h * HE, 0), HE) * D * (int)sizeof(OT);
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (i * 1024 + lane * 16 < valid) store16<NT>(gh + i * 1024, fv[i]);
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

typedef const uint32_t __attribute__((address_space(4))) cmem_u32;   // constant address space: uniform addresses become s_load

constexpr int TB_MAX = 2048;   // plan_tb rows staged in LDS per block

// (m & a) | (~m & b) as the one instruction it is (left to itself the compiler hoists ~m out of a loop and issues two)
__device__ __forceinline__ uint32_t bfi32(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}

template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ int dpp_from(int idv, int v) { return __builtin_amdgcn_update_dpp(idv, v, CTRL, ROWS, 0xf, false); }

}  // namespace
