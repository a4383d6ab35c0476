// snac_hip.hip -- gfx950 (MI355X) kernels and the C ABI of include/snac_hip.h.
//
// One wavefront (64 lanes) owns one env.  The env record is read once (16-byte header by scalar load,
// grid row(s) with unit-stride lanes), kept on chip for all T steps of a launch, and written back once:
//   1D  30 heights, one per lane, in a VGPR; the 5-cell window is a cross-lane read
//   2D  20x20 occupancy bit-board, one 20-bit row per lane in a VGPR; the 7x7 window is one
//       ds_bpermute (LDS crossbar) + shift/mask per lane
//   3D  20x20 height map and its plan staged in LDS (2 x 800 B per wave); the 7x7 window is one
//       ds_read_u16 per lane and stays in a VGPR as the collision neighbourhood of the next step
// Lanes 0..W-1 (W = 5 or 49) each produce one window cell of the observation, lanes W and W+1 the two
// scalar slots, so the observation row of an env is one contiguous store of obs_dim elements.
// Integer / indexing work only -- no MFMA; the bound is HBM (obs writes).  See DESIGN.md.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "snac_hip.h"

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char* msg) {
    std::snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int fail_hip(hipError_t e, const char* where) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return SNAC_ERR_HIP;
}

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
    uint32_t elo = (uint32_t)env, ehi = (uint32_t)(env >> 32);
    EnvKeys k;
    k.e0 = mix32(key ^ mix32(elo + 0x85EBCA6Bu * ehi + 0x1B873593u));
    k.e1 = mix32((key + 0x27D4EB2Fu) ^ mix32((elo ^ 0x165667B1u) + 0xC2B2AE35u * ehi));
    return k;
}
__device__ inline uint32_t rng_word(EnvKeys k, uint32_t t) { return mix32(mix32(k.e0 ^ (0x9E3779B9u * t)) + k.e1); }

// ------------------------------------------------------------------------------------------------
struct KArgs {
    int32_t n, num_plans, static_plan, T, auto_reset, obs_mode;
    uint32_t t0, key_step, key_plan;
    int64_t env_id_base;
    snac_env_hdr* hdr;
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
    // reset kernel only
    const uint8_t* mask;
    const int16_t* plan_idx_in;
    int32_t observe_only;
    double* out_f64;
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr double FX40 = 1099511627776.0;  // 2^40

// Scalar part of an env, shared by the three kinds.  Everything here is wave-uniform.
struct Scalars {
    int r, c, cb, cs, tb, pidx, ep_ret, flags;
    __device__ void from(const snac_env_hdr& h) {
        r = h.pos_r; c = h.pos_c; flags = h.flags; cb = h.count_brick; cs = h.count_step; tb = h.total_brick;
        pidx = h.plan_idx; ep_ret = h.ep_return;
    }
    __device__ snac_env_hdr to() const {
        snac_env_hdr h;
        h.pos_r = (int8_t)r; h.pos_c = (int8_t)c; h.flags = (uint8_t)flags; h.reserved = 0;
        h.count_brick = (int16_t)cb; h.count_step = (int16_t)cs; h.total_brick = (int16_t)tb; h.plan_idx = (int16_t)pidx;
        h.ep_return = ep_ret;
        return h;
    }
};

// ------------------------------------------------------------------------------------------------
// 1D: Env/1D/DMP_Env_1D_static.py, Env/1D/DMP_Env_1D_dynamic_usedata_plan.py
template <bool DYN_>
struct Env1D {
    static constexpr bool DYN = DYN_;
    static constexpr int D = 7, W = 5, A = 3, TS = 750, GE = 32, LDS_BYTES = 0;
    Scalars s;
    int g, p;  // lane l < 30: height / plan of interior cell l (bordered index l + 2)

    __device__ void bind(char*, int) {}
    __device__ void load(const KArgs& a, int env, int lane) {
        s.from(a.hdr[env]);
        g = lane < GE ? ((const int16_t*)a.grid)[(size_t)env * GE + lane] : 0;
        load_plan(a, lane);
    }
    __device__ void load_plan(const KArgs& a, int lane) {
        p = lane < GE ? ((const int16_t*)a.plans)[(size_t)s.pidx * GE + lane] : 0;
    }
    __device__ void store(const KArgs& a, int env, int lane) const {
        if (lane == 0) a.hdr[env] = s.to();
        if (lane < GE) ((int16_t*)a.grid)[(size_t)env * GE + lane] = (int16_t)g;
    }
    // reset: DMP_Env_1D_static.py:66-83, DMP_Env_1D_dynamic_usedata_plan.py:40-70
    __device__ void reset(const KArgs& a, int pidx, int lane) {
        s.pidx = pidx; s.tb = a.plan_tb[pidx];
        s.r = 2; s.c = 0; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.flags = 0;
        g = 0;
        load_plan(a, lane);
    }
    // step: DMP_Env_1D_static.py:85-136
    __device__ void step(int act, int k, int lane, int& reward, bool& done) {
        s.cs += 1;
        reward = 0;
        done = s.cs >= TS;
        if (act == 0) s.r = max(s.r - k, 2);            // clip_position :57-64
        else if (act == 1) s.r = min(s.r + k, 31);
        else if (act == 2) {
            const int cell = s.r - 2;
            s.cb += 1;
            const int h = rdlane(g, cell) + 1;
            const int pl = rdlane(p, cell);
            if (lane == cell) g = h;
            if (s.cb >= s.tb) { reward = 0; done = true; }      // :107-114, before the time limit
            else reward = h > pl ? -1 : (h == pl ? 10 : 1);     // :117-123
        }
    }
    __device__ double obs_value(int lane) const {
        const int cell = s.r - 2 + lane - 2;                    // interior index of window cell `lane`
        const int v = __shfl(g, cell, 64);
        const bool inside = (unsigned)cell < 30u;
        const double num = lane == W ? (double)s.cb : (double)s.cs;
        const double den = lane == W ? (double)s.tb : (double)TS;
        const double sc = DYN ? num / den : num;
        return lane < W ? (double)(inside ? v : -1) : sc;
    }
    // iou: DMP_Env_1D_static.py:138-151
    __device__ double iou(int lane) const {
        const int a1 = wave_sum(p), a2 = wave_sum(g), k = wave_sum(max(g - p, 0));
        const int cross = a2 - k;
        return (double)cross / (double)(a1 + a2 - cross);
    }
};

// ------------------------------------------------------------------------------------------------
// 2D: Env/2D/DMP_Env_2D_static.py, Env/2D/DMP_Env_2D_dynamic_usedata_plan.py
template <bool DYN_>
struct Env2D {
    static constexpr bool DYN = DYN_;
    static constexpr int D = 51, W = 49, A = 5, TS = 600, GE = 20, LDS_BYTES = 0;
    Scalars s;
    uint32_t g, p;  // lane l < 20: occupancy / plan bits of interior row l
    int wi, wj;     // window coordinates of this lane

    __device__ void bind(char*, int lane) { wi = lane / 7; wj = lane - 7 * wi; }
    __device__ void load(const KArgs& a, int env, int lane) {
        s.from(a.hdr[env]);
        g = lane < GE ? ((const uint32_t*)a.grid)[(size_t)env * GE + lane] : 0u;
        load_plan(a, lane);
    }
    __device__ void load_plan(const KArgs& a, int lane) {
        p = lane < GE ? ((const uint32_t*)a.plans)[(size_t)s.pidx * GE + lane] : 0u;
    }
    __device__ void store(const KArgs& a, int env, int lane) const {
        if (lane == 0) a.hdr[env] = s.to();
        if (lane < GE) ((uint32_t*)a.grid)[(size_t)env * GE + lane] = g;
    }
    // reset: DMP_Env_2D_dynamic_usedata_plan.py:34-66 (total_brick floor of 30 is folded into plan_tb)
    __device__ void reset(const KArgs& a, int pidx, int lane) {
        s.pidx = pidx; s.tb = a.plan_tb[pidx];
        s.r = 3; s.c = 3; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.flags = 0;
        g = 0u;
        load_plan(a, lane);
    }
    // step: DMP_Env_2D_dynamic_usedata_plan.py:85-147
    __device__ void step(int act, int k, int lane, int& reward, bool& done) {
        s.cs += 1;
        reward = 0;
        done = s.cs >= TS;
        if (act == 0) s.c = max(s.c - k, 3);                    // clip_position :74-83
        else if (act == 1) s.c = min(s.c + k, 22);
        else if (act == 2) s.r = min(s.r + k, 22);              // "up" is row + k (:100-103)
        else if (act == 3) s.r = max(s.r - k, 3);
        else if (act == 4) {
            const int row = s.r - 3;
            const uint32_t bit = 1u << (s.c - 3);
            s.cb += 1;
            const bool was = (rdlane((int)g, row) & bit) != 0u;
            const bool planned = (rdlane((int)p, row) & bit) != 0u;
            if (lane == row) g |= bit;                          // += 1 then clamp to 1 (:115, :134-135)
            if (s.cb >= s.tb) { reward = 0; done = true; }      // :117-126, before the time limit
            else reward = (!was && planned) ? 5 : 0;            // un-clamped cell vs plan (:129-133)
        }
    }
    __device__ double obs_value(int lane) const {
        const int row = s.r - 6 + wi, col = s.c - 6 + wj;       // interior coordinates of the window cell
        const uint32_t bits = (uint32_t)__shfl((int)g, row, 64);
        const bool inside = (unsigned)row < 20u && (unsigned)col < 20u;
        const int v = inside ? (int)((bits >> (col & 31)) & 1u) : -1;
        const double num = lane == W ? (double)s.cb : (double)s.cs;
        const double den = lane == W ? (double)s.tb : (double)TS;
        const double sc = DYN ? num / den : num;
        return lane < W ? (double)v : sc;
    }
    // boolean IoU: script/DQN/2d/DQN_2d_dynamic.py:63-71
    __device__ double iou(int lane) const {
        const int inter = wave_sum(__popc(g & p)), uni = wave_sum(__popc(g | p));
        return (double)inter / (double)uni;
    }
};

// ------------------------------------------------------------------------------------------------
// 3D: Env/3D/DMP_simulator_3d_static_circle.py, Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py
template <bool DYN_>
struct Env3D {
    static constexpr bool DYN = DYN_;
    static constexpr int D = 51, W = 49, A = 8, TS = DYN_ ? 1000 : 1300, GE = 400, LDS_BYTES = 2 * GE * 2;
    Scalars s;
    int16_t* lg;    // LDS: this wave's 20x20 height map
    int16_t* lp;    // LDS: this wave's plan
    int win;        // lane < 49: current window cell (frame = -1)
    int wi, wj;

    __device__ void bind(char* lds, int lane) { lg = (int16_t*)lds; lp = lg + GE; wi = lane / 7; wj = lane - 7 * wi; }
    __device__ int window_cell() const {
        const int row = s.r - 6 + wi, col = s.c - 6 + wj;
        const bool inside = (unsigned)row < 20u && (unsigned)col < 20u;
        return inside ? (int)lg[row * 20 + col] : -1;
    }
    __device__ void load(const KArgs& a, int env, int lane) {
        s.from(a.hdr[env]);
        const uint32_t* src = (const uint32_t*)((const int16_t*)a.grid + (size_t)env * GE);
        for (int i = lane; i < GE / 2; i += 64) ((uint32_t*)lg)[i] = src[i];
        load_plan(a, lane);
        win = lane < W ? window_cell() : 0;
    }
    __device__ void load_plan(const KArgs& a, int lane) {
        const uint32_t* src = (const uint32_t*)((const int16_t*)a.plans + (size_t)s.pidx * GE);
        for (int i = lane; i < GE / 2; i += 64) ((uint32_t*)lp)[i] = src[i];
    }
    __device__ void store(const KArgs& a, int env, int lane) const {
        if (lane == 0) a.hdr[env] = s.to();
        uint32_t* dst = (uint32_t*)((int16_t*)a.grid + (size_t)env * GE);
        for (int i = lane; i < GE / 2; i += 64) dst[i] = ((const uint32_t*)lg)[i];
    }
    // reset: DMP_simulator_3d_dynamic_triangle_usedata.py:45-75
    __device__ void reset(const KArgs& a, int pidx, int lane) {
        s.pidx = pidx; s.tb = a.plan_tb[pidx];
        s.r = 3; s.c = 3; s.cb = 0; s.cs = 0; s.ep_ret = 0; s.flags = 0;
        for (int i = lane; i < GE / 2; i += 64) ((uint32_t*)lg)[i] = 0u;
        load_plan(a, lane);
        win = lane < W ? window_cell() : 0;
    }
    // step: DMP_simulator_3d_static_circle.py:153-230, DMP_simulator_3d_dynamic_triangle_usedata.py:142-231
    __device__ void step(int act, int k, int lane, int& reward, bool& done) {
        s.cs += 1;
        reward = 0;
        // check_sur (:88-102 / :77-91) from the window of the previous observation: lanes (3,2) (3,4) (4,3) (2,3)
        int nb[4] = { rdlane(win, 23), rdlane(win, 25), rdlane(win, 31), rdlane(win, 17) };
        const bool boxed_pre = nb[0] != 0 && nb[1] != 0 && nb[2] != 0 && nb[3] != 0;
        done = (s.cs >= TS) || (!DYN && boxed_pre);             // static :226, dynamic :226
        if ((unsigned)act > 7u) return;
        const int d = act & 3;
        const int dl = d == 0 ? -1 : (d == 1 ? 1 : (d == 2 ? 7 : -7));   // window-lane step of the direction
        const int dr = d == 2 ? 1 : (d == 3 ? -1 : 0), dc = d == 0 ? -1 : (d == 1 ? 1 : 0);
        const int nd = d == 0 ? nb[0] : (d == 1 ? nb[1] : (d == 2 ? nb[2] : nb[3]));
        if (act < 4) {
            if (nd == 0) {                                      // check[act] == 0
                // move_step (:104-134): consecutive free cells, at most k
                const int c2 = rdlane(win, 24 + 2 * dl), c3 = rdlane(win, 24 + 3 * dl);
                int m = 1;
                if (k >= 2 && c2 == 0) { m = 2; if (k >= 3 && c3 == 0) m = 3; }
                s.r += dr * m; s.c += dc * m;                   // clip_position is a no-op: walls stop the move
                win = lane < W ? window_cell() : 0;
            }
            return;
        }
        bool built = false;
        int newh = 0, tcell = 0;
        if (nd != -1) {                                         // check[act] == 0 for act in 4..7
            built = true;
            s.cb += 1;
            newh = nd + 1;
            tcell = (s.r - 3 + dr) * 20 + (s.c - 3 + dc);
            if (lane == 0) lg[tcell] = (int16_t)newh;
            if (lane == 24 + dl) win = newh;
        }
        if (DYN) {
            // neighbours re-evaluated AFTER the build (:199-206)
            const bool boxed_post = built ? ((d == 0 || nb[0] != 0) && (d == 1 || nb[1] != 0) && (d == 2 || nb[2] != 0) &&
                                             (d == 3 || nb[3] != 0))
                                          : boxed_pre;
            if (boxed_post) { reward = -100; done = true; return; }
            if (s.cb >= s.tb) { reward = 0; done = true; return; }        // :207-213
        } else {
            if (s.cb >= s.tb || boxed_pre) { reward = 0; done = true; return; }   // :210-215
        }
        if (built) {                                            // reward_check (:232-239); time limit NOT tested
            const int pl = lp[tcell];
            reward = newh > pl ? -1 : (newh == pl ? 10 : 1);
            done = false;
        }
    }
    __device__ double obs_value(int lane) const {
        const double num = lane == W ? (double)s.cb : (double)s.cs;
        const double den = lane == W ? (double)s.tb : (double)TS;
        const double sc = DYN ? num / den : num;
        return lane < W ? (double)win : sc;
    }
    // iou: DMP_simulator_3d_static_circle.py:257-276
    __device__ double iou(int lane) const {
        int cross = 0;
        for (int i = lane; i < GE; i += 64) cross += min((int)lg[i], (int)lp[i]);
        cross = wave_sum(cross);
        return (double)cross / (double)(s.tb + s.cb - cross);
    }
};

// ------------------------------------------------------------------------------------------------
template <class E, int WPB>
__device__ __forceinline__ char* wave_lds() {
    if constexpr (E::LDS_BYTES > 0) {
        __shared__ __attribute__((aligned(16))) char lds[WPB * E::LDS_BYTES];
        return lds + (threadIdx.x >> 6) * E::LDS_BYTES;
    } else {
        return nullptr;
    }
}

template <class E>
__device__ __forceinline__ int pick_plan(const KArgs& a, EnvKeys pk, int episode) {
    if (E::DYN) return (int)__umulhi(rng_word(pk, (uint32_t)episode), (uint32_t)a.num_plans);
    return a.static_plan;
}

// T fused vector steps (T = 1: one step() call).  One wave per env.
template <class E, typename OT, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_rollout(const KArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl((int)blockIdx.x * WPB + (int)(threadIdx.x >> 6));
    if (env >= a.n) return;
    E e;
    e.bind(wave_lds<E, WPB>(), lane);
    e.load(a, env, lane);
    int episode = a.episode[env];
    const uint64_t gid = (uint64_t)(a.env_id_base + env);
    const EnvKeys sk = env_keys(a.key_step, gid), pk = env_keys(a.key_plan, gid);
    int d_eps = 0, d_ret = 0;
    long long d_iou = 0;
    OT* const obs = (OT*)a.obs;
    for (int t = 0; t < a.T; ++t) {
        if (a.auto_reset && (e.s.flags & SNAC_FLAG_NEED_RESET)) {
            episode += 1;
            e.reset(a, pick_plan<E>(a, pk, episode), lane);
        }
        const uint32_t w = rng_word(sk, a.t0 + (uint32_t)t);
        const size_t row = (size_t)t * (size_t)a.n + (size_t)env;
        const int act = a.actions ? (int)a.actions[row] : (int)(((w >> 16) * (uint32_t)E::A) >> 16);
        const int k = a.step_size ? (int)a.step_size[row] : 1 + (int)(((w & 0xffffu) * 3u) >> 16);
        int reward;
        bool done;
        e.step(act, k, lane, reward, done);
        e.s.ep_ret += reward;
        e.s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
        if (done) {
            d_eps += 1;
            d_ret += e.s.ep_ret;
            d_iou += __double2ll_rn(e.iou(lane) * FX40);
        }
        if (a.obs_mode == SNAC_OBS_ALL || (a.obs_mode == SNAC_OBS_LAST && t == a.T - 1)) {
            const double v = e.obs_value(lane);
            const size_t orow = a.obs_mode == SNAC_OBS_ALL ? row : (size_t)env;
            if (lane < E::D) obs[orow * E::D + lane] = (OT)v;
        }
        if (lane == 0) {
            if (a.reward) a.reward[row] = (float)reward;
            if (a.done) a.done[row] = done ? 1 : 0;
        }
    }
    e.store(a, env, lane);
    if (lane == 0) {
        a.episode[env] = episode;
        if (d_eps) {
            a.stat_episodes[env] += d_eps;
            a.stat_return[env] += d_ret;
            a.stat_iou_fx[env] += d_iou;
        }
    }
}

// reset(mask, plan_idx_in) + observation of every env
template <class E, typename OT, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_reset(const KArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl((int)blockIdx.x * WPB + (int)(threadIdx.x >> 6));
    if (env >= a.n) return;
    E e;
    e.bind(wave_lds<E, WPB>(), lane);
    const bool doit = !a.observe_only && (a.mask ? a.mask[env] != 0 : true);
    if (doit) {
        const int episode = a.episode[env] + 1;
        int pidx;
        if (a.plan_idx_in) pidx = a.plan_idx_in[env];
        else pidx = pick_plan<E>(a, env_keys(a.key_plan, (uint64_t)(a.env_id_base + env)), episode);
        pidx = min(max(pidx, 0), a.num_plans - 1);
        e.reset(a, pidx, lane);   // reads nothing from the old state
        e.store(a, env, lane);
        if (lane == 0) a.episode[env] = episode;
    } else if (a.obs) {
        e.load(a, env, lane);
    }
    if (a.obs) {
        const double v = e.obs_value(lane);
        if (lane < E::D) ((OT*)a.obs)[(size_t)env * E::D + lane] = (OT)v;
    }
}

template <class E, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_iou(const KArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl((int)blockIdx.x * WPB + (int)(threadIdx.x >> 6));
    if (env >= a.n) return;
    E e;
    e.bind(wave_lds<E, WPB>(), lane);
    e.load(a, env, lane);
    const double v = e.iou(lane);
    if (lane == 0) a.out_f64[env] = v;
}

// environment_memory with its -1 frame, float64 [N][H][W]; one thread per cell
template <int KIND>
__global__ void k_export(const KArgs a, long long total) {
    constexpr int H = KIND == 1 ? 1 : 26, Wd = KIND == 1 ? 34 : 26, HW = KIND == 1 ? 2 : 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long env = i / (H * Wd);
        const int cell = (int)(i - env * (H * Wd));
        const int r = cell / Wd, c = cell - r * Wd;
        int v = -1;
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
// host side
int check_common(const snac_env_desc* d, const snac_state* st) {
    if (!d || !st) return fail(SNAC_ERR_ARG, "null desc/state");
    if (d->kind < SNAC_ENV_1D || d->kind > SNAC_ENV_3D) return fail(SNAC_ERR_ARG, "unknown env kind");
    if (d->num_envs <= 0) return fail(SNAC_ERR_ARG, "num_envs must be positive");
    if (d->num_plans <= 0 || d->num_plans > 32767) return fail(SNAC_ERR_ARG, "num_plans out of range");
    if (d->static_plan < 0 || d->static_plan >= d->num_plans) return fail(SNAC_ERR_ARG, "static_plan out of range");
    if (d->obs_dtype != SNAC_OBS_F64 && d->obs_dtype != SNAC_OBS_F32) return fail(SNAC_ERR_ARG, "unknown obs_dtype");
    if (!st->hdr || !st->episode || !st->grid || !st->plans || !st->plan_tb || !st->stat_episodes || !st->stat_return ||
        !st->stat_iou_fx)
        return fail(SNAC_ERR_ARG, "null pointer in snac_state");
    return SNAC_OK;
}

KArgs make_args(const snac_env_desc* d, const snac_state* st) {
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = d->num_envs; a.num_plans = d->num_plans; a.static_plan = d->static_plan;
    a.key_step = stream_key(d->seed, 0); a.key_plan = stream_key(d->seed, 1);
    a.env_id_base = d->env_id_base;
    a.hdr = st->hdr; a.episode = st->episode; a.grid = st->grid; a.plans = st->plans; a.plan_tb = st->plan_tb;
    a.stat_episodes = st->stat_episodes; a.stat_return = st->stat_return; a.stat_iou_fx = st->stat_iou_fx;
    return a;
}

enum Op { OP_ROLLOUT, OP_RESET, OP_IOU };

template <class E, typename OT, int WPB>
void launch_op(Op op, const KArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.n + WPB - 1) / WPB)), block(WPB * 64);
    if (op == OP_ROLLOUT) hipLaunchKernelGGL((k_rollout<E, OT, WPB>), grid, block, 0, s, a);
    else if (op == OP_RESET) hipLaunchKernelGGL((k_reset<E, OT, WPB>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_iou<E, WPB>), grid, block, 0, s, a);
}

template <class E>
void launch_dt(Op op, int obs_dtype, const KArgs& a, hipStream_t s) {
    constexpr int WPB = 4;
    if (obs_dtype == SNAC_OBS_F32) launch_op<E, float, WPB>(op, a, s);
    else launch_op<E, double, WPB>(op, a, s);
}

int launch(Op op, const snac_env_desc* d, const KArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool dyn = d->dynamic != 0;
    switch (d->kind) {
        case SNAC_ENV_1D: dyn ? launch_dt<Env1D<true>>(op, d->obs_dtype, a, s) : launch_dt<Env1D<false>>(op, d->obs_dtype, a, s); break;
        case SNAC_ENV_2D: dyn ? launch_dt<Env2D<true>>(op, d->obs_dtype, a, s) : launch_dt<Env2D<false>>(op, d->obs_dtype, a, s); break;
        default:          dyn ? launch_dt<Env3D<true>>(op, d->obs_dtype, a, s) : launch_dt<Env3D<false>>(op, d->obs_dtype, a, s); break;
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

int snac_reset(const snac_env_desc* d, const snac_state* st, const uint8_t* mask, const int16_t* plan_idx_in, void* obs,
               void* stream) {
    if (int rc = check_common(d, st)) return rc;
    KArgs a = make_args(d, st);
    a.mask = mask; a.plan_idx_in = plan_idx_in; a.obs = obs;
    return launch(OP_RESET, d, a, stream);
}

int snac_rollout(const snac_env_desc* d, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                 const int8_t* step_size, int obs_mode, void* obs, float* reward, uint8_t* done, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (T < 0) return fail(SNAC_ERR_ARG, "T must be >= 0");
    if (obs_mode < SNAC_OBS_NONE || obs_mode > SNAC_OBS_LAST) return fail(SNAC_ERR_ARG, "unknown obs_mode");
    if (obs_mode != SNAC_OBS_NONE && !obs) return fail(SNAC_ERR_ARG, "obs_mode set but obs is null");
    if (T == 0) return SNAC_OK;
    KArgs a = make_args(d, st);
    a.T = T; a.t0 = t0; a.auto_reset = 1; a.obs_mode = obs_mode;
    a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    return launch(OP_ROLLOUT, d, a, stream);
}

int snac_step(const snac_env_desc* d, const snac_state* st, uint32_t t, const int8_t* actions, const int8_t* step_size,
              int auto_reset, void* obs, float* reward, uint8_t* done, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    KArgs a = make_args(d, st);
    a.T = 1; a.t0 = t; a.auto_reset = auto_reset ? 1 : 0; a.obs_mode = obs ? SNAC_OBS_ALL : SNAC_OBS_NONE;
    a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    return launch(OP_ROLLOUT, d, a, stream);
}

int snac_observe(const snac_env_desc* d, const snac_state* st, void* obs, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (!obs) return fail(SNAC_ERR_ARG, "null obs");
    KArgs a = make_args(d, st);
    a.obs = obs;
    a.observe_only = 1;   // k_reset with no env selected: load + observe
    return launch(OP_RESET, d, a, stream);
}

int snac_iou(const snac_env_desc* d, const snac_state* st, double* out, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (!out) return fail(SNAC_ERR_ARG, "null out");
    KArgs a = make_args(d, st);
    a.out_f64 = out;
    return launch(OP_IOU, d, a, stream);
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
