// k_reset.hip -- k_reset: snac_reset / snac_reset_scalar of EVERY env of a batch (no mask), the canonical layout; k_iou: snac_iou (round 6)
#include "snac_dev.h"
#include "rows1d.h"

#ifndef SNAC_RESET_NTZ_MIN_BYTES
#define SNAC_RESET_NTZ_MIN_BYTES (256u << 20)   // what the call writes (records + rows) overflows the 256 MB Infinity Cache
#endif
namespace {

// ------------------------------------------------------------------------------------------------
// reset() of a whole batch (Env/*/: reset() -- DMP_Env_1D_static.py:66-83, DMP_Env_2D_dynamic_usedata_plan.py:57-83,
// DMP_simulator_3d_dynamic_triangle_usedata.py:88-140) needs nothing of the old state but the episode counter: the header is K::reset's, the
// record is empty, and the observation is the same window for every env -- the agent at its start position over an empty map, the frame on
// two sides -- with both counters at zero.  The tile kernel k_aux (rounds 1-2) loads every record into LDS, clears it env by env and
// stores it back through narrow accesses: 412 us for 524 288 3D envs, 81 us in 2D, 45 us in 1D.  Here a wave takes 64 envs, lane = env:
//   header / episode   K::reset() on a cleared header, the plan row from plan_idx / the scalar / the counter RNG (pick_plan), 16 + 4 bytes per lane
//   records            the tile's records are one contiguous run: zeroed 16 bytes per lane (4 / 5 / 50 stores per lane for 1D / 2D / 3D)
//   rows               emit_tile (2D / 3D) or Rows1D (1D) with the constant window; the scalar slots by the expressions of the row writers
//                      (0 / total_brick and 0 / total_step in the dynamic classes: the same bits, whatever total_brick is)
// A reset with a mask (the other envs report their current observation), the layout variants, N % 4 != 0 or an unaligned obs stay on k_aux.
// NTZ: the records zeroed by non-temporal stores -- when what the call writes does not fit the 256 MB Infinity Cache anyway (3D: 524 288 envs
// 119 -> 87 us, 262 144 envs 52 -> 46; 196 608 envs, 237 MB, 32 -> 36 and 65 536 envs, whose records the next step finds cached, 12.8 -> 14.6:
// not there; r06_reset.txt).
template <int KIND, bool DYN, typename OT, int WPB, bool NTZ>
__global__ __launch_bounds__(WPB * 64) void k_reset(const KArgs a) {
    using K = typename std::conditional<KIND == 1, K1D<DYN, 64>, typename std::conditional<KIND == 2, K2D<DYN, 64>, K3D<DYN, 8>>::type>::type;
    constexpr int E = 64, PIECES = KIND == 1 ? 4 : (KIND == 2 ? 5 : 50);   // 16-byte pieces of a record
    constexpr int STG_BYTES = KIND == 1 ? E * 7 * (int)sizeof(OT) : TILE_STG_BYTES;
    __shared__ __attribute__((aligned(16))) char lds_all[WPB * STG_BYTES];
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int env0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * WPB + wv) * E);
    if (env0 >= a.n) return;
    const int nenv = min(E, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    char* const stg = lds_all + wv * STG_BYTES;
    Lane s;
    s.clear();
    if (active) {
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
    // ---- the records: one run of nenv x PIECES pieces
    {
        uint4* const g4 = (uint4*)a.grid + (size_t)env0 * PIECES;
        const int total = nenv * PIECES;
#pragma unroll 10
        for (int i = 0; i < PIECES; ++i) {
            const int g = i * 64 + lane;
            if (g < total) store16<NTZ>((char*)(g4 + g), make_uint4(0u, 0u, 0u, 0u));
        }
    }
    if (!a.obs) return;
    const double z = 0.0;
    const double v0 = DYN ? z / (double)s.tb : z, v1 = DYN ? z / (double)a.total_step : z;
    if constexpr (KIND == 1) {
        const int win[5] = {-1, -1, 0, 0, 0};                         // the agent at cell 0: two frame cells to its left
        Rows1D<OT> rows;
        rows.stage(stg, lane, win, v0, v1);
        rows.template flush<false>((char*)a.obs + (size_t)env0 * 7 * sizeof(OT), lane, nenv);
    } else {
        emit_tile<OT, false>(stg, (char*)a.obs + (size_t)env0 * 51 * sizeof(OT), lane, nenv,
                             [&](int el) { const int i = el / 7, j = el - 7 * i; return (i < 3 || j < 3) ? -1 : 0; }, v0, v1);   // the agent at (0, 0): the frame above and left
    }
}

// k_iou: snac_iou of a batch (env.iou(): DMP_Env_1D_static.py:138-151, the 2D scripts' boolean IoU script/DQN/2d/DQN_2d_dynamic.py:63-71,
// DMP_simulator_3d_*.py:257-276), lane = env, no LDS.  k_aux loads every record into LDS first -- also in 3D, where the value comes from the
// header's running sum alone.  Here: 3D the header; 2D the env's 20 row words against its plan's (popcounts); 1D its 30 heights against the
// plan's.  The formulas are K::iou's.
template <int KIND>
__global__ __launch_bounds__(256) void k_iou(const KArgs a) {
    const int env = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (env >= a.n) return;
    Lane s;
    s.unpack(a.hdr[env]);
    double v;
    if constexpr (KIND == 3) {
        v = (double)s.cross / (double)(s.tb + s.cb - s.cross);
    } else if constexpr (KIND == 2) {
        const uint4* const g4 = (const uint4*)a.grid + (size_t)env * 5;
        const uint32_t* const p = (const uint32_t*)a.plans + (size_t)s.pidx * 20;
        int inter = 0, uni = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint4 g = g4[i];
            inter += __popc(g.x & p[4 * i]) + __popc(g.y & p[4 * i + 1]) + __popc(g.z & p[4 * i + 2]) + __popc(g.w & p[4 * i + 3]);
            uni += __popc(g.x | p[4 * i]) + __popc(g.y | p[4 * i + 1]) + __popc(g.z | p[4 * i + 2]) + __popc(g.w | p[4 * i + 3]);
        }
        v = (double)inter / (double)uni;
    } else {
        const uint4* const g4 = (const uint4*)a.grid + (size_t)env * 4;
        const uint32_t* const p = (const uint32_t*)((const int16_t*)a.plans + (size_t)s.pidx * 32);
        int a1 = 0, a2 = 0, kk = 0;
        auto cell2 = [&](uint32_t gw, uint32_t pw, bool second) {     // two cells per dword
            const int g0 = (int)(int16_t)(gw & 0xFFFFu), p0 = (int)(int16_t)(pw & 0xFFFFu);
            a1 += p0; a2 += g0; kk += max(g0 - p0, 0);
            if (second) { const int g1 = (int)(int16_t)(gw >> 16), p1 = (int)(int16_t)(pw >> 16); a1 += p1; a2 += g1; kk += max(g1 - p1, 0); }
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 g = g4[i];
            cell2(g.x, p[4 * i], true); cell2(g.y, p[4 * i + 1], true); cell2(g.z, p[4 * i + 2], true);
            if (i < 3) cell2(g.w, p[4 * i + 3], true);                // dword 15 holds the record's two padding cells
        }
        const int cross = a2 - kk;
        v = (double)cross / (double)(a1 + a2 - cross);
    }
    a.out_f64[env] = v;
}

template <int KIND, bool DYN, typename OT>
void launch_r(const KArgs& a, hipStream_t s) {
    const int tiles = (a.n + 63) / 64;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    constexpr size_t REC = KIND == 1 ? 64 : (KIND == 2 ? 80 : 800);
    const size_t rowb = a.obs ? (size_t)(KIND == 1 ? 7 : 51) * sizeof(OT) : 0;
    if ((size_t)a.n * (REC + rowb) >= (size_t)SNAC_RESET_NTZ_MIN_BYTES) hipLaunchKernelGGL((k_reset<KIND, DYN, OT, 4, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_reset<KIND, DYN, OT, 4, false>), grid, block, 0, s, a);
}
template <int KIND>
void launch_rk(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_r<KIND, true, float>(a, s) : launch_r<KIND, true, double>(a, s);
    else f32 ? launch_r<KIND, false, float>(a, s) : launch_r<KIND, false, double>(a, s);
}

}  // namespace

namespace snac_detail {

void launch_reset(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    if (d->kind == SNAC_ENV_1D) launch_rk<1>(d, a, s);
    else if (d->kind == SNAC_ENV_2D) launch_rk<2>(d, a, s);
    else launch_rk<3>(d, a, s);
}

void launch_iou(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.n + 255) / 256)), block(256);
    if (d->kind == SNAC_ENV_1D) hipLaunchKernelGGL((k_iou<1>), grid, block, 0, s, a);
    else if (d->kind == SNAC_ENV_2D) hipLaunchKernelGGL((k_iou<2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_iou<3>), grid, block, 0, s, a);
}

}  // namespace snac_detail
