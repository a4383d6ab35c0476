// k_misc.hip -- environment_memory export, state import, observation equality, plan generators, replay gather: kernels and entry points
#include "snac_dev.h"

using namespace snac_detail;

namespace {

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
// default-policy evaluation of tree leaves (script/MCTS/utils/mcts.py:100-110: `estimate += reward * (gamma**t)` while not terminal): the
// discounted sums of a rollout's reward / done [H][m], one leaf per lane, sequential in t, product and sum each rounded to float64
// (fp contract(off): no fma -- the reference's python floats round twice).  A wave leaves when all its leaves have ended.
__global__ __launch_bounds__(256) void k_discount(int H, int m, const float* reward, const uint8_t* done, const uint8_t* terminal, const double* gpow,
                                                   double* est, int64_t* steps) {
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const bool in = i < m;
    const int ii = in ? i : 0;
    bool alive = in && !(terminal && terminal[ii]);
    double e = est[ii];
    long long n = 0;
    for (int t = 0; t < H; ++t) {
        if (!__any(alive)) break;
        const size_t at = (size_t)t * (size_t)m + (size_t)ii;
        const float r = reward[at];
        const bool d = done[at] != 0;
        if (alive) {
#pragma clang fp contract(off)                                      // (hipcc contracts a * b + c into an fma by default, also through __dmul_rn / __dadd_rn: one rounding instead of two)
            const double p = (double)r * gpow[t];
            e = e + p;
            n += 1; alive = !d;
        }
    }
    if (in) { est[i] = e; if (steps) steps[i] = n; }
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
// plan rows from grids (hindsight relabelling): the DRQN_hindsight scripts overwrite the hindsight env's
// plan with the finished episode's own final grid and replay the recorded actions against it
// (script/DRQN_hindsight/2d/DRQN_hindsight_2D_dynamic.py:270-282: plan[3:23, 3:23] = environment_memory[3:23, 3:23]).  Here the
// "overwrite" is one launch: plan row first + i <- max(interior of grid src_rows[i], 0), taken from the packed records of a batch
// (its state arrays) or from environment_memory in the reference's own format; total_brick of the row from the caller, or the source
// env's header (the total_brick its episode ran with).  One wave per plan row.
struct FArgs {
    int32_t m, src_envs, first;
    const int32_t* src_rows;   // [m] env of the source batch (NULL: i)
    const void* grid;          // packed records of the source batch, or NULL
    const int4* hdr;           // its headers (total_brick when tb is NULL), or NULL
    const double* mem;         // [m][H][W] environment_memory with its frame (when grid is NULL)
    const int32_t* tb;         // [m] total_brick per row or NULL
    void* plans;
    int16_t* plan_tb;
};

template <int KIND>
__global__ __launch_bounds__(256) void k_plans_from_grids(const FArgs g) {
    const int lane = threadIdx.x & 63;
    const int i = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (i >= g.m) return;
    const int src = g.src_rows ? min(max(g.src_rows[i], 0), g.src_envs - 1) : i;
    const size_t row = (size_t)(g.first + i);
    if constexpr (KIND == 1) {
        int16_t* const dst = (int16_t*)g.plans + row * 32;
        if (lane < 32) {
            int v = 0;
            if (lane < 30) v = g.grid ? (int)((const int16_t*)g.grid)[(size_t)src * 32 + lane] : (int)g.mem[(size_t)i * 34 + lane + 2];
            dst[lane] = (int16_t)min(max(v, 0), CNT_MAX);
        }
    } else if constexpr (KIND == 2) {
        uint32_t* const dst = (uint32_t*)g.plans + row * 20;
        if (lane < 20) {
            uint32_t w = 0u;
            if (g.grid) w = ((const uint32_t*)g.grid)[(size_t)src * 20 + lane] & 0xFFFFFu;
            else {
                const double* const mr = g.mem + (size_t)i * 676 + (size_t)(lane + 3) * 26 + 3;
                for (int c = 0; c < 20; ++c) w |= (mr[c] > 0.0 ? 1u : 0u) << c;
            }
            dst[lane] = w;
        }
    } else {
        int16_t* const dst = (int16_t*)g.plans + row * 400;
        for (int cell = lane; cell < 400; cell += 64) {
            const int r = cell / 20, c = cell - 20 * r;
            const int v = g.grid ? (int)((const int16_t*)g.grid)[(size_t)src * 400 + cell] : (int)g.mem[(size_t)i * 676 + (size_t)(r + 3) * 26 + c + 3];
            dst[cell] = (int16_t)min(max(v, 0), CNT_MAX);
        }
    }
    if (lane == 0) {
        int tb = g.tb ? g.tb[i] : (g.hdr ? (int)(int16_t)(g.hdr[src].z & 0xffff) : 1);
        g.plan_tb[row] = (int16_t)min(max(tb, 1), CNT_MAX);          // the step kernels divide by it
    }
}

}  // namespace

extern "C" {

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

int snac_discounted_return(int32_t H, int32_t m, const float* reward, const uint8_t* done, const uint8_t* terminal, const double* gpow,
                           double* est, int64_t* steps, void* stream) {
    if (H < 0 || m < 0) return fail(SNAC_ERR_ARG, "H and m must be >= 0");
    if (m == 0) return SNAC_OK;
    if (!est || (H > 0 && (!reward || !done || !gpow))) return fail(SNAC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_discount, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H, m, reward, done, terminal, gpow, est, steps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "discounted return launch");
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

int snac_plans_from_grids(const snac_env_desc* d, const snac_state* st, int32_t m, int32_t first, const snac_state* src,
                          int32_t src_envs, const int32_t* src_rows, const double* environment_memory, const int32_t* total_brick,
                          void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (m < 0 || first < 0 || (int64_t)first + m > d->num_plans) return fail(SNAC_ERR_ARG, "plan rows out of range");
    if ((src != nullptr) == (environment_memory != nullptr)) return fail(SNAC_ERR_ARG, "exactly one of src / environment_memory");
    if (src && (!src->grid || !src->hdr || src_envs <= 0)) return fail(SNAC_ERR_ARG, "src needs grid, hdr and src_envs > 0");
    if (src && !src_rows && m > src_envs) return fail(SNAC_ERR_ARG, "m exceeds the source batch");
    if (!src && src_rows) return fail(SNAC_ERR_ARG, "src_rows index a source batch");
    if (!src && !total_brick) return fail(SNAC_ERR_ARG, "environment_memory needs total_brick");
    if (m == 0) return SNAC_OK;
    FArgs g;
    g.m = m; g.src_envs = src ? src_envs : m; g.first = first; g.src_rows = src_rows;
    g.grid = src ? src->grid : nullptr; g.hdr = src ? (const int4*)src->hdr : nullptr; g.mem = environment_memory; g.tb = total_brick;
    g.plans = const_cast<void*>(st->plans); g.plan_tb = const_cast<int16_t*>(st->plan_tb);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((m + 3) / 4)), block(256);
    if (d->kind == SNAC_ENV_1D) hipLaunchKernelGGL((k_plans_from_grids<1>), grid, block, 0, s, g);
    else if (d->kind == SNAC_ENV_2D) hipLaunchKernelGGL((k_plans_from_grids<2>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((k_plans_from_grids<3>), grid, block, 0, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "plans_from_grids launch");
    return SNAC_OK;
}

}  // extern "C"
