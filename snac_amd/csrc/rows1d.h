// rows1d.h -- the 1D observation rows of a wave of 64 envs (lane = env) as ONE run of 64 x 7 values: k_rollout1dl, k_step1d.  Internal.
#pragma once
#include "snac_dev.h"

namespace {

template <typename OT>
struct Rows1D {
    static constexpr int D = 7, E = 64, ROWB = D * (int)sizeof(OT), RUN = E * ROWB, NF = (RUN + 1023) / 1024;   // 3584 / 1792 bytes: 4 / 2 pieces per lane
    uint4 fv[NF];
    // stage: the lane files its 7 values, then the wave's run is read back 16 bytes per lane.  The lanes exchange their values through LDS
    // without a barrier (one wave: its LDS operations complete in order); the reads are volatile and spell out the address space so that
    // the compiler performs them where they stand (emit_tile, snac_dev.h)
    __device__ __forceinline__ void stage(char* stg, int lane, const int (&win)[5], double v0, double v1) {
        OT* const o = (OT*)(stg + lane * ROWB);
#pragma unroll
        for (int i = 0; i < 5; ++i) o[i] = (OT)win[i];
        o[5] = (OT)v0; o[6] = (OT)v1;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            typedef uint32_t u32x4_ld __attribute__((ext_vector_type(4)));
            const u32x4_ld t = *(const volatile __attribute__((address_space(3))) u32x4_ld*)(stg + min(i * 1024 + lane * 16, RUN - 16));
            fv[i] = make_uint4(t.x, t.y, t.z, t.w);
        }
    }
    // flush: the run leaves (between stage and flush the caller does work that does not need the rows: the LDS round trip is hidden)
    template <bool NT>
    __device__ __forceinline__ void flush(char* g, int lane, int nenv) const {
        if (nenv == E) {                                             // a full tile: no test per piece but the last one's lanes
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if ((i + 1) * 1024 <= RUN || i * 1024 + lane * 16 < RUN) store16<NT>(g + i * 1024 + lane * 16, fv[i]);
        } else {
            const int valid = nenv * ROWB;                           // a multiple of 16: N % 4 == 0
#pragma unroll
            for (int i = 0; i < NF; ++i)
                if (i * 1024 + lane * 16 < valid) store16<NT>(g + i * 1024 + lane * 16, fv[i]);
        }
    }
};

// The layout variants of snac_env_desc for the same wave (rows of LD = 7 + tail values: the position, the plan's 30 heights, the record's 8
// values, in the descriptor's order; frame cells shown as frame_val): the lane files its whole row, the run of nrow x LD values leaves 16 bytes
// per lane.  `fill` is also what writes a row value by value where the run cannot leave as pieces.
constexpr int ROWS1D_VAR_MAX_LD = 46;
template <typename OT>
__device__ __forceinline__ void fill_row1d_var(OT* o, int tail, int frame_val, const int (&win)[5], double v0, double v1, int pos, const int16_t* prow, const int (&recv)[8]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = (OT)(double)(win[i] < 0 ? frame_val : win[i]);
    o[5] = (OT)v0; o[6] = (OT)v1;
    OT* q = o + 7;
    if (tail & SNAC_TAIL_POSITION) { q[0] = (OT)(double)pos; q += 1; }
    if (tail & SNAC_TAIL_PLAN) {
#pragma unroll
        for (int c = 0; c < 30; ++c) q[c] = (OT)(double)prow[c];
        q += 30;
    }
    if (tail & SNAC_TAIL_RECORD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = (OT)(double)recv[j];
    }
}
template <typename OT, bool NT>
__device__ __forceinline__ void flush_rows1d_var(const char* stg, char* g, int lane, int nrow, int LD) {
    const int total = nrow * LD * (int)sizeof(OT);                  // a multiple of 16: whole groups of four rows
    for (int off = lane * 16; off < total; off += 1024) {
        typedef uint32_t u32x4_ld __attribute__((ext_vector_type(4)));
        const u32x4_ld t = *(const volatile __attribute__((address_space(3))) u32x4_ld*)(stg + off);
        store16<NT>(g + off, make_uint4(t.x, t.y, t.z, t.w));
    }
}

}  // namespace
