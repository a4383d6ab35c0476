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

}  // namespace
