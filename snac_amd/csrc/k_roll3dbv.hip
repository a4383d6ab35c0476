// k_roll3dbv.hip -- k_rollout3db for the layout variants of snac_env_desc (MODE 1: rows without the plan tail, 2: with it)
#include "k_roll3db.h"

namespace snac_detail {

void launch_roll3dbv(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (a.tail & SNAC_TAIL_PLAN) {
        if (dyn) f32 ? launch_roll3db_w<true, float, 2>(a, s) : launch_roll3db_w<true, double, 2>(a, s);
        else f32 ? launch_roll3db_w<false, float, 2>(a, s) : launch_roll3db_w<false, double, 2>(a, s);
    } else {
        if (dyn) f32 ? launch_roll3db_w<true, float, 1>(a, s) : launch_roll3db_w<true, double, 1>(a, s);
        else f32 ? launch_roll3db_w<false, float, 1>(a, s) : launch_roll3db_w<false, double, 1>(a, s);
    }
}

}  // namespace snac_detail
