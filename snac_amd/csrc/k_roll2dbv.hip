// k_roll2dbv.hip -- k_rollout2db for the layout variants without the plan tail (rows of 51 .. 61 values)
#include "k_roll2db.h"

namespace snac_detail {

void launch_roll2dbv(const snac_env_desc* d, const KArgs& a, int steppers, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll2db_n<true, float, true>(a, steppers, s) : launch_roll2db_n<true, double, true>(a, steppers, s);
    else f32 ? launch_roll2db_n<false, float, true>(a, steppers, s) : launch_roll2db_n<false, double, true>(a, steppers, s);
}

}  // namespace snac_detail
