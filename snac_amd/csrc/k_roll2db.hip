// k_roll2db.hip -- k_rollout2db for the canonical rows
#include "k_roll2db.h"

namespace snac_detail {

void launch_roll2db(const snac_env_desc* d, const KArgs& a, int steppers, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll2db_n<true, float, false>(a, steppers, s) : launch_roll2db_n<true, double, false>(a, steppers, s);
    else f32 ? launch_roll2db_n<false, float, false>(a, steppers, s) : launch_roll2db_n<false, double, false>(a, steppers, s);
}

}  // namespace snac_detail
