// k_roll3db.hip -- k_rollout3db for the canonical rows
#include "k_roll3db.h"

namespace snac_detail {

void launch_roll3db(const snac_env_desc* d, const KArgs& a, hipStream_t s) {
    const bool dyn = d->dynamic != 0, f32 = d->obs_dtype == SNAC_OBS_F32;
    if (dyn) f32 ? launch_roll3db_w<true, float, 0>(a, s) : launch_roll3db_w<true, double, 0>(a, s);
    else f32 ? launch_roll3db_w<false, float, 0>(a, s) : launch_roll3db_w<false, double, 0>(a, s);
}

}  // namespace snac_detail
