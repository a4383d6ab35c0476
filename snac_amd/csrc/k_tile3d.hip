// k_tile3d.hip -- the tile kernels (k_tile.inc) for the 3D env classes
#include "k_tile.inc"

namespace snac_detail {
// 3D: 2.1 KB of LDS per env -> tiles of 16 (or 8 for small batches: two waves per SIMD sooner)
void launch_tile3d(Op op, bool dyn, int E, int obs_dtype, const KArgs& a, hipStream_t s) {
    if (E == 8 && a.n < 8192) dyn ? launch_dt<K3D, true, 8, 1>(op, obs_dtype, a, s) : launch_dt<K3D, false, 8, 1>(op, obs_dtype, a, s);
    else if (E == 8) dyn ? launch_dt<K3D, true, 8, 4>(op, obs_dtype, a, s) : launch_dt<K3D, false, 8, 4>(op, obs_dtype, a, s);
    else dyn ? launch_dt<K3D, true, 16, 2>(op, obs_dtype, a, s) : launch_dt<K3D, false, 16, 2>(op, obs_dtype, a, s);
}
}  // namespace snac_detail
