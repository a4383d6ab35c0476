// k_tile2d.hip -- the tile kernels (k_tile.inc) for the 2D env classes
#include "k_tile.inc"

namespace snac_detail {
void launch_tile2d(Op op, bool dyn, int E, int obs_dtype, const KArgs& a, hipStream_t s) { launch_tile<K2D, 4>(op, dyn, E, obs_dtype, a, s); }
}  // namespace snac_detail
