// k_tile1d.hip -- the tile kernels (k_tile.inc) for the 1D env classes
#include "k_tile.inc"

namespace snac_detail {
void launch_tile1d(Op op, bool dyn, int E, int obs_dtype, const KArgs& a, hipStream_t s) { launch_tile<K1D, 4>(op, dyn, E, obs_dtype, a, s); }
}  // namespace snac_detail
