// One-lane-per-chain BPS kernels (lane_bps_kernel.hpp) for d = 25 .. 36 (7 .. 9 slots of four).
#include "lane_bps_kernel.hpp"

namespace pdmp {
hipError_t launch_lane_bps_a(int target, const GroupParams &gp, int rounds, hipStream_t stream) { return launch_lane_bps_range<7, 9>(target, gp, rounds, stream); }
}  // namespace pdmp
