// One-lane-per-chain BPS kernels (lane_bps_kernel.hpp) for d = 37 .. 48 (10 .. 12 slots of four).
#include "lane_bps_kernel.hpp"

namespace pdmp {
hipError_t launch_lane_bps_b(int target, const GroupParams &gp, int rounds, hipStream_t stream) { return launch_lane_bps_range<10, 12>(target, gp, rounds, stream); }
}  // namespace pdmp
