// One-lane-per-chain kernels of the Bouncy Particle Sampler (lane_bps_kernel.hpp): isotropic Gaussian and banana, d = 49 .. 52 (13 slots of four).
#include "lane_bps_kernel.hpp"

namespace pdmp {
hipError_t launch_lane_bps_13(int target, const GroupParams &gp, int rounds, hipStream_t stream) { return launch_lane_bps_range<13, 13>(target, gp, rounds, stream); }
}  // namespace pdmp
