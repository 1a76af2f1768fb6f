// Lane-pair-per-chain kernels with two coordinates of a slot per lane (lane_kernel.hpp, GL = 2, PG = 2): isotropic Gaussian and banana,
// d = 25 .. 52 (7 .. 13 slots of four), for launches that leave most SIMDs idle at 64 chains per wave.  Two waves per SIMD, no AGPRs.
#include "lane_kernel.hpp"

PDMP_NS_BEGIN
hipError_t launch_lane2q_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    return launch_lane_range<7, 13, 2, 2>(target, gp, flag, rounds, stream);
}
PDMP_NS_END
