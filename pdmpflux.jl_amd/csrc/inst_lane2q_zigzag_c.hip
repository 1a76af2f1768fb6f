// Lane-pair-per-chain kernels with two coordinates of a slot per lane (lane_kernel.hpp, GL = 2, PG = 2) for d = 9 .. 24 (3 .. 6 slots of
// four; round 4): mid-sized launches.
#include "lane_kernel.hpp"

PDMP_NS_BEGIN
hipError_t launch_lane2q_zigzag_c(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    return launch_lane_range<3, 6, 2, 2>(target, gp, flag, rounds, stream);
}
PDMP_NS_END
