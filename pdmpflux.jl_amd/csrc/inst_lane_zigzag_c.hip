// Lane-per-chain kernels (lane_kernel.hpp) for d = 9 .. 24 (3 .. 6 slots of four; round 4).
#include "lane_kernel.hpp"

PDMP_NS_BEGIN
hipError_t launch_lane_zigzag_c(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    return launch_lane_range<3, 6>(target, gp, flag, rounds, stream);
}
PDMP_NS_END
