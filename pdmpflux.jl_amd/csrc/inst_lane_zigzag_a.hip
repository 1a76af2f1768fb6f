// Lane-per-chain kernels (lane_kernel.hpp) for d = 25 .. 36 (7 .. 9 slots of four).
#include "lane_kernel.hpp"

PDMP_NS_BEGIN
hipError_t launch_lane_zigzag_a(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    return launch_lane_range<7, 9>(target, gp, flag, rounds, stream);
}
PDMP_NS_END
