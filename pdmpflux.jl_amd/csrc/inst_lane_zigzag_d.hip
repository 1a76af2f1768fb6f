// Lane-per-chain kernels (lane_kernel.hpp) for d = 1 .. 8 (1, 2 slots of four; round 4): rows shorter than a 64-byte line.
#include "lane_kernel.hpp"

PDMP_NS_BEGIN
hipError_t launch_lane_zigzag_d(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    return launch_lane_range<1, 2>(target, gp, flag, rounds, stream);
}
PDMP_NS_END
