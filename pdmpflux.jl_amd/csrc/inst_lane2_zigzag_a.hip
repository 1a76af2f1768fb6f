// Lane-pair-per-chain kernels (lane_kernel.hpp, GL = 2) for d = 53 .. 72 (7 .. 9 slots of eight).
#include "lane_kernel.hpp"

PDMP_NS_BEGIN
hipError_t launch_lane2_zigzag_a(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    return launch_lane_range<7, 9, 2>(target, gp, flag, rounds, stream);
}
PDMP_NS_END
