// Lane-PAIR-per-chain kernels of the Zig-Zag hot path (lane_kernel.hpp, GL = 2): isotropic Gaussian and banana, d = 97 .. 104 (13 slots of eight).
#include "lane_kernel.hpp"

PDMP_NS_BEGIN
hipError_t launch_lane2_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    return launch_lane_range<13, 13, 2>(target, gp, flag, rounds, stream);
}
PDMP_NS_END
