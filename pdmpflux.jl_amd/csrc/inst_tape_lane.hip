// Tape-replay instantiations of the one-lane-per-chain kernel (lane_kernel.hpp, TAPE = true): d = 49 .. 52 (13 slots of four), with
// the diagnostic arrays (a replay is compared field by field).  pdmp_set_tape + a launch the lane kernel would serve anyway.
#include "lane_kernel.hpp"

PDMP_NS_BEGIN
hipError_t launch_lane_zigzag_tape(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    if ((gp.s.d + 3) / 4 != 13) return hipErrorNotSupported;
    switch (target) {
    case T_GAUSS_ISO: return launch_lane<T_GAUSS_ISO, 13, true, true>(gp, flag, rounds, stream);
    case T_BANANA: return launch_lane<T_BANANA, 13, true, true>(gp, flag, rounds, stream);
    default: return hipErrorNotSupported;
    }
}
PDMP_NS_END
