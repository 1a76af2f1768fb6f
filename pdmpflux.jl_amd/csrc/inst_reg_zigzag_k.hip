// Exact-K register-form kernels (K = 2..8, forward-mode slopes) of the group kernel for sampler S_ZIGZAG (see inst_reg.hpp);
// K = 9..12 are in inst_reg_zigzag_k2.hip.
#include "inst_reg.hpp"

PDMP_NS_BEGIN
hipError_t launch_group_reg_zigzag_exact_hi(int target, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_zigzag_exact_lo(int target, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_zigzag_exact(int target, const GroupParams &gp, hipStream_t stream) {
    if (gp.K < 7 && gp.G >= 2) return launch_group_reg_zigzag_exact_lo(target, gp, stream);
    return gp.K >= 9 ? launch_group_reg_zigzag_exact_hi(target, gp, stream) : exact_by_target<S_ZIGZAG, false, 0>(target, gp, stream);
}
PDMP_NS_END
