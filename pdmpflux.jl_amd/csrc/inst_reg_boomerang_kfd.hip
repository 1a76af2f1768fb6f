// Exact-K register-form kernels (K = 2..8, finite-difference slopes) of the group kernel for sampler S_BOOMERANG (see inst_reg.hpp);
// K = 9..12 are in inst_reg_boomerang_kfd2.hip.
#include "inst_reg.hpp"

PDMP_NS_BEGIN
hipError_t launch_group_reg_boomerang_exact_fd_hi(int target, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_boomerang_exact_fd(int target, const GroupParams &gp, hipStream_t stream) {
    return gp.K >= 9 ? launch_group_reg_boomerang_exact_fd_hi(target, gp, stream) : exact_by_target<S_BOOMERANG, true, 0>(target, gp, stream);
}
PDMP_NS_END
