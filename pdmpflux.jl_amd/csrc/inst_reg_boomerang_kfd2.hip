// Exact-K register-form kernels (K = 9..12, finite-difference slopes) of the group kernel for sampler S_BOOMERANG (see inst_reg.hpp).
#include "inst_reg.hpp"

PDMP_NS_BEGIN
hipError_t launch_group_reg_boomerang_exact_fd_hi(int target, const GroupParams &gp, hipStream_t stream) { return exact_by_target<S_BOOMERANG, true, 1>(target, gp, stream); }
PDMP_NS_END
