// Exact-K register-form kernels (K = 9..12, forward-mode slopes) of the group kernel for sampler S_ZIGZAG (see inst_reg.hpp).
#include "inst_reg.hpp"

PDMP_NS_BEGIN
hipError_t launch_group_reg_zigzag_exact_hi(int target, const GroupParams &gp, hipStream_t stream) { return exact_by_target<S_ZIGZAG, false, 1>(target, gp, stream); }
PDMP_NS_END
