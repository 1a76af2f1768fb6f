// Exact-K register-form kernels for K = 2..6 slots with G = 2..16 lanes per chain (forward-mode slopes) of the group kernel for
// sampler S_BPS (see inst_reg.hpp): the layouts of the occupancy rule for launches with few chains (pdmp_suggest_group_lanes).
#include "inst_reg.hpp"

PDMP_NS_BEGIN
hipError_t launch_group_reg_bps_exact_lo(int target, const GroupParams &gp, hipStream_t stream) { return exact_by_target<S_BPS, false, 2>(target, gp, stream); }
PDMP_NS_END
