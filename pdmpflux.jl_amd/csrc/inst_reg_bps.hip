// Register-form kernels of the group kernel for sampler S_BPS (see inst_reg.hpp).
#include "inst_reg.hpp"

PDMP_NS_BEGIN
hipError_t launch_group_reg_bps(int target, bool fd, const GroupParams &gp, hipStream_t stream) { return reg_by_target<S_BPS>(target, fd, gp, stream); }
PDMP_NS_END
