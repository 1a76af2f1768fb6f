// Register-form kernels of the group kernel for sampler S_RHMC (see inst_reg.hpp): x, v in VGPRs through the velocity-Verlet steps,
// one LDS tile per wave to hand the refreshment's normals across the lanes of a chain -- round 4.
#include "inst_reg.hpp"

PDMP_NS_BEGIN
// (the slope mode plays no part in RHMC -- its bound is the constant refresh clock --: one instantiation serves both)
hipError_t launch_group_reg_rhmc(int target, const GroupParams &gp, hipStream_t stream) { return reg_by_target<S_RHMC>(target, false, gp, stream); }
PDMP_NS_END
