// Explicit instantiations of the group kernel for sampler S_RHMC (src/Samplers/RandomizedHamiltonianMonteCarlo.jl):
// velocity-Verlet flow, constant refresh clock, Horowitz refreshment.
#include "group_kernel.hpp"

PDMP_NS_BEGIN
PDMP_DEFINE_GROUP_LAUNCHER(S_RHMC, T_GAUSS_ISO)
PDMP_DEFINE_GROUP_LAUNCHER(S_RHMC, T_GAUSS_DIAG)
PDMP_DEFINE_GROUP_LAUNCHER(S_RHMC, T_BANANA)
PDMP_NS_END
