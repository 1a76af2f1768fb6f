// Explicit instantiations of the group kernel for sampler S_FECMC (ForwardECMC, src/Samplers/ForwardEventChainMonteCarlo.jl):
// the flow, rate and bound of the Bouncy Particle Sampler without refreshment, its own velocity jump (fecmc_jump).
#include "group_kernel.hpp"

PDMP_NS_BEGIN
PDMP_DEFINE_GROUP_LAUNCHER(S_FECMC, T_GAUSS_ISO)
PDMP_DEFINE_GROUP_LAUNCHER(S_FECMC, T_GAUSS_DIAG)
PDMP_DEFINE_GROUP_LAUNCHER(S_FECMC, T_BANANA)
PDMP_NS_END
