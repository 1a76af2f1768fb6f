// Explicit instantiations of the skeleton kernel for sampler S_BOOMERANG (one translation unit per sampler so
// the three compile in parallel).
#include "group_kernel.hpp"

PDMP_NS_BEGIN
PDMP_DEFINE_LAUNCHER(S_BOOMERANG, T_GAUSS_ISO)
PDMP_DEFINE_LAUNCHER(S_BOOMERANG, T_GAUSS_DIAG)
PDMP_DEFINE_LAUNCHER(S_BOOMERANG, T_BANANA)
PDMP_DEFINE_GROUP_LAUNCHER(S_BOOMERANG, T_GAUSS_ISO)
PDMP_DEFINE_GROUP_LAUNCHER(S_BOOMERANG, T_GAUSS_DIAG)
PDMP_DEFINE_GROUP_LAUNCHER(S_BOOMERANG, T_BANANA)
PDMP_DEFINE_GROUP_LAUNCHER(S_BOOMERANG, T_GAUSS_DENSE)      // round 4: the bit-exact per-chain mat-vec form (small ensembles, tape replay, d > 1536)
PDMP_NS_END
