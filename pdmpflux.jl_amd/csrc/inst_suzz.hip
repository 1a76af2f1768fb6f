// Explicit instantiations of the group kernel for sampler S_SUZZ (SpeedUpZigZag, src/Samplers/SpeedUpZigZagSamplers.jl):
// the Zig-Zag loop with the closed-form non-linear flow (:71-79) and the effective gradient (:81-83).
#include "group_kernel.hpp"

PDMP_NS_BEGIN
PDMP_DEFINE_GROUP_LAUNCHER(S_SUZZ, T_GAUSS_ISO)
PDMP_DEFINE_GROUP_LAUNCHER(S_SUZZ, T_GAUSS_DIAG)
PDMP_DEFINE_GROUP_LAUNCHER(S_SUZZ, T_BANANA)
PDMP_NS_END
