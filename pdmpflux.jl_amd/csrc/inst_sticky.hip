// Explicit instantiations of the group kernel for sampler S_STICKY (StickyZigZag, src/Samplers/StickyZigZagSamplers.jl with the loop
// of src/StickySamplingLoop.jl): the Zig-Zag loop on v .* is_active with sticking at the axes and thawing at rates kappa.
#include "group_kernel.hpp"

PDMP_NS_BEGIN
PDMP_DEFINE_GROUP_LAUNCHER(S_STICKY, T_GAUSS_ISO)
PDMP_DEFINE_GROUP_LAUNCHER(S_STICKY, T_GAUSS_DIAG)
PDMP_DEFINE_GROUP_LAUNCHER(S_STICKY, T_BANANA)
PDMP_NS_END
