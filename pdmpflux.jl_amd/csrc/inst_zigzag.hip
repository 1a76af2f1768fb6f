// Explicit instantiations of the skeleton kernel for sampler S_ZIGZAG (one translation unit per sampler so
// the three compile in parallel).
#include "group_kernel.hpp"

namespace pdmp {
PDMP_DEFINE_LAUNCHER(S_ZIGZAG, T_GAUSS_ISO)
PDMP_DEFINE_LAUNCHER(S_ZIGZAG, T_GAUSS_DIAG)
PDMP_DEFINE_LAUNCHER(S_ZIGZAG, T_BANANA)
PDMP_DEFINE_GROUP_LAUNCHER(S_ZIGZAG, T_GAUSS_ISO)
PDMP_DEFINE_GROUP_LAUNCHER(S_ZIGZAG, T_GAUSS_DIAG)
PDMP_DEFINE_GROUP_LAUNCHER(S_ZIGZAG, T_BANANA)
PDMP_DEFINE_GROUP_LAUNCHER(S_ZIGZAG, T_GAUSS_DENSE)
}  // namespace pdmp
