// Register form of StickyZigZag (round 4): x, v and the active mask in VGPRs, kappa staged in LDS.  Exact-K kernels only (forward-mode
// slopes; the 13-slot kernels spill): K = 2, 3 with one lane per chain and K = 4, 5 with every G here, K = 6, 7 in inst_reg_sticky_k2.hip.
#include "inst_reg.hpp"

PDMP_NS_BEGIN
template <int T>
static hipError_t sticky_lo(const GroupParams &gp, hipStream_t stream) {
    switch (gp.K) {
    case 2: return gp.G == 1 ? launch_group_impl<S_STICKY, T, false, 2, 1, true>(gp, stream) : hipErrorNotSupported;
    case 3: return gp.G == 1 ? launch_group_impl<S_STICKY, T, false, 3, 1, true>(gp, stream) : hipErrorNotSupported;
    case 4: return exact_by_lanes<S_STICKY, T, false, 4>(gp, stream);
    case 5: return exact_by_lanes<S_STICKY, T, false, 5>(gp, stream);
    default: return hipErrorNotSupported;
    }
}
hipError_t launch_group_reg_sticky_exact_hi(int target, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_sticky_exact(int target, const GroupParams &gp, hipStream_t stream) {
    if (gp.K >= 6) return launch_group_reg_sticky_exact_hi(target, gp, stream);
    switch (target) {
    case T_GAUSS_ISO: return sticky_lo<T_GAUSS_ISO>(gp, stream);
    case T_GAUSS_DIAG: return sticky_lo<T_GAUSS_DIAG>(gp, stream);
    case T_BANANA: return sticky_lo<T_BANANA>(gp, stream);
    default: return hipErrorNotSupported;
    }
}
PDMP_NS_END
