// Register form of SpeedUpZigZag (round 4, last): x, v in VGPRs.  Exact-K kernels only, for the slot counts that stay free of spills at two
// waves per SIMD (the flow's chain-level scalars and the speed duals ride on top of the Zig-Zag state): the isotropic Gaussian K = 2 .. 7
// here; the diagonal Gaussian K = 2 .. 6 and the banana K = 2 .. 4 in inst_reg_suzz_k2.hip.  Forward-mode slopes.
#include "inst_reg.hpp"

PDMP_NS_BEGIN
hipError_t launch_group_reg_suzz_exact_other(int target, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_suzz_exact(int target, const GroupParams &gp, hipStream_t stream) {
    if (target != T_GAUSS_ISO) return launch_group_reg_suzz_exact_other(target, gp, stream);
    switch (gp.K) {
    case 2: return exact_by_lanes<S_SUZZ, T_GAUSS_ISO, false, 2>(gp, stream);
    case 3: return exact_by_lanes<S_SUZZ, T_GAUSS_ISO, false, 3>(gp, stream);
    case 4: return exact_by_lanes<S_SUZZ, T_GAUSS_ISO, false, 4>(gp, stream);
    case 5: return exact_by_lanes<S_SUZZ, T_GAUSS_ISO, false, 5>(gp, stream);
    case 6: return exact_by_lanes<S_SUZZ, T_GAUSS_ISO, false, 6>(gp, stream);
    case 7: return exact_by_lanes<S_SUZZ, T_GAUSS_ISO, false, 7>(gp, stream);
    default: return hipErrorNotSupported;
    }
}
PDMP_NS_END
