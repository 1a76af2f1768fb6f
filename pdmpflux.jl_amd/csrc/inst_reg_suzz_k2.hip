// Register form of SpeedUpZigZag on the diagonal Gaussian (K = 2 .. 6) and the banana (K = 2 .. 4): see inst_reg_suzz_k.hip.
#include "inst_reg.hpp"

PDMP_NS_BEGIN
hipError_t launch_group_reg_suzz_exact_other(int target, const GroupParams &gp, hipStream_t stream) {
    if (target == T_GAUSS_DIAG) {
        switch (gp.K) {
        case 2: return exact_by_lanes<S_SUZZ, T_GAUSS_DIAG, false, 2>(gp, stream);
        case 3: return exact_by_lanes<S_SUZZ, T_GAUSS_DIAG, false, 3>(gp, stream);
        case 4: return exact_by_lanes<S_SUZZ, T_GAUSS_DIAG, false, 4>(gp, stream);
        case 5: return exact_by_lanes<S_SUZZ, T_GAUSS_DIAG, false, 5>(gp, stream);
        case 6: return exact_by_lanes<S_SUZZ, T_GAUSS_DIAG, false, 6>(gp, stream);
        default: return hipErrorNotSupported;
        }
    }
    if (target == T_BANANA) {
        switch (gp.K) {
        case 2: return exact_by_lanes<S_SUZZ, T_BANANA, false, 2>(gp, stream);
        case 3: return exact_by_lanes<S_SUZZ, T_BANANA, false, 3>(gp, stream);
        case 4: return exact_by_lanes<S_SUZZ, T_BANANA, false, 4>(gp, stream);
        default: return hipErrorNotSupported;
        }
    }
    return hipErrorNotSupported;
}
PDMP_NS_END
