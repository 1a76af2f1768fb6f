// Register form of ForwardECMC, K = 6, 7 slots per lane (see inst_reg_fecmc_k.hip).
#include "inst_reg.hpp"

PDMP_NS_BEGIN
template <int T>
static hipError_t fecmc_hi(const GroupParams &gp, hipStream_t stream) {
    switch (gp.K) {
    case 6: return exact_by_lanes<S_FECMC, T, false, 6>(gp, stream);
    case 7: return exact_by_lanes<S_FECMC, T, false, 7>(gp, stream);
    default: return hipErrorNotSupported;
    }
}
hipError_t launch_group_reg_fecmc_exact_hi(int target, const GroupParams &gp, hipStream_t stream) {
    switch (target) {
    case T_GAUSS_ISO: return fecmc_hi<T_GAUSS_ISO>(gp, stream);
    case T_GAUSS_DIAG: return fecmc_hi<T_GAUSS_DIAG>(gp, stream);
    case T_BANANA: return fecmc_hi<T_BANANA>(gp, stream);
    default: return hipErrorNotSupported;
    }
}
PDMP_NS_END
