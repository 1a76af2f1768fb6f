// Register form of ForwardECMC (round 4): x, v in VGPRs; the jump's four d-vectors (n, v_orth, e1, e2) in LDS tiles next to the exchange
// tile -- five tiles of K x 64 doubles per wave, so the lanes-per-chain rule keeps K <= 7 (two blocks per CU; choose_group_lanes).
// Exact-K kernels only (forward-mode slopes): K = 2, 3 with one lane per chain and K = 4, 5 with every G here, K = 6, 7 in
// inst_reg_fecmc_k2.hip.
#include "inst_reg.hpp"

PDMP_NS_BEGIN
template <int T>
static hipError_t fecmc_lo(const GroupParams &gp, hipStream_t stream) {
    switch (gp.K) {
    case 2: return gp.G == 1 ? launch_group_impl<S_FECMC, T, false, 2, 1, true>(gp, stream) : hipErrorNotSupported;
    case 3: return gp.G == 1 ? launch_group_impl<S_FECMC, T, false, 3, 1, true>(gp, stream) : hipErrorNotSupported;
    case 4: return exact_by_lanes<S_FECMC, T, false, 4>(gp, stream);
    case 5: return exact_by_lanes<S_FECMC, T, false, 5>(gp, stream);
    default: return hipErrorNotSupported;
    }
}
hipError_t launch_group_reg_fecmc_exact_hi(int target, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_fecmc_exact(int target, const GroupParams &gp, hipStream_t stream) {
    if (gp.K >= 6) return launch_group_reg_fecmc_exact_hi(target, gp, stream);
    switch (target) {
    case T_GAUSS_ISO: return fecmc_lo<T_GAUSS_ISO>(gp, stream);
    case T_GAUSS_DIAG: return fecmc_lo<T_GAUSS_DIAG>(gp, stream);
    case T_BANANA: return fecmc_lo<T_BANANA>(gp, stream);
    default: return hipErrorNotSupported;
    }
}
PDMP_NS_END
