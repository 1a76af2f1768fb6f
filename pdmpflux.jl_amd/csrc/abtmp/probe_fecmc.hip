#include "../inst_reg.hpp"
namespace pdmp {
hipError_t probe_fecmc(int target, const GroupParams &gp, hipStream_t stream) {
    if (target == 0) return launch_group_impl<S_FECMC, T_GAUSS_ISO, false, 7, 8, true>(gp, stream);
    if (target == 1) return launch_group_impl<S_FECMC, T_GAUSS_DIAG, false, 7, 8, true>(gp, stream);
    return launch_group_impl<S_FECMC, T_BANANA, false, 7, 8, true>(gp, stream);
}
}
