// Register-form instantiations of the group kernel for the Zig-Zag sampler (x, v in VGPRs; GroupChain<..., KC>): up to
// KREG slots per lane, lanes per chain G fixed at compile time; a second instantiation for K == KREG exactly (d = 50, G = 4; d = 100, G = 8; ...).
#include "group_kernel.hpp"

namespace pdmp {

template <int T>
static hipError_t by_lanes(int G, const GroupParams &gp, hipStream_t stream) {
    switch (G) {
    case 1: return launch_group_impl<S_ZIGZAG, T, false, KREG, 1>(gp, stream);
    case 2: return launch_group_impl<S_ZIGZAG, T, false, KREG, 2>(gp, stream);
    case 4: return launch_group_impl<S_ZIGZAG, T, false, KREG, 4>(gp, stream);
    case 8: return launch_group_impl<S_ZIGZAG, T, false, KREG, 8>(gp, stream);
    case 16: return launch_group_impl<S_ZIGZAG, T, false, KREG, 16>(gp, stream);
    default: return hipErrorNotSupported;
    }
}

// hipErrorNotSupported: no register form for this configuration (the caller falls back to the LDS form)
hipError_t launch_group_reg(int sampler, int target, bool fd, const GroupParams &gp, hipStream_t stream) {
    // forward-mode slopes only: with finite differences (three positions per grid time) the LDS form measured faster
    if (sampler != S_ZIGZAG || fd || gp.K > KREG) return hipErrorNotSupported;
    switch (target) {
    case T_GAUSS_ISO: return by_lanes<T_GAUSS_ISO>(gp.G, gp, stream);
    case T_GAUSS_DIAG: return by_lanes<T_GAUSS_DIAG>(gp.G, gp, stream);
    case T_BANANA: return by_lanes<T_BANANA>(gp.G, gp, stream);
    default: return hipErrorNotSupported;
    }
}

}  // namespace pdmp
