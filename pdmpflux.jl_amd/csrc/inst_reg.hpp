// Register-form instantiations of the group kernel (x, v in VGPRs; GroupChain<..., KC>): up to KREG slots per lane, lanes
// per chain G fixed at compile time, and a second instantiation for K == KREG exactly (d = 50, G = 4; d = 100, G = 8; ...)
// where that one stays free of spills.  One translation unit per sampler (inst_reg_*.hip) so they compile in parallel.
#pragma once
#include "group_kernel.hpp"

namespace pdmp {

template <int S, int T, bool FD>
static hipError_t reg_by_lanes(const GroupParams &gp, hipStream_t stream) {
    if constexpr (!RegForm<S, T, FD>::kAny) return hipErrorNotSupported;
    else {
        switch (gp.G) {
        case 1: return launch_group_impl<S, T, FD, KREG, 1>(gp, stream);
        case 2: return launch_group_impl<S, T, FD, KREG, 2>(gp, stream);
        case 4: return launch_group_impl<S, T, FD, KREG, 4>(gp, stream);
        case 8: return launch_group_impl<S, T, FD, KREG, 8>(gp, stream);
        case 16: return launch_group_impl<S, T, FD, KREG, 16>(gp, stream);
        default: return hipErrorNotSupported;
        }
    }
}

template <int S>
static hipError_t reg_by_target(int target, bool fd, const GroupParams &gp, hipStream_t stream) {
    if (gp.K > KREG || (gp.G >= 2 && gp.K < (KREG + 1) / 2)) return hipErrorNotSupported;   // (forced lane counts only: the LDS form serves them)
    switch (target) {
    case T_GAUSS_ISO: return fd ? reg_by_lanes<S, T_GAUSS_ISO, true>(gp, stream) : reg_by_lanes<S, T_GAUSS_ISO, false>(gp, stream);
    case T_GAUSS_DIAG: return fd ? reg_by_lanes<S, T_GAUSS_DIAG, true>(gp, stream) : reg_by_lanes<S, T_GAUSS_DIAG, false>(gp, stream);
    case T_BANANA: return fd ? reg_by_lanes<S, T_BANANA, true>(gp, stream) : reg_by_lanes<S, T_BANANA, false>(gp, stream);
    default: return hipErrorNotSupported;
    }
}

// Exact-K kernels (inst_reg_*_k.hip): the FULL variant for every K = (KREG+1)/2 .. KREG-1 that the lanes-per-chain rule can
// produce with G >= 2, forward-mode slopes -- no per-slot skips, so the slot bodies interleave (38-41 % faster than the
// partial variant of KREG slots where both were measured).
template <int S, int T, int KC>
static hipError_t exact_by_lanes(const GroupParams &gp, hipStream_t stream) {
    switch (gp.G) {
    case 2: return launch_group_impl<S, T, false, KC, 2, true>(gp, stream);
    case 4: return launch_group_impl<S, T, false, KC, 4, true>(gp, stream);
    case 8: return launch_group_impl<S, T, false, KC, 8, true>(gp, stream);
    case 16: return launch_group_impl<S, T, false, KC, 16, true>(gp, stream);
    default: return hipErrorNotSupported;
    }
}
template <int S, int T>
static hipError_t exact_by_slots(const GroupParams &gp, hipStream_t stream) {
    static_assert(KREG == 13, "one case per K below");
    switch (gp.K) {
    case 7: return exact_by_lanes<S, T, 7>(gp, stream);
    case 8: return exact_by_lanes<S, T, 8>(gp, stream);
    case 9: return exact_by_lanes<S, T, 9>(gp, stream);
    case 10: return exact_by_lanes<S, T, 10>(gp, stream);
    case 11: return exact_by_lanes<S, T, 11>(gp, stream);
    case 12: return exact_by_lanes<S, T, 12>(gp, stream);
    default: return hipErrorNotSupported;
    }
}
template <int S>
static hipError_t exact_by_target(int target, bool fd, const GroupParams &gp, hipStream_t stream) {
    if (fd) return hipErrorNotSupported;
    switch (target) {
    case T_GAUSS_ISO: return exact_by_slots<S, T_GAUSS_ISO>(gp, stream);
    case T_GAUSS_DIAG: return exact_by_slots<S, T_GAUSS_DIAG>(gp, stream);
    case T_BANANA: return exact_by_slots<S, T_BANANA>(gp, stream);
    default: return hipErrorNotSupported;
    }
}

}  // namespace pdmp
