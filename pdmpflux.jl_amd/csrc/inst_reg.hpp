// Register-form instantiations of the group kernel (x, v in VGPRs; GroupChain<..., KC>): up to KREG slots per lane, lanes
// per chain G fixed at compile time, and a second instantiation for K == KREG exactly (d = 50, G = 4; d = 100, G = 8; ...)
// where that one stays free of spills.  One translation unit per sampler (inst_reg_*.hip) so they compile in parallel.
#pragma once
#include "group_kernel.hpp"

PDMP_NS_BEGIN

// TAPE: the tape-replay instantiations (inst_tape_reg_*.hip; group_kernel_reg_tape, one wave per SIMD).  A replay runs the
// 13-slot kernels (with the per-slot skips where K < 13) for every configuration the register form serves with the Philox stream.
template <int S, int T, bool FD, bool TAPE = false>
static hipError_t reg_by_lanes(const GroupParams &gp, hipStream_t stream) {
    if constexpr (!RegForm<S, T, FD>::kAny) return hipErrorNotSupported;
    else {
        switch (gp.G) {
        case 1: return launch_group_impl<S, T, FD, KREG, 1, false, TAPE>(gp, stream);
        case 2: return launch_group_impl<S, T, FD, KREG, 2, false, TAPE>(gp, stream);
        case 4: return launch_group_impl<S, T, FD, KREG, 4, false, TAPE>(gp, stream);
        case 8: return launch_group_impl<S, T, FD, KREG, 8, false, TAPE>(gp, stream);
        case 16: return launch_group_impl<S, T, FD, KREG, 16, false, TAPE>(gp, stream);
        // (round 4, last: Zig-Zag at d = 209 .. 832 -- half a wave or a wave per chain; the other samplers keep the LDS form there)
        case 32: if constexpr (S == S_ZIGZAG && !FD && !TAPE) return launch_group_impl<S, T, FD, KREG, 32, false, TAPE>(gp, stream); else return hipErrorNotSupported;
        // (a whole wave per chain: the exact-slot-count kernels only, K = 7 .. 12 -- with thirteen slots the diagonal Gaussian and the banana spill 2 - 18 registers)
        default: return hipErrorNotSupported;
        }
    }
}

template <int S, bool TAPE = false>
static hipError_t reg_by_target(int target, bool fd, const GroupParams &gp, hipStream_t stream) {
    if (gp.K > KREG || (!TAPE && gp.G >= 2 && gp.K < (KREG + 1) / 2)) return hipErrorNotSupported;   // (the exact-K kernels or the LDS form serve them)
    switch (target) {
    case T_GAUSS_ISO: return fd ? reg_by_lanes<S, T_GAUSS_ISO, true, TAPE>(gp, stream) : reg_by_lanes<S, T_GAUSS_ISO, false, TAPE>(gp, stream);
    case T_GAUSS_DIAG: return fd ? reg_by_lanes<S, T_GAUSS_DIAG, true, TAPE>(gp, stream) : reg_by_lanes<S, T_GAUSS_DIAG, false, TAPE>(gp, stream);
    case T_BANANA: return fd ? reg_by_lanes<S, T_BANANA, true, TAPE>(gp, stream) : reg_by_lanes<S, T_BANANA, false, TAPE>(gp, stream);
    default: return hipErrorNotSupported;
    }
}

// Exact-K kernels (inst_reg_*_k.hip): the FULL variant for every K = (KREG+1)/2 .. KREG-1 that the lanes-per-chain rule can
// produce with G >= 2, and for K = 2 .. KREG-1 with one lane per chain (d <= 12: the reference's demo regime, 64 chains per wave) (Zig-Zag, BPS: forward-mode slopes; Boomerang: both slope modes) -- no per-slot skips, so the slot bodies interleave (38-41 % faster than the
// partial variant of KREG slots where both were measured).
template <int S, int T, bool FD, int KC>
static hipError_t exact_by_lanes(const GroupParams &gp, hipStream_t stream) {
    switch (gp.G) {
    case 1: return launch_group_impl<S, T, FD, KC, 1, true>(gp, stream);
    case 2: return launch_group_impl<S, T, FD, KC, 2, true>(gp, stream);
    case 4: return launch_group_impl<S, T, FD, KC, 4, true>(gp, stream);
    case 8: return launch_group_impl<S, T, FD, KC, 8, true>(gp, stream);
    case 16: return launch_group_impl<S, T, FD, KC, 16, true>(gp, stream);
    case 32: if constexpr (S == S_ZIGZAG && !FD && KC >= 7) return launch_group_impl<S, T, FD, KC, 32, true>(gp, stream); else return hipErrorNotSupported;
    case 64: if constexpr (S == S_ZIGZAG && !FD && KC >= 7) return launch_group_impl<S, T, FD, KC, 64, true>(gp, stream); else return hipErrorNotSupported;
    default: return hipErrorNotSupported;
    }
}
// HALF splits the instantiations over translation units per (sampler, slope mode): 0 = K <= 8, 1 = K >= 9 (build time);
// 2 = K = 2..6 with G >= 2 lanes per chain -- the layouts the occupancy rule picks when a launch has too few chains to fill the
// machine at the default G (pdmp_suggest_group_lanes: d = 50 on a 12 500-chain shard runs G = 16, K = 4)
template <int S, int T, bool FD, int HALF>
static hipError_t exact_by_slots(const GroupParams &gp, hipStream_t stream) {
    static_assert(KREG == 13, "one case per K below");
    if constexpr (HALF == 2) {
        if (gp.G < 2) return hipErrorNotSupported;
        if constexpr (S == S_ZIGZAG && !FD) {     // d <= 4 in the lane kernel's layout (G = 4 by default there since round 4): small ensembles
            if (gp.K == 1) return gp.G == 4 ? launch_group_impl<S, T, FD, 2, 4, false>(gp, stream) : hipErrorNotSupported;   // (two slots, the second skipped: GroupChain needs KC >= 2)
        }
        switch (gp.K) {
        case 2: return exact_by_lanes<S, T, FD, 2>(gp, stream);
        case 3: return exact_by_lanes<S, T, FD, 3>(gp, stream);
        case 4: return exact_by_lanes<S, T, FD, 4>(gp, stream);
        case 5: return exact_by_lanes<S, T, FD, 5>(gp, stream);
        case 6: return exact_by_lanes<S, T, FD, 6>(gp, stream);
        default: return hipErrorNotSupported;
        }
    } else
    if (gp.K < 7 && gp.G != 1) return hipErrorNotSupported;     // (K < 7 with G >= 2: HALF 2, where that translation unit exists)
    if constexpr (HALF == 0) {
        switch (gp.K) {
        case 2: return launch_group_impl<S, T, FD, 2, 1, true>(gp, stream);
        case 3: return launch_group_impl<S, T, FD, 3, 1, true>(gp, stream);
        case 4: return launch_group_impl<S, T, FD, 4, 1, true>(gp, stream);
        case 5: return launch_group_impl<S, T, FD, 5, 1, true>(gp, stream);
        case 6: return launch_group_impl<S, T, FD, 6, 1, true>(gp, stream);
        case 7: return exact_by_lanes<S, T, FD, 7>(gp, stream);
        case 8: return exact_by_lanes<S, T, FD, 8>(gp, stream);
        default: return hipErrorNotSupported;
        }
    } else {
        switch (gp.K) {
        case 9: return exact_by_lanes<S, T, FD, 9>(gp, stream);
        case 10: return exact_by_lanes<S, T, FD, 10>(gp, stream);
        case 11: return exact_by_lanes<S, T, FD, 11>(gp, stream);
        case 12: return exact_by_lanes<S, T, FD, 12>(gp, stream);
        default: return hipErrorNotSupported;
        }
    }
}
template <int S, bool FD, int HALF>       // one translation unit per (sampler, slope mode, half)
static hipError_t exact_by_target(int target, const GroupParams &gp, hipStream_t stream) {
    switch (target) {
    case T_GAUSS_ISO: return exact_by_slots<S, T_GAUSS_ISO, FD, HALF>(gp, stream);
    case T_GAUSS_DIAG: return exact_by_slots<S, T_GAUSS_DIAG, FD, HALF>(gp, stream);
    case T_BANANA: return exact_by_slots<S, T_BANANA, FD, HALF>(gp, stream);
    default: return hipErrorNotSupported;
    }
}

PDMP_NS_END
