// Lane-per-chain kernels of the Zig-Zag hot path (lane_kernel.hpp): isotropic Gaussian and banana.
#include "lane_kernel.hpp"

namespace pdmp {
hipError_t launch_lane_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    switch (target) {
    case T_GAUSS_ISO: return launch_lane<T_GAUSS_ISO>(gp, flag, rounds, stream);
    case T_BANANA: return launch_lane<T_BANANA>(gp, flag, rounds, stream);
    default: return hipErrorNotSupported;
    }
}
hipError_t launch_lane_check(const double *vinit, int64_t n, int32_t *flag, hipStream_t stream) {
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(lane_check_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, vinit, n, flag);
    return hipGetLastError();
}
}  // namespace pdmp
