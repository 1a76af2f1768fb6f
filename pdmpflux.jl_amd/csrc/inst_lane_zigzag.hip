// Lane-per-chain kernels of the Zig-Zag hot path (lane_kernel.hpp): isotropic Gaussian and banana, d = 49 .. 52 (13 slots of four).
#include "lane_kernel.hpp"

PDMP_NS_BEGIN
// flag[0] != 0 after this kernel: some |v_i| != 1 -- the lane kernel leaves the launch to the group kernel (GroupParams::only_if)
__global__ __launch_bounds__(256) void lane_check_kernel(const double *vinit, int64_t n, int32_t *flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool bad = i < n && fabs(vinit[i]) != 1.0;
    if (__ballot(bad) != 0ull && lane_id() == 0) atomicOr(flag, 1);
}

hipError_t launch_lane_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    return launch_lane_range<13, 13>(target, gp, flag, rounds, stream);
}
hipError_t launch_lane_check(const double *vinit, int64_t n, int32_t *flag, hipStream_t stream) {
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(lane_check_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, vinit, n, flag);
    return hipGetLastError();
}
PDMP_NS_END
