// bm_probe.hip -- throughput of the normal-pair generators of wave_math.hpp under the occupancy of the register-form group kernels
// (two waves per SIMD), per variant:  hipcc -DVARIANT=n ...   prints ns per pair-lane-iteration and pairs/s.
#include "../../pdmpflux.jl_amd/csrc/wave_math.hpp"
#include <cstdio>
#include <cstdlib>
using namespace pdmp;

#ifndef PROBE_RV
#define PROBE_RV 2
#endif
#ifndef UNROLL
#define UNROLL 1
#endif

__global__ __launch_bounds__(256, 2) void probe(uint64_t seed, int iters, double *out) {
    ChainRngT<PROBE_RV> rng;
    rng.init_lane(seed, (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x));
    double acc = 0.0;
    for (int i = 0; i < iters; i += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            double z0, z1;
            rng.normal_pair(i + u, z0, z1);
            acc += z0 * z1;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 2;                       // 2 blocks x 4 waves per CU = 2 waves per SIMD
    double *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, 1ull, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, 1ull, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double pairs = (double)blocks * 256 * iters;
    double h[4];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("RV=%d UNROLL=%d %s: %.3f ms, %.3e pairs/s, %.1f SIMD-cycles per wave-pair at 2.4 GHz (2 waves/SIMD -> x2 per wave)  [%g]\n", PROBE_RV, UNROLL,
#ifdef PDMP_INLINE_MATH
           "inline",
#else
           "call",
#endif
           ms, pairs / (ms * 1e-3), ms * 1e-3 * 2.4e9 / ((double)iters * 2), h[0]);
    return 0;
}
