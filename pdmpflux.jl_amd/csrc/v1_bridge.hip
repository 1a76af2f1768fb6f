// v1_bridge.hip -- compiled with -DPDMP_RNG_VERSION=1 (csrc/Makefile: v1/v1_bridge.o): the entry points through which the engine
// (compiled in the default arithmetic contract, namespace pdmp::av2) reaches the kernels of contract v1 (namespace pdmp::av1): the
// LDS-form group kernels of every sampler and the wave-per-chain kernel, built from the same sources as their v2 twins (v1/inst_*.o).
#include "group_kernel.hpp"
#if PDMP_RNG_VERSION != 1
#error "v1_bridge.hip is the bridge into the v1 build: compile it with -DPDMP_RNG_VERSION=1"
#endif

PDMP_NS_BEGIN
typedef hipError_t (*glaunch_fn)(const GroupParams &, hipStream_t);
typedef hipError_t (*launch_fn)(const SkelParams &, hipStream_t);
PDMP_NS_END

namespace pdmp {
hipError_t v1_launch_group(int sampler, int target, const void *group_params, hipStream_t stream) {
    static const glaunch_fn table[7][4] = {
        {launch_group<S_ZIGZAG, T_GAUSS_ISO>, launch_group<S_ZIGZAG, T_GAUSS_DIAG>, launch_group<S_ZIGZAG, T_BANANA>, launch_group<S_ZIGZAG, T_GAUSS_DENSE>},
        {launch_group<S_BPS, T_GAUSS_ISO>, launch_group<S_BPS, T_GAUSS_DIAG>, launch_group<S_BPS, T_BANANA>, launch_group<S_BPS, T_GAUSS_DENSE>},
        {launch_group<S_BOOMERANG, T_GAUSS_ISO>, launch_group<S_BOOMERANG, T_GAUSS_DIAG>, launch_group<S_BOOMERANG, T_BANANA>, launch_group<S_BOOMERANG, T_GAUSS_DENSE>},
        {launch_group<S_FECMC, T_GAUSS_ISO>, launch_group<S_FECMC, T_GAUSS_DIAG>, launch_group<S_FECMC, T_BANANA>, nullptr},
        {launch_group<S_SUZZ, T_GAUSS_ISO>, launch_group<S_SUZZ, T_GAUSS_DIAG>, launch_group<S_SUZZ, T_BANANA>, nullptr},
        {launch_group<S_RHMC, T_GAUSS_ISO>, launch_group<S_RHMC, T_GAUSS_DIAG>, launch_group<S_RHMC, T_BANANA>, nullptr},
        {launch_group<S_STICKY, T_GAUSS_ISO>, launch_group<S_STICKY, T_GAUSS_DIAG>, launch_group<S_STICKY, T_BANANA>, nullptr},
    };
    if (sampler < 0 || sampler > 6 || target < 0 || target > 3 || !table[sampler][target]) return hipErrorNotSupported;
    return table[sampler][target](*static_cast<const GroupParams *>(group_params), stream);
}
hipError_t v1_launch_skeleton(int sampler, int target, const void *skel_params, hipStream_t stream) {
    static const launch_fn table[3][3] = {
        {launch_skeleton<S_ZIGZAG, T_GAUSS_ISO>, launch_skeleton<S_ZIGZAG, T_GAUSS_DIAG>, launch_skeleton<S_ZIGZAG, T_BANANA>},
        {launch_skeleton<S_BPS, T_GAUSS_ISO>, launch_skeleton<S_BPS, T_GAUSS_DIAG>, launch_skeleton<S_BPS, T_BANANA>},
        {launch_skeleton<S_BOOMERANG, T_GAUSS_ISO>, launch_skeleton<S_BOOMERANG, T_GAUSS_DIAG>, launch_skeleton<S_BOOMERANG, T_BANANA>},
    };
    if (sampler < 0 || sampler > 2 || target < 0 || target > 2) return hipErrorNotSupported;
    return table[sampler][target](*static_cast<const SkelParams *>(skel_params), stream);
}
}  // namespace pdmp
