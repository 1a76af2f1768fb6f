// user_jit.hip -- potentials given as SOURCE: the engine's counterpart of the `U::Function` argument of the reference's
// *AD constructors (ZigZagSamplers.jl:118-126, BouncyParticleSamplers.jl:86-94, BoomerangSamplers.jl:79-87) and of
// create_gradient_function (ADBackend.jl:30-142).  A Julia closure cannot cross a C ABI; a few lines of HIP can:
//
//   kind 0:  template <class X> __device__ auto U(const X &x, int d)            -- scalar potential, x[j] = coordinate j
//   kind 1:  template <class X> __device__ auto grad_i(int i, const X &x, int d) -- coordinate i of grad-U
//
// written with + - * /, sq, sqrt_, log_, sin_, cos_ (user_ad.hpp) so that the same text runs on doubles, on duals in time
// (the slope of the rate along the flow) and, for kind 0, on the nested forward-mode numbers that produce grad-U itself.
// The text is compiled by hiprtc against the very headers this library was built from (embedded at build time) into
// group_kernel_body<SAMPLER, T_USER, FD>, loaded with the module API and launched like the built-in instantiations.
#include "../../include/pdmp_hip.h"
#include "group_kernel.hpp"

#include <hip/hiprtc.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "embedded_headers.inc"

using namespace pdmp;

namespace {
struct Compiled { hipModule_t module; hipFunction_t fn; };
std::mutex g_mu;
std::map<std::tuple<int, int, int, int, std::string>, Compiled> g_cache;   // (device, sampler, fd, kind, source)

void put(char *err, int errlen, const std::string &s) {
    if (err && errlen > 0) { snprintf(err, (size_t)errlen, "%s", s.c_str()); }
}
}  // namespace

// 0 ok; -1 the source does not compile (log in err); -3 runtime failure
extern "C" int pdmp_user_compile(int device, int sampler, int fd, int kind, int rng_version, int rtgrid, const char *source, void **fn_out, char *err, int errlen) {   // rtgrid: grid_size other than 10 / 0 (GroupChain: RTGRID)
    std::lock_guard<std::mutex> lock(g_mu);
    const auto key = std::make_tuple(device, sampler, fd, kind + 16 * rng_version + 64 * (rtgrid ? 1 : 0), std::string(source));   // (the arithmetic contract is part of the kernel)
    auto it = g_cache.find(key);
    if (it != g_cache.end()) { *fn_out = (void *)it->second.fn; return 0; }
    // The kernel may not use scratch: a register spilled inside a lane-dependent branch comes back wrong in the lanes that were
    // inactive there (GroupOcc in group_kernel.hpp).  The static instantiations are checked at build time; here the register
    // budget is doubled (2 waves per SIMD, then 1) until the compiled kernel is scratch-free, else the potential is refused.
    // (A potential that indexes a private array dynamically needs scratch as well and is refused with the same message.)
    Compiled c{};
    bool ok = false;
    for (int waves = 2; waves >= 1 && !ok; --waves) {
        std::string prog;
        prog += "#define PDMP_RNG_VERSION " + std::to_string(rng_version == 1 ? 1 : 2) + "\n";
        prog += "#define PDMP_USER_KIND " + std::to_string(kind) + "\n";
        prog += "#include \"wave_math.hpp\"\n#include \"user_ad.hpp\"\n";
        prog += "namespace pdmp_user {\nusing namespace pdmp;\n#line 1 \"potential\"\n";
        prog += source;
        prog += "\n}\n#include \"group_kernel.hpp\"\n";
        prog += "extern \"C\" __global__ __launch_bounds__(256, " + std::to_string(waves) + ") void pdmp_user_kernel(const pdmp::GroupParams gp) {\n";
        prog += "    if (pdmp::served_elsewhere(gp)) return;\n";
        prog += "    pdmp::group_kernel_body<" + std::to_string(sampler) + ", pdmp::T_USER, " + (fd ? "true" : "false") + ", 0, 0, false, false, " + "true>(gp);   // (grid_size at run time, as the built-in LDS-form kernels)\n}\n";
        hiprtcProgram p;
        if (hiprtcCreateProgram(&p, prog.c_str(), "pdmp_user.hip", k_hdr_count, (const char **)k_hdr_src, (const char **)k_hdr_names) != HIPRTC_SUCCESS) {
            put(err, errlen, "hiprtcCreateProgram failed");
            return -3;
        }
        hipDeviceProp_t prop;
        std::string arch = "--offload-arch=gfx950";
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.gcnArchName[0]) {
            std::string a(prop.gcnArchName);
            arch = "--offload-arch=" + a.substr(0, a.find(':'));
        }
        const char *opts[] = {arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-w"};   // the arithmetic contract of the static build; -w: the log is for errors
        const hiprtcResult r = hiprtcCompileProgram(p, 5, opts);
        if (r != HIPRTC_SUCCESS) {
            size_t n = 0;
            hiprtcGetProgramLogSize(p, &n);
            std::string log(n, '\0');
            if (n) hiprtcGetProgramLog(p, &log[0]);
            hiprtcDestroyProgram(&p);
            put(err, errlen, log);
            return -1;
        }
        size_t cs = 0;
        hiprtcGetCodeSize(p, &cs);
        std::vector<char> code(cs);
        hiprtcGetCode(p, code.data());
        hiprtcDestroyProgram(&p);
        hipError_t e = hipModuleLoadData(&c.module, code.data());
        if (e == hipSuccess) e = hipModuleGetFunction(&c.fn, c.module, "pdmp_user_kernel");
        if (e != hipSuccess) { put(err, errlen, std::string("loading the compiled potential: ") + hipGetErrorString(e)); return -3; }
        int scratch = 0;
        e = hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, c.fn);
        if (e != hipSuccess) { put(err, errlen, std::string("querying the compiled potential: ") + hipGetErrorString(e)); return -3; }
        ok = scratch == 0;
        if (!ok) (void)hipModuleUnload(c.module);
    }
    if (!ok) {
        put(err, errlen, "the compiled potential needs more registers than a wave has (it would spill; see DESIGN.md 4.5): simplify U or lower dim");
        return -3;
    }
    g_cache[key] = c;
    *fn_out = (void *)c.fn;
    return 0;
}

// launch_group_impl (group_kernel.hpp) for a module function; blocks sized to the default 64 KiB dynamic-LDS limit
extern "C" int pdmp_user_launch(void *fn_, const GroupParams *gp_in, void *stream) {
    hipFunction_t fn = (hipFunction_t)fn_;
    GroupParams gp = *gp_in;
    const int cpw = WAVE / gp.G;
    const size_t per_wave = (size_t)gp.tiles * gp.K * WAVE * sizeof(double);
    int wpb = (int)((64 * 1024 - (size_t)gp.tpar * sizeof(double)) / per_wave);
    if (wpb > 4) wpb = 4;
    if (wpb < 1) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)gp.tpar * sizeof(double) + (size_t)wpb * per_wave;
    int per_cu = 0, dev = 0, cus = 0;
    hipError_t e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, WAVE * wpb, lds);
    if (e != hipSuccess) return (int)e;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (per_cu < 1) per_cu = 1;
    if (cus < 1) cus = 256;
    gp.nb = (gp.s.C + cpw - 1) / cpw;
    const int64_t slots = (int64_t)cus * per_cu * wpb;
    group_schedule(gp.nb, slots, gp.s.n_sk, gp.nseg, gp.nseg, gp.seg_len);
    int64_t blocks = (gp.nb + wpb - 1) / wpb;
    if (blocks > (int64_t)cus * per_cu) blocks = (int64_t)cus * per_cu;
    void *args[] = {&gp};
    e = hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, WAVE * wpb, 1, 1, (unsigned)lds, (hipStream_t)stream, args, nullptr);
    return (int)e;
}
