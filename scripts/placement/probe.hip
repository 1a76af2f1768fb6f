// Placement probe (round 4, DESIGN.md section 6 "placement lottery"): why does the config-2 launch run 14.4 ms on some allocations of
// its output arrays and 18 ms on others?  One process, no torch: the output arrays are built from PHYSICAL chunks
// (hipMemCreate / hipMemMap) so that the same physical memory can be re-mapped in another order, chunk sizes can be chosen, and
// a synthetic writer with the lane kernel's access pattern (many slowly advancing 64-byte-line streams, two arrays at once) can
// be pointed at parts of the arrays.  Prints one JSON object per measurement.
//
//   probe malloc  R            R fresh hipMalloc allocations, engine launch timed on each
//   probe vmm     S_MiB R P    R fresh pools of S_MiB-chunks; per pool: identity mapping, then P permutations of the same chunks
//   probe synth   S_MiB R      as vmm, but timing the synthetic writer (+ engine) and the writer on chunk subsets
//   probe frag    H_MiB        free VRAM turned into H-MiB holes, then plain hipMalloc of the arrays out of the holes
//
// Build: make -C scripts/placement      Run on the GPU box: scripts/placement/probe vmm 2048 4 4
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../include/pdmp_hip.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

static const int64_t C = 100000, D = 50, NSK = 1000;
static const size_t XB = size_t(C) * NSK * D * 8, TB = size_t(C) * NSK * 8;

// ---------------------------------------------------------------- synthetic writer
// The lane kernel's store pattern without its arithmetic: a wave owns 64 consecutive chains; per step it writes ONE 64-byte line of
// each chain's X stream and of its V stream (eight lanes per line, eight chains per store instruction, 16-byte stores by pairs would
// be closer but 8-byte pieces merge in the L2 the same way); the streams of a chain are `stride` doubles apart.
__global__ __launch_bounds__(64) void synth_writer(double *X, double *V, int64_t stride, int64_t n_chains, int lines, int *queue,
                                                   int n_batches, int arrays, int nt)
{
    __shared__ int s_b;
    const int lane = threadIdx.x;
    for (;;) {
        if (lane == 0) s_b = atomicAdd(queue, 1);
        __syncthreads();
        const int b = s_b;
        __syncthreads();
        if (b >= n_batches) return;
        for (int l = 0; l < lines; ++l) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int64_t c = int64_t(b) * 64 + 8 * r + (lane >> 3);
                if (c < n_chains) {
                    const int64_t off = c * stride + int64_t(l) * 8 + (lane & 7);
                    const double val = double(l + lane);
                    if (nt) {
                        __builtin_nontemporal_store(val, X + off);
                        if (arrays > 1) __builtin_nontemporal_store(-val, V + off);
                    } else {
                        X[off] = val;
                        if (arrays > 1) V[off] = -val;
                    }
                }
            }
        }
    }
}

__global__ void fill_kernel(double *p, size_t n, double val)
{
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = val;
}

__global__ void probe_read_kernel(const double *p, size_t n, double *out)
{
    if (threadIdx.x == 0) *out = p[n / 2];
}

static int *g_queue = nullptr;
static float run_synth(double *X, double *V, int64_t stride, int64_t chains, int lines, int waves, int arrays, int nt, int reps = 3)
{
    if (!g_queue) HIPCHECK(hipMalloc(&g_queue, 4));
    hipEvent_t a, b;
    HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b));
    float best = 1e30f;
    const int nb = int((chains + 63) / 64);
    for (int i = 0; i < reps; ++i) {
        HIPCHECK(hipMemsetAsync(g_queue, 0, 4, 0));
        HIPCHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(synth_writer, dim3(waves), dim3(64), 0, 0, X, V, stride, chains, lines, g_queue, nb, arrays, nt);
        HIPCHECK(hipEventRecord(b, 0));
        HIPCHECK(hipEventSynchronize(b));
        float ms; HIPCHECK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    HIPCHECK(hipEventDestroy(a)); HIPCHECK(hipEventDestroy(b));
    return best;
}

// ---------------------------------------------------------------- engine launch
static pdmp_handle g_h = nullptr;
static double *g_x0 = nullptr, *g_v0 = nullptr;
static void engine_init()
{
    pdmp_config_t cfg; memset(&cfg, 0, sizeof cfg);
    cfg.sampler = PDMP_ZIGZAG; cfg.dim = int(D); cfg.grid_size = 10; cfg.vectorized_bound = 1; cfg.signed_bound = 1; cfg.adaptive = 1;
    cfg.deriv_mode = PDMP_DERIV_FORWARD; cfg.target_id = PDMP_T_GAUSS_ISO; cfg.device = -1; cfg.tmax = 2.0;
    cfg.fe_mix_p = 0.5; cfg.fe_switch = 1; cfg.fe_positive = 1; cfg.rh_step_size = 0.05; cfg.rh_phi = 1.5707963267948966; cfg.fe_speed_factor = 1.0;
    if (pdmp_create(&cfg, &g_h) != PDMP_OK) { fprintf(stderr, "pdmp_create: %s\n", pdmp_last_error()); exit(2); }
    std::vector<double> x(C * D), v(C * D);
    std::mt19937_64 rng(1); std::normal_distribution<double> nd;
    for (auto &e : x) e = nd(rng);
    for (auto &e : v) e = (rng() & 1) ? 1.0 : -1.0;
    HIPCHECK(hipMalloc(&g_x0, C * D * 8)); HIPCHECK(hipMalloc(&g_v0, C * D * 8));
    HIPCHECK(hipMemcpy(g_x0, x.data(), C * D * 8, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(g_v0, v.data(), C * D * 8, hipMemcpyHostToDevice));
}
// minimum and the individual times of `reps` launches after one untimed launch
static float run_engine(double *X, double *V, double *t, std::string *all = nullptr, int reps = 3)
{
    hipEvent_t a, b;
    HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int i = 0; i <= reps; ++i) {
        HIPCHECK(hipEventRecord(a, 0));
        if (pdmp_sample_skeleton_device(g_h, C, NSK, 0, g_x0, g_v0, 2024 + i, X, V, t, nullptr, nullptr) != PDMP_OK) {
            fprintf(stderr, "launch: %s\n", pdmp_last_error()); exit(2);
        }
        HIPCHECK(hipEventRecord(b, 0));
        HIPCHECK(hipEventSynchronize(b));
        float ms; HIPCHECK(hipEventElapsedTime(&ms, a, b));
        if (i == 0) continue;
        best = std::min(best, ms);
        if (all) { char buf[32]; snprintf(buf, sizeof buf, "%s%.2f", all->empty() ? "" : ",", ms); *all += buf; }
    }
    if (pdmp_check_status(g_h, nullptr) != PDMP_OK) { fprintf(stderr, "status: %s\n", pdmp_last_error()); exit(2); }
    HIPCHECK(hipEventDestroy(a)); HIPCHECK(hipEventDestroy(b));
    return best;
}

// ---------------------------------------------------------------- physical chunks
// hipMemUnmap + hipMemMap of OTHER memory at an address leaves the compute queues' translation of that address pointing at the old
// memory (mode `stale2`: kernels keep writing the chunk that was mapped there first, even after it has been released; hipMemcpy sees the
// new one).  A hipMalloc + hipFree of 64 MiB in between makes the new mapping take effect (the unmap of an ordinary allocation flushes
// the TLBs).  The modes of calls 1 - 13 ran WITHOUT this (PROBE_NO_FLUSH=1 reproduces them): wherever they re-mapped an address, the
// kernels were still writing the first chunks mapped there.
static bool g_flush_after_map = true;
static void flush_translations()
{
    void *t = nullptr;
    HIPCHECK(hipMalloc(&t, size_t(64) << 20));
    HIPCHECK(hipFree(t));
}
struct Pool {
    size_t S = 0;
    std::vector<hipMemGenericAllocationHandle_t> h;
    hipMemAllocationProp prop;
    void create(size_t chunk, size_t n)
    {
        S = chunk;
        memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        h.resize(n);
        for (size_t i = 0; i < n; ++i) HIPCHECK(hipMemCreate(&h[i], S, &prop, 0));
    }
    void release() { for (auto &x : h) HIPCHECK(hipMemRelease(x)); h.clear(); }
};
struct Mapped {
    char *base = nullptr; size_t n = 0, S = 0;
    void reserve(size_t bytes, size_t chunk) { S = chunk; n = (bytes + S - 1) / S; HIPCHECK(hipMemAddressReserve((void **)&base, n * S, 0, nullptr, 0)); }
    void map(Pool &p, const std::vector<int> &ids)
    {
        for (size_t i = 0; i < n; ++i) HIPCHECK(hipMemMap(base + i * S, S, 0, p.h[ids[i]], 0));
        hipMemAccessDesc d; d.location.type = hipMemLocationTypeDevice; d.location.id = 0; d.flags = hipMemAccessFlagsProtReadWrite;
        HIPCHECK(hipMemSetAccess(base, n * S, &d, 1));
        if (g_flush_after_map) flush_translations();
    }
    void unmap() { for (size_t i = 0; i < n; ++i) HIPCHECK(hipMemUnmap(base + i * S, S)); }
    void free_va() { HIPCHECK(hipMemAddressFree(base, n * S)); base = nullptr; }
};

static void malloc_ctl(const char *name)
{
    double *X, *V, *t;
    HIPCHECK(hipMalloc(&X, XB)); HIPCHECK(hipMalloc(&V, XB)); HIPCHECK(hipMalloc(&t, TB));
    std::string all; const float e = run_engine(X, V, t, &all);
    const float s2 = run_synth(X, V, NSK * D, C, int(NSK * D / 8), 1024, 2, 1);
    printf("{\"probe\":\"malloc_ctl\",\"name\":\"%s\",\"engine_ms\":%.2f,\"engine_all\":[%s],\"synth2_ms\":%.2f,\"x\":\"%p\",\"v\":\"%p\"}\n", name, e, all.c_str(), s2, (void *)X, (void *)V);
    fflush(stdout);
    HIPCHECK(hipFree(X)); HIPCHECK(hipFree(V)); HIPCHECK(hipFree(t));
}

static std::string ids_str(const std::vector<int> &v, size_t lim = 64)
{
    std::string s = "[";
    for (size_t i = 0; i < v.size() && i < lim; ++i) s += (i ? "," : "") + std::to_string(v[i]);
    return s + (v.size() > lim ? ",...]" : "]");
}

int main(int argc, char **argv)
{
    if (getenv("PROBE_NO_FLUSH")) g_flush_after_map = false;
    const std::string mode = argc > 1 ? argv[1] : "malloc";
    HIPCHECK(hipSetDevice(0));
    size_t fr, tot; HIPCHECK(hipMemGetInfo(&fr, &tot));
    printf("{\"probe\":\"start\",\"mode\":\"%s\",\"free_gb\":%.1f,\"total_gb\":%.1f}\n", mode.c_str(), fr / 1e9, tot / 1e9);
    engine_init();
    if (mode == "malloc") {
        const int R = argc > 2 ? atoi(argv[2]) : 6;
        void *spacer = nullptr;
        for (int r = 0; r < R; ++r) {
            double *X, *V, *t;
            HIPCHECK(hipMalloc(&X, XB)); HIPCHECK(hipMalloc(&V, XB)); HIPCHECK(hipMalloc(&t, TB));
            std::string all; float e = run_engine(X, V, t, &all);
            float s2 = run_synth(X, V, NSK * D, C, int(NSK * D / 8), 1024, 2, 1);
            float s1 = run_synth(X, V, NSK * D, C, int(NSK * D / 8), 1024, 1, 1);
            printf("{\"probe\":\"malloc\",\"rep\":%d,\"X\":\"%p\",\"V\":\"%p\",\"engine_ms\":%.2f,\"engine_all\":[%s],\"synth2_ms\":%.2f,\"synth1_ms\":%.2f}\n", r, (void *)X, (void *)V, e, all.c_str(), s2, s1);
            fflush(stdout);
            HIPCHECK(hipFree(X)); HIPCHECK(hipFree(V)); HIPCHECK(hipFree(t));
            if (r % 2 == 1) { if (spacer) HIPCHECK(hipFree(spacer)); HIPCHECK(hipMalloc(&spacer, size_t(1 + r) << 28)); }
        }
        return 0;
    }
    if (mode == "frag") {
        // Fragmentation injection: fill the free VRAM with H-MiB physical chunks, release every other one -- the free memory is now holes
        // of H MiB -- and hipMalloc the output arrays out of those holes.  Does a plain hipMalloc then show the slow mode?
        const size_t H = size_t(argc > 2 ? atol(argv[2]) : 8) << 20;
        const double keep_gb = argc > 3 ? atof(argv[3]) : 6.0;            // left untouched (runtime, handle scratch)
        size_t f2, t2; HIPCHECK(hipMemGetInfo(&f2, &t2));
        const size_t n = size_t((double(f2) - keep_gb * 1e9) / double(H));
        Pool pool; pool.S = H;
        memset(&pool.prop, 0, sizeof pool.prop);
        pool.prop.type = hipMemAllocationTypePinned; pool.prop.location.type = hipMemLocationTypeDevice; pool.prop.location.id = 0;
        pool.h.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            hipMemGenericAllocationHandle_t hh;
            if (hipMemCreate(&hh, H, &pool.prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            pool.h.push_back(hh);
        }
        size_t held = 0;
        for (size_t i = 0; i < pool.h.size(); ++i) if (i & 1) { HIPCHECK(hipMemRelease(pool.h[i])); } else ++held;
        HIPCHECK(hipMemGetInfo(&f2, &t2));
        printf("{\"probe\":\"frag_setup\",\"hole_mib\":%zu,\"chunks\":%zu,\"held\":%zu,\"free_gb_after\":%.1f}\n", H >> 20, pool.h.size(), held, f2 / 1e9);
        fflush(stdout);
        for (int r = 0; r < 2; ++r) {
            double *X, *V, *t;
            HIPCHECK(hipMalloc(&X, XB)); HIPCHECK(hipMalloc(&V, XB)); HIPCHECK(hipMalloc(&t, TB));
            std::string all; float e = run_engine(X, V, t, &all);
            float s2 = run_synth(X, V, NSK * D, C, int(NSK * D / 8), 1024, 2, 1);
            printf("{\"probe\":\"frag\",\"hole_mib\":%zu,\"rep\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s],\"synth2_ms\":%.2f}\n", H >> 20, r, e, all.c_str(), s2);
            fflush(stdout);
            HIPCHECK(hipFree(X)); HIPCHECK(hipFree(V)); HIPCHECK(hipFree(t));
        }
        for (size_t i = 0; i < pool.h.size(); ++i) if (!(i & 1)) HIPCHECK(hipMemRelease(pool.h[i]));
        // and once more after everything was released: does the allocator find large blocks again?
        for (int r = 0; r < 2; ++r) {
            double *X, *V, *t;
            HIPCHECK(hipMalloc(&X, XB)); HIPCHECK(hipMalloc(&V, XB)); HIPCHECK(hipMalloc(&t, TB));
            std::string all; float e = run_engine(X, V, t, &all);
            printf("{\"probe\":\"frag_after_release\",\"hole_mib\":%zu,\"rep\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s]}\n", H >> 20, r, e, all.c_str());
            fflush(stdout);
            HIPCHECK(hipFree(X)); HIPCHECK(hipFree(V)); HIPCHECK(hipFree(t));
        }
        return 0;
    }
    if (mode == "contig") {
        // hipExtMallocWithFlags(hipDeviceMallocContiguous): does the driver hand out physically contiguous arrays of this size, and are they fast?
        const int R = argc > 2 ? atoi(argv[2]) : 3;
        for (int r = 0; r < R; ++r) {
            double *X = nullptr, *V = nullptr, *t = nullptr;
            hipError_t ex = hipExtMallocWithFlags((void **)&X, XB, hipDeviceMallocContiguous);
            hipError_t ev = ex == hipSuccess ? hipExtMallocWithFlags((void **)&V, XB, hipDeviceMallocContiguous) : ex;
            if (ex != hipSuccess || ev != hipSuccess) {
                (void)hipGetLastError();
                printf("{\"probe\":\"contig\",\"rep\":%d,\"error\":\"%s\"}\n", r, hipGetErrorString(ex != hipSuccess ? ex : ev));
                if (X) (void)hipFree(X);
                break;
            }
            HIPCHECK(hipMalloc(&t, TB));
            std::string all; float e = run_engine(X, V, t, &all);
            float s2 = run_synth(X, V, NSK * D, C, int(NSK * D / 8), 1024, 2, 1);
            printf("{\"probe\":\"contig\",\"rep\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s],\"synth2_ms\":%.2f}\n", r, e, all.c_str(), s2);
            fflush(stdout);
            HIPCHECK(hipFree(X)); HIPCHECK(hipFree(V)); HIPCHECK(hipFree(t));
        }
        return 0;
    }
    if (mode == "map") {
        // Is the speed a property of WHERE in the VRAM a chunk lies?  All free memory as 1-GiB chunks; every chunk timed alone with a dense
        // writer (65 536 streams of 16 KiB inside the chunk: the machine is full, the bytes stay inside this chunk); then the engine on arrays
        // built from the fastest chunks, from the slowest, and from the first / last allocated.
        const size_t S = size_t(argc > 2 ? atol(argv[2]) : 1024) << 20;
        size_t f2, t2; HIPCHECK(hipMemGetInfo(&f2, &t2));
        const int n = int((double(f2) - 8e9) / double(S));
        Pool pool; pool.create(S, n);
        Mapped one; one.reserve(S, S);
        std::vector<float> m1(n);
        std::string ms;
        const int64_t st = 2048, nch = int64_t(S / 8 / st);                   // streams of `st` doubles
        for (int i = 0; i < n; ++i) {
            one.map(pool, std::vector<int>{i});
            m1[i] = run_synth((double *)one.base, (double *)one.base, st, nch, int(st / 8), 1024, 1, 1, 3);
            one.unmap();
            char buf[32]; snprintf(buf, sizeof buf, "%s%.4f", i ? "," : "", m1[i]); ms += buf;
        }
        printf("{\"probe\":\"map\",\"chunk_mib\":%zu,\"chunks\":%d,\"chunk_ms\":[%s]}\n", S >> 20, n, ms.c_str());
        fflush(stdout);
        Mapped X, V, T;
        X.reserve(XB, S); V.reserve(XB, S); T.reserve(TB, S);
        const int nx = int(X.n), need = 2 * nx + int(T.n);
        std::vector<int> order(n); for (int i = 0; i < n; ++i) order[i] = i;
        std::vector<int> by = order;
        std::sort(by.begin(), by.end(), [&](int a, int b) { return m1[a] < m1[b]; });
        auto run_sel = [&](const char *name, const std::vector<int> &sel) {
            std::vector<int> ix(sel.begin(), sel.begin() + nx), iv(sel.begin() + nx, sel.begin() + 2 * nx), it(sel.begin() + 2 * nx, sel.begin() + need);
            X.map(pool, ix); V.map(pool, iv); T.map(pool, it);
            std::string all; float e = run_engine((double *)X.base, (double *)V.base, (double *)T.base, &all);
            float s2 = run_synth((double *)X.base, (double *)V.base, NSK * D, C, int(NSK * D / 8), 1024, 2, 1);
            double mean = 0; for (int i = 0; i < need; ++i) mean += m1[sel[i]]; mean /= need;
            printf("{\"probe\":\"map_select\",\"selection\":\"%s\",\"mean_chunk_ms\":%.4f,\"engine_ms\":%.2f,\"engine_all\":[%s],\"synth2_ms\":%.2f}\n", name, mean, e, all.c_str(), s2);
            fflush(stdout);
            X.unmap(); V.unmap(); T.unmap();
        };
        if (n >= need) {
            run_sel("fastest", std::vector<int>(by.begin(), by.begin() + need));
            run_sel("slowest", std::vector<int>(by.end() - need, by.end()));
            run_sel("first_allocated", std::vector<int>(order.begin(), order.begin() + need));
            run_sel("last_allocated", std::vector<int>(order.end() - need, order.end()));
            if (n >= 2 * need) run_sel("middle_allocated", std::vector<int>(order.begin() + need, order.begin() + 2 * need));
            run_sel("fastest_again", std::vector<int>(by.begin(), by.begin() + need));
        }
        pool.release();
        return 0;
    }
    if (mode == "sets") {
        // What bench.py does: several sets of arrays alive at once (1-GiB chunks, allocated one set after the other); the engine on each;
        // per (X chunk, V chunk) pair of each set the dense two-array writer.
        const size_t S = size_t(argc > 2 ? atol(argv[2]) : 1024) << 20;
        const int nsets = argc > 3 ? atoi(argv[3]) : 3;
        std::vector<Pool> pools(nsets);
        std::vector<Mapped> Xs(nsets), Vs(nsets), Ts(nsets);
        for (int k = 0; k < nsets; ++k) {
            Xs[k].reserve(XB, S); Vs[k].reserve(XB, S); Ts[k].reserve(TB, S);
            const int nx = int(Xs[k].n), nt_ = int(Ts[k].n);
            pools[k].create(S, 2 * nx + nt_);
            std::vector<int> ix(nx), iv(nx), it(nt_);
            for (int i = 0; i < nx; ++i) { ix[i] = i; iv[i] = nx + i; }
            for (int i = 0; i < nt_; ++i) it[i] = 2 * nx + i;
            Xs[k].map(pools[k], ix); Vs[k].map(pools[k], iv); Ts[k].map(pools[k], it);
        }
        const int64_t st = 2048, nch = int64_t(S / 8 / st);
        for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < nsets; ++k) {
            std::string all; float e = run_engine((double *)Xs[k].base, (double *)Vs[k].base, (double *)Ts[k].base, &all);
            std::string pers;
            const int nx = int(Xs[k].n);
            for (int i = 0; i < nx && rep == 0; ++i) {
                const float m = run_synth((double *)(Xs[k].base + size_t(i) * S), (double *)(Vs[k].base + size_t(i) * S), st, nch, int(st / 8), 1024, 2, 1, 2);
                char buf[32]; snprintf(buf, sizeof buf, "%s%.4f", i ? "," : "", m); pers += buf;
            }
            printf("{\"probe\":\"sets\",\"set\":%d,\"rep\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s],\"pair_ms\":[%s]}\n", k, rep, e, all.c_str(), pers.c_str());
            fflush(stdout);
        }
        return 0;
    }
    if (mode == "valign") {
        // Does the VIRTUAL address the chunks are mapped at matter (the page-table fragment a mapping can use is bounded by the alignment of
        // the virtual address as well as of the physical block)?  One pool of 1-GiB chunks, mapped at a 1-GiB-aligned address plus an offset
        // of 0 / 2 / 4 / ... MiB inside an over-sized reservation; the engine on each; a hipMalloc'ed set before and after as the control.
        const size_t S = size_t(argc > 2 ? atol(argv[2]) : 1024) << 20, G1 = size_t(1) << 30;
        malloc_ctl("hipmalloc_before");
        const int nx = int((XB + S - 1) / S), nt_ = int((TB + S - 1) / S);
        Pool pool; pool.create(S, 2 * nx + nt_);
        std::vector<int> ix(nx), iv(nx), it(nt_);
        for (int i = 0; i < nx; ++i) { ix[i] = i; iv[i] = nx + i; }
        for (int i = 0; i < nt_; ++i) it[i] = 2 * nx + i;
        hipMemAccessDesc ad; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
        // what a plain reservation and one that asks for 1 GiB alignment return
        for (size_t al : {size_t(0), G1}) {
            void *p = nullptr; HIPCHECK(hipMemAddressReserve(&p, size_t(nx) * S, al, nullptr, 0));
            printf("{\"probe\":\"valign_reserve\",\"alignment_asked\":%zu,\"ptr\":\"%p\",\"mod_1gib_mib\":%zu}\n", al, p, (size_t(p) % G1) >> 20);
            HIPCHECK(hipMemAddressFree(p, size_t(nx) * S));
        }
        const size_t span = size_t(nx) * S + 2 * G1;
        char *rx = nullptr, *rv = nullptr, *rt = nullptr;
        HIPCHECK(hipMemAddressReserve((void **)&rx, span, 0, nullptr, 0));
        HIPCHECK(hipMemAddressReserve((void **)&rv, span, 0, nullptr, 0));
        HIPCHECK(hipMemAddressReserve((void **)&rt, size_t(nt_) * S + 2 * G1, 0, nullptr, 0));
        auto up = [&](char *p) { return (char *)((size_t(p) + G1 - 1) / G1 * G1); };
        const size_t offs_mib[] = {0, 2, 4, 8, 16, 32, 64, 128, 256, 512, 0};
        for (size_t om : offs_mib) {
            const size_t off = om << 20;
            char *bx = up(rx) + off, *bv = up(rv) + off, *bt = up(rt) + off;
            for (int i = 0; i < nx; ++i) { HIPCHECK(hipMemMap(bx + size_t(i) * S, S, 0, pool.h[ix[i]], 0)); HIPCHECK(hipMemMap(bv + size_t(i) * S, S, 0, pool.h[iv[i]], 0)); }
            for (int i = 0; i < nt_; ++i) HIPCHECK(hipMemMap(bt + size_t(i) * S, S, 0, pool.h[it[i]], 0));
            HIPCHECK(hipMemSetAccess(bx, size_t(nx) * S, &ad, 1)); HIPCHECK(hipMemSetAccess(bv, size_t(nx) * S, &ad, 1)); HIPCHECK(hipMemSetAccess(bt, size_t(nt_) * S, &ad, 1));
            std::string all; const float e = run_engine((double *)bx, (double *)bv, (double *)bt, &all);
            printf("{\"probe\":\"valign\",\"offset_mib\":%zu,\"engine_ms\":%.2f,\"engine_all\":[%s],\"x\":\"%p\",\"v\":\"%p\"}\n", om, e, all.c_str(), (void *)bx, (void *)bv);
            fflush(stdout);
            for (int i = 0; i < nx; ++i) { HIPCHECK(hipMemUnmap(bx + size_t(i) * S, S)); HIPCHECK(hipMemUnmap(bv + size_t(i) * S, S)); }
            for (int i = 0; i < nt_; ++i) HIPCHECK(hipMemUnmap(bt + size_t(i) * S, S));
        }
        pool.release();
        malloc_ctl("hipmalloc_after");
        return 0;
    }
    if (mode == "tswap") {
        // Which allocation decides?  One pool; (a) X, V fixed, the t array on twelve different chunks; (b) t fixed, X / V on other chunks;
        // (c) everything fixed, the engine handle (its scheduler, hand-off and status buffers) destroyed and re-created, with a spacer
        // allocation in between so that the new buffers land elsewhere; (d) the initial states re-allocated.
        const size_t S = size_t(argc > 2 ? atol(argv[2]) : 1024) << 20;
        Mapped X, V, T;
        X.reserve(XB, S); V.reserve(XB, S); T.reserve(TB, S);
        const int nx = int(X.n), nt_ = int(T.n), nsp = 14, npool = 4 * nx + nsp * nt_;
        Pool pool; pool.create(S, npool);
        std::vector<int> ix(nx), iv(nx), it(nt_);
        auto run = [&](const char *what, int k) {
            X.map(pool, ix); V.map(pool, iv); T.map(pool, it);
            std::string all; const float e = run_engine((double *)X.base, (double *)V.base, (double *)T.base, &all);
            X.unmap(); V.unmap(); T.unmap();
            printf("{\"probe\":\"tswap\",\"varied\":\"%s\",\"k\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s],\"x0\":%d,\"v0\":%d,\"t0\":%d}\n", what, k, e, all.c_str(), ix[0], iv[0], it[0]);
            fflush(stdout);
        };
        for (int i = 0; i < nx; ++i) { ix[i] = i; iv[i] = nx + i; }
        for (int k = 0; k < nsp; ++k) { for (int i = 0; i < nt_; ++i) it[i] = 4 * nx + k * nt_ + i; run("t", k); }
        for (int i = 0; i < nt_; ++i) it[i] = 4 * nx + i;
        for (int k = 0; k < 2; ++k) { for (int i = 0; i < nx; ++i) { ix[i] = 2 * k * nx + i; iv[i] = (2 * k + 1) * nx + i; } run("xv", k); }
        for (int i = 0; i < nx; ++i) { ix[i] = i; iv[i] = 3 * nx + i; }
        run("xv", 2);
        for (int i = 0; i < nx; ++i) { ix[i] = i; iv[i] = nx + i; }
        std::vector<void *> spacers;
        for (int k = 0; k < 6; ++k) {
            pdmp_destroy(g_h); g_h = nullptr;
            void *sp = nullptr; HIPCHECK(hipMalloc(&sp, size_t(37 + 61 * k) << 20)); spacers.push_back(sp);
            double *keep_x0 = g_x0, *keep_v0 = g_v0;
            engine_init();                                                    // (allocates new initial states as well; the old ones are freed below)
            HIPCHECK(hipFree(keep_x0)); HIPCHECK(hipFree(keep_v0));
            run("handle_and_inits", k);
        }
        for (void *sp : spacers) HIPCHECK(hipFree(sp));
        pool.release();
        return 0;
    }
    if (mode == "spacer") {
        // Is the process-wide constant the physical OFFSET the pool starts at (chunks created back to back are physically consecutive, so
        // every chunk of a pool has the same address modulo its size)?  A spacer allocation of k x 2 MiB is created -- and held -- before
        // the pool; the engine on each pool; (b) the same with the spacer between the X chunks and the V chunks.
        const size_t S = size_t(argc > 2 ? atol(argv[2]) : 1024) << 20, M2 = size_t(2) << 20;
        Mapped X, V, T;
        X.reserve(XB, S); V.reserve(XB, S); T.reserve(TB, S);
        const int nx = int(X.n), nt_ = int(T.n);
        hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        const size_t ks[] = {0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 3, 5, 96, 0};
        for (int where = 0; where < 2; ++where)
        for (size_t k : ks) {
            hipMemGenericAllocationHandle_t sp{}; 
            Pool px, pv;
            if (where == 0 && k) HIPCHECK(hipMemCreate(&sp, k * M2, &prop, 0));
            px.create(S, nx);
            if (where == 1 && k) HIPCHECK(hipMemCreate(&sp, k * M2, &prop, 0));
            pv.create(S, nx + nt_);
            std::vector<int> ix(nx), iv(nx), it(nt_);
            for (int i = 0; i < nx; ++i) { ix[i] = i; iv[i] = i; }
            for (int i = 0; i < nt_; ++i) it[i] = nx + i;
            X.map(px, ix); V.map(pv, iv); T.map(pv, it);
            std::string all; const float e = run_engine((double *)X.base, (double *)V.base, (double *)T.base, &all);
            const float pr = run_synth((double *)X.base, (double *)V.base, 2048, int64_t(S / 8 / 2048), 256, 1024, 2, 1, 2);
            X.unmap(); V.unmap(); T.unmap();
            printf("{\"probe\":\"spacer\",\"where\":\"%s\",\"spacer_mib\":%zu,\"engine_ms\":%.2f,\"engine_all\":[%s],\"pair0_ms\":%.3f}\n", where ? "between_x_and_v" : "before_pool", 2 * k, e, all.c_str(), pr);
            fflush(stdout);
            px.release(); pv.release();
            if (k) HIPCHECK(hipMemRelease(sp));
        }
        return 0;
    }
    if (mode == "vadelta") {
        // Is the process-wide constant the VIRTUAL distance between the X and the V array (their reservations are made once per process)?
        // One pool, one big reservation; X mapped at its 1-GiB-aligned start, V at X + 38 GiB + delta for a list of deltas; t fixed.
        const size_t S = size_t(1024) << 20, G1 = size_t(1) << 30;
        const int nx = int((XB + S - 1) / S), nt_ = int((TB + S - 1) / S);
        Pool pool; pool.create(S, 2 * nx + nt_);
        hipMemAccessDesc ad; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
        const size_t span = size_t(2 * nx + 70) * G1;
        char *r = nullptr, *rt = nullptr;
        HIPCHECK(hipMemAddressReserve((void **)&r, span, 0, nullptr, 0));
        HIPCHECK(hipMemAddressReserve((void **)&rt, size_t(nt_ + 1) * G1, 0, nullptr, 0));
        char *bx = (char *)((size_t(r) + G1 - 1) / G1 * G1), *bt = (char *)((size_t(rt) + G1 - 1) / G1 * G1);
        for (int i = 0; i < nx; ++i) HIPCHECK(hipMemMap(bx + size_t(i) * S, S, 0, pool.h[i], 0));
        for (int i = 0; i < nt_; ++i) HIPCHECK(hipMemMap(bt + size_t(i) * S, S, 0, pool.h[2 * nx + i], 0));
        HIPCHECK(hipMemSetAccess(bx, size_t(nx) * S, &ad, 1)); HIPCHECK(hipMemSetAccess(bt, size_t(nt_) * S, &ad, 1));
        const double deltas_gib[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 64, 0.5, 0.25, 0.001953125 /* 2 MiB */, 1.5, 0};
        for (double dg : deltas_gib) {
            char *bv = bx + size_t(nx) * S + size_t(dg * double(G1));
            for (int i = 0; i < nx; ++i) HIPCHECK(hipMemMap(bv + size_t(i) * S, S, 0, pool.h[nx + i], 0));
            HIPCHECK(hipMemSetAccess(bv, size_t(nx) * S, &ad, 1));
            std::string all; const float e = run_engine((double *)bx, (double *)bv, (double *)bt, &all);
            printf("{\"probe\":\"vadelta\",\"delta_gib\":%.6f,\"engine_ms\":%.2f,\"engine_all\":[%s],\"x\":\"%p\",\"v\":\"%p\",\"t\":\"%p\"}\n", dg, e, all.c_str(), (void *)bx, (void *)bv, (void *)bt);
            fflush(stdout);
            for (int i = 0; i < nx; ++i) HIPCHECK(hipMemUnmap(bv + size_t(i) * S, S));
        }
        // and X, V, t in fresh, separate reservations (what the allocator does today), six times
        for (int k = 0; k < 6; ++k) {
            Mapped X, V, T;
            void *pad = nullptr; HIPCHECK(hipMemAddressReserve(&pad, size_t(3 + 5 * k) * G1 + (size_t(k) << 21), 0, nullptr, 0));
            X.reserve(XB, S); V.reserve(XB, S); T.reserve(TB, S);
            std::vector<int> ix(nx), iv(nx), it(nt_);
            for (int i = 0; i < nx; ++i) { ix[i] = i; iv[i] = nx + i; }
            for (int i = 0; i < nt_; ++i) it[i] = 2 * nx + i;
            for (int i = 0; i < nx; ++i) HIPCHECK(hipMemUnmap(bx + size_t(i) * S, S));          // a handle is mapped once at a time here
            for (int i = 0; i < nt_; ++i) HIPCHECK(hipMemUnmap(bt + size_t(i) * S, S));
            X.map(pool, ix); V.map(pool, iv); T.map(pool, it);
            std::string all; const float e = run_engine((double *)X.base, (double *)V.base, (double *)T.base, &all);
            printf("{\"probe\":\"vadelta_fresh\",\"k\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s],\"x\":\"%p\",\"v\":\"%p\",\"t\":\"%p\"}\n", k, e, all.c_str(), (void *)X.base, (void *)V.base, (void *)T.base);
            fflush(stdout);
            X.unmap(); V.unmap(); T.unmap();
            for (int i = 0; i < nx; ++i) HIPCHECK(hipMemMap(bx + size_t(i) * S, S, 0, pool.h[i], 0));
            for (int i = 0; i < nt_; ++i) HIPCHECK(hipMemMap(bt + size_t(i) * S, S, 0, pool.h[2 * nx + i], 0));
            HIPCHECK(hipMemSetAccess(bx, size_t(nx) * S, &ad, 1)); HIPCHECK(hipMemSetAccess(bt, size_t(nt_) * S, &ad, 1));
        }
        pool.release();
        return 0;
    }
    if (mode == "vamap") {
        // The launch time as a function of WHERE in the virtual address space X and V are mapped (same physical chunks throughout): X at
        // base + ox GiB, V at base + ov GiB inside one reservation whose base is aligned to 256 GiB; t fixed elsewhere.
        const size_t S = size_t(1024) << 20, G1 = size_t(1) << 30;
        const int nx = int((XB + S - 1) / S), nt_ = int((TB + S - 1) / S);
        Pool pool; pool.create(S, 2 * nx + nt_);
        hipMemAccessDesc ad; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
        const size_t span = size_t(1024 + 256) * G1;
        char *r = nullptr, *rt = nullptr;
        HIPCHECK(hipMemAddressReserve((void **)&r, span, 0, nullptr, 0));
        HIPCHECK(hipMemAddressReserve((void **)&rt, size_t(nt_ + 1) * G1, 0, nullptr, 0));
        char *base = (char *)((size_t(r) + 256 * G1 - 1) / (256 * G1) * (256 * G1)), *bt = (char *)((size_t(rt) + G1 - 1) / G1 * G1);
        printf("{\"probe\":\"vamap_base\",\"base\":\"%p\",\"t\":\"%p\"}\n", (void *)base, (void *)bt);
        for (int i = 0; i < nt_; ++i) HIPCHECK(hipMemMap(bt + size_t(i) * S, S, 0, pool.h[2 * nx + i], 0));
        HIPCHECK(hipMemSetAccess(bt, size_t(nt_) * S, &ad, 1));
        auto at = [&](int ox, int ov) {
            char *bx = base + size_t(ox) * G1, *bv = base + size_t(ov) * G1;
            for (int i = 0; i < nx; ++i) { HIPCHECK(hipMemMap(bx + size_t(i) * S, S, 0, pool.h[i], 0)); HIPCHECK(hipMemMap(bv + size_t(i) * S, S, 0, pool.h[nx + i], 0)); }
            HIPCHECK(hipMemSetAccess(bx, size_t(nx) * S, &ad, 1)); HIPCHECK(hipMemSetAccess(bv, size_t(nx) * S, &ad, 1));
            std::string all; const float e = run_engine((double *)bx, (double *)bv, (double *)bt, &all, 2);
            printf("{\"probe\":\"vamap\",\"ox\":%d,\"ov\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s]}\n", ox, ov, e, all.c_str());
            fflush(stdout);
            for (int i = 0; i < nx; ++i) { HIPCHECK(hipMemUnmap(bx + size_t(i) * S, S)); HIPCHECK(hipMemUnmap(bv + size_t(i) * S, S)); }
        };
        const int oxs[] = {0, 8, 16, 32, 64, 100};
        for (int ox : oxs)
            for (int d = 0; d <= 160; d += 4) at(ox, ox + nx + d);
        for (int d = 0; d <= 160; d += 8) at(200 + nx + d, 200);          // V below X
        at(0, nx); at(0, nx + 24);
        pool.release();
        return 0;
    }
    if (mode == "vascan") {
        // The launch time over the virtual address of X across two and a half 512-GiB windows (V right behind X, t fixed far away), then
        // with t right behind V, then V far from X: which array's address decides, and with which period?
        const size_t S = size_t(1024) << 20, G1 = size_t(1) << 30;
        const int nx = int((XB + S - 1) / S), nt_ = int((TB + S - 1) / S);
        Pool pool; pool.create(S, 2 * nx + nt_);
        hipMemAccessDesc ad; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
        const size_t span = size_t(512 * 3 + 512) * G1;
        char *r = nullptr;
        HIPCHECK(hipMemAddressReserve((void **)&r, span, 0, nullptr, 0));
        char *base = (char *)((size_t(r) + 512 * G1 - 1) / (512 * G1) * (512 * G1));
        printf("{\"probe\":\"vascan_base\",\"reserved\":\"%p\",\"base\":\"%p\",\"x0\":\"%p\",\"v0\":\"%p\"}\n", (void *)r, (void *)base, (void *)g_x0, (void *)g_v0);
        auto at = [&](const char *what, int ox, int ov, int ot) {
            char *bx = base + size_t(ox) * G1, *bv = base + size_t(ov) * G1, *bt = base + size_t(ot) * G1;
            for (int i = 0; i < nx; ++i) { HIPCHECK(hipMemMap(bx + size_t(i) * S, S, 0, pool.h[i], 0)); HIPCHECK(hipMemMap(bv + size_t(i) * S, S, 0, pool.h[nx + i], 0)); }
            for (int i = 0; i < nt_; ++i) HIPCHECK(hipMemMap(bt + size_t(i) * S, S, 0, pool.h[2 * nx + i], 0));
            HIPCHECK(hipMemSetAccess(bx, size_t(nx) * S, &ad, 1)); HIPCHECK(hipMemSetAccess(bv, size_t(nx) * S, &ad, 1)); HIPCHECK(hipMemSetAccess(bt, size_t(nt_) * S, &ad, 1));
            std::string all; const float e = run_engine((double *)bx, (double *)bv, (double *)bt, &all, 2);
            printf("{\"probe\":\"vascan\",\"series\":\"%s\",\"ox\":%d,\"ov\":%d,\"ot\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s]}\n", what, ox, ov, ot, e, all.c_str());
            fflush(stdout);
            for (int i = 0; i < nx; ++i) { HIPCHECK(hipMemUnmap(bx + size_t(i) * S, S)); HIPCHECK(hipMemUnmap(bv + size_t(i) * S, S)); }
            for (int i = 0; i < nt_; ++i) HIPCHECK(hipMemUnmap(bt + size_t(i) * S, S));
        };
        const int TFAR = 512 * 3 + 100;
        for (int ox = 0; ox <= 1320; ox += 24) at("x_moves_t_far", ox, ox + nx, TFAR);
        for (int ox = 0; ox <= 600; ox += 40) at("x_moves_t_behind_v", ox, ox + nx, ox + 2 * nx);
        for (int ox = 0; ox <= 600; ox += 40) at("x_moves_t_before_x", ox + 1, ox + 1 + nx, ox);
        for (int ov = 40; ov <= 640; ov += 40) at("v_moves_x_at_0", 0, ov, TFAR);
        for (int ot = 80; ot <= 680; ot += 40) at("t_moves_x_at_0", 0, nx, ot);
        pool.release();
        return 0;
    }
    if (mode == "stale") {
        // Does an address that is unmapped and then mapped to OTHER physical memory really point there?  Chunk A mapped at p, filled with
        // 1; p unmapped; chunk B mapped at p, filled with 2 through p.  Then p is read (expect 2), and A and B are each mapped at a FRESH
        // address (never used before) and read: A should hold 1, B should hold 2.  variant 1: with a device synchronisation and 20 ms of
        // sleep after the re-map; variant 2: the reservation of p is freed and reserved again between the two mappings.
        const size_t S = size_t(argc > 2 ? atol(argv[2]) : 64) << 20;
        const int iters = argc > 3 ? atoi(argv[3]) : 4;
        const size_t n = S / 8;
        std::vector<double> host(n);
        auto count = [&](const void *dev, double want) {
            HIPCHECK(hipMemcpy(host.data(), dev, S, hipMemcpyDeviceToHost));
            long bad = 0; for (size_t i = 0; i < n; ++i) bad += host[i] != want;
            printf("{\"probe\":\"stale_read\",\"ptr\":\"%p\",\"want\":%g,\"first\":%g,\"middle\":%g,\"last\":%g,\"wrong\":%ld}\n", dev, want, host[0], host[n / 2], host[n - 1], bad);
            return bad;
        };
        for (int variant = 0; variant < 3; ++variant)
        for (int it = 0; it < iters; ++it) {
            Pool pool; pool.create(S, 2);
            Mapped P, QA, QB; P.reserve(S, S);
            void *p_first = P.base;
            P.map(pool, std::vector<int>{0});
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (double *)P.base, n, 1.0);
            HIPCHECK(hipDeviceSynchronize());
            P.unmap();
            if (variant == 2) { P.free_va(); P.reserve(S, S); }
            P.map(pool, std::vector<int>{1});
            if (variant == 1) { HIPCHECK(hipDeviceSynchronize()); usleep(20000); }
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (double *)P.base, n, 2.0);
            HIPCHECK(hipDeviceSynchronize());
            const long bad_p = count(P.base, 2.0);
            QA.reserve(S, S); QA.map(pool, std::vector<int>{0});
            const long bad_a = count(QA.base, 1.0);
            QB.reserve(S, S); QB.map(pool, std::vector<int>{1});
            const long bad_b = count(QB.base, 2.0);
            printf("{\"probe\":\"stale\",\"variant\":%d,\"iteration\":%d,\"p_first\":\"%p\",\"p_second\":\"%p\",\"qa\":\"%p\",\"qb\":\"%p\",\"wrong_through_p\":%ld,\"wrong_in_A\":%ld,\"wrong_in_B\":%ld,\"doubles_per_chunk\":%zu}\n",
                   variant, it, p_first, (void *)P.base, (void *)QA.base, (void *)QB.base, bad_p, bad_a, bad_b, n);
            fflush(stdout);
            P.unmap(); QA.unmap(); QB.unmap();
            P.free_va(); QA.free_va(); QB.free_va();
            pool.release();
        }
        return 0;
    }
    if (mode == "stale2") {
        // (a) the product's allocator: pdmp_device_alloc, fill by a kernel, pdmp_device_free, pdmp_device_alloc again (same address?), fill
        // with another value, read back by hipMemcpy AND by a kernel (sum).  (b) what makes a re-mapped address valid for kernels: nothing /
        // a hipMalloc + hipFree of 64 MiB in between / a second of sleep / a dummy kernel first.
        const size_t S = size_t(64) << 20, n = S / 8;
        std::vector<double> host(n);
        auto read_copy = [&](const void *dev) { HIPCHECK(hipMemcpy(host.data(), dev, S, hipMemcpyDeviceToHost)); return host[n / 2]; };
        double *d_sum; HIPCHECK(hipMalloc(&d_sum, 8));
        auto read_kernel = [&](const void *dev) {
            HIPCHECK(hipMemset(d_sum, 0, 8));
            hipLaunchKernelGGL(probe_read_kernel, dim3(1), dim3(64), 0, 0, (const double *)dev, n, d_sum);
            double v; HIPCHECK(hipMemcpy(&v, d_sum, 8, hipMemcpyDeviceToHost)); return v;
        };
        for (int it = 0; it < 3; ++it) {
            void *a1 = nullptr, *a2 = nullptr;
            if (pdmp_device_alloc(0, 4 * S, S, &a1) != PDMP_OK) { fprintf(stderr, "%s\n", pdmp_device_alloc_error()); return 2; }
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (double *)a1, 4 * n, 1.0 + it);
            HIPCHECK(hipDeviceSynchronize());
            const double c1 = read_copy(a1), k1 = read_kernel(a1);
            pdmp_device_free(a1);
            if (pdmp_device_alloc(0, 4 * S, S, &a2) != PDMP_OK) { fprintf(stderr, "%s\n", pdmp_device_alloc_error()); return 2; }
            const double c2_before = read_copy(a2), k2_before = read_kernel(a2);
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (double *)a2, 4 * n, 100.0 + it);
            HIPCHECK(hipDeviceSynchronize());
            const double c2 = read_copy(a2), k2 = read_kernel(a2);
            printf("{\"probe\":\"stale2_alloc_free\",\"iteration\":%d,\"a1\":\"%p\",\"a2\":\"%p\",\"a1_copy\":%g,\"a1_kernel\":%g,\"a2_before_copy\":%g,\"a2_before_kernel\":%g,\"a2_copy\":%g,\"a2_kernel\":%g,\"want_a2\":%g}\n",
                   it, a1, a2, c1, k1, c2_before, k2_before, c2, k2, 100.0 + it);
            fflush(stdout);
            pdmp_device_free(a2);
        }
        for (int flush = 0; flush < 4; ++flush) {
            Pool pool; pool.create(S, 2);
            Mapped P, QA, QB; P.reserve(S, S);
            P.map(pool, std::vector<int>{0});
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (double *)P.base, n, 1.0);
            HIPCHECK(hipDeviceSynchronize());
            P.unmap();
            P.map(pool, std::vector<int>{1});
            if (flush == 1) { void *t = nullptr; HIPCHECK(hipMalloc(&t, S)); HIPCHECK(hipFree(t)); }
            if (flush == 2) sleep(1);
            if (flush == 3) { hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, 0, d_sum, size_t(1), 0.0); HIPCHECK(hipDeviceSynchronize()); }
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (double *)P.base, n, 2.0);
            HIPCHECK(hipDeviceSynchronize());
            const double pc = read_copy(P.base), pk = read_kernel(P.base);
            QA.reserve(S, S); QA.map(pool, std::vector<int>{0});
            const double ac = read_copy(QA.base);
            QB.reserve(S, S); QB.map(pool, std::vector<int>{1});
            const double bc = read_copy(QB.base);
            printf("{\"probe\":\"stale2_remap\",\"between\":\"%s\",\"p_copy\":%g,\"p_kernel\":%g,\"A_holds\":%g,\"B_holds\":%g,\"want\":\"p 2, A 1, B 2\"}\n",
                   flush == 0 ? "nothing" : flush == 1 ? "hipMalloc+hipFree 64 MiB" : flush == 2 ? "sleep 1 s" : "dummy kernel", pc, pk, ac, bc);
            fflush(stdout);
            P.unmap(); QA.unmap(); QB.unmap(); P.free_va(); QA.free_va(); QB.free_va();
            pool.release();
        }
        return 0;
    }
    if (mode == "pairs") {
        // Is "slow" a RELATION between the X chunk and the V chunk written together, and can an allocator arrange it?  1-GiB chunks; the dense
        // two-array writer on (chunk i, chunk j) for a block of the pool (the matrix shows whether the relation has classes); then the engine
        // on three sets made of the same pool: the pairing as allocated, a pairing chosen so that every (X chunk, V chunk) pair is fast, and
        // one chosen so that every pair is slow.
        const size_t S = size_t(argc > 2 ? atol(argv[2]) : 1024) << 20;
        const int spare = argc > 3 ? atoi(argv[3]) : 44, nmat = argc > 4 ? atoi(argv[4]) : 24;
        const float thr = argc > 5 ? float(atof(argv[5])) : 0.52f;
        Mapped X, V, T;
        X.reserve(XB, S); V.reserve(XB, S); T.reserve(TB, S);
        const int nx = int(X.n), nt_ = int(T.n), npool = 2 * nx + nt_ + spare;
        malloc_ctl("hipmalloc_before");
        Pool pool; pool.create(S, npool);
        Mapped A, B; A.reserve(S, S); B.reserve(S, S);
        const int64_t st = 2048, nch = int64_t(S / 8 / st);
        int tests = 0;
        {   // every chunk alone (one array): a globally slow box shows here, a relation between chunks does not
            std::string one;
            for (int i = 0; i < npool; ++i) {
                A.map(pool, std::vector<int>{i});
                const float m = run_synth((double *)A.base, (double *)A.base, st, nch, int(st / 8), 1024, 1, 1, 2);
                A.unmap();
                char buf[32]; snprintf(buf, sizeof buf, "%s%.3f", i ? "," : "", m); one += buf;
            }
            printf("{\"probe\":\"single_chunk_ms\",\"ms\":[%s]}\n", one.c_str());
            fflush(stdout);
        }
        auto pair = [&](int i, int j) {
            A.map(pool, std::vector<int>{i}); B.map(pool, std::vector<int>{j});
            const float m = run_synth((double *)A.base, (double *)B.base, st, nch, int(st / 8), 1024, 2, 1, 2);
            A.unmap(); B.unmap(); ++tests;
            return m;
        };
        const int mstep = std::max(1, npool / nmat);               // the matrix over chunks spread through the pool
        for (int i = 0; i < nmat && i * mstep < npool; ++i) {
            std::string row;
            for (int j = 0; j < nmat && j * mstep < npool; ++j) {
                char buf[32]; snprintf(buf, sizeof buf, "%s%.3f", j ? "," : "", i == j ? 0.f : pair(i * mstep, j * mstep)); row += buf;
            }
            printf("{\"probe\":\"pair_matrix\",\"i\":%d,\"ms\":[%s]}\n", i * mstep, row.c_str());
            fflush(stdout);
        }
        // sets "as allocated": set k = chunks [k per, (k + 1) per): X, then V, then t -- what an allocator that takes chunks in order gets
        const int per = 2 * nx + nt_, nsets = npool / per;
        std::vector<int> ix(nx), iv(nx), it(nt_);
        for (int i = 0; i < nt_; ++i) it[i] = 2 * nx + i;
        T.map(pool, it);
        auto run_set = [&](const char *name, int k, const std::vector<int> &jv) {
            std::string pers; int slow = 0;
            for (int i = 0; i < nx; ++i) {
                const float m = pair(ix[i], jv[i]); slow += m >= thr;
                char buf[32]; snprintf(buf, sizeof buf, "%s%.3f", i ? "," : "", m); pers += buf;
            }
            X.map(pool, ix); V.map(pool, jv);
            std::string all; const float e = run_engine((double *)X.base, (double *)V.base, (double *)T.base, &all);
            const float s2 = run_synth((double *)X.base, (double *)V.base, NSK * D, C, int(NSK * D / 8), 1024, 2, 1);
            X.unmap(); V.unmap();
            printf("{\"probe\":\"pairs_set\",\"pairing\":\"%s\",\"set\":%d,\"slow_pairs\":%d,\"of\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s],\"synth2_ms\":%.2f,\"v_ids\":%s,\"pair_ms\":[%s]}\n",
                   name, k, slow, nx, e, all.c_str(), s2, ids_str(jv).c_str(), pers.c_str());
            fflush(stdout);
            return slow;
        };
        int worst = 0, worst_slow = -1;
        for (int k = 0; k < nsets; ++k) {
            for (int i = 0; i < nx; ++i) { ix[i] = k * per + i; iv[i] = k * per + nx + i; }
            const int sl = run_set("as_allocated", k, iv);
            if (sl > worst_slow) { worst_slow = sl; worst = k; }
        }
        // the set with the most slow pairs, re-paired: for X chunk i the first unused candidate (any chunk of the pool that is not an X chunk
        // of this set or a t chunk) whose pair time is on the wanted side of the threshold
        for (int i = 0; i < nx; ++i) { ix[i] = worst * per + i; iv[i] = worst * per + nx + i; }
        auto arrange = [&](bool want_fast) {
            std::vector<int> cand;
            for (int j = 0; j < npool; ++j) {
                const bool is_x = j >= worst * per && j < worst * per + nx, is_t = j >= 2 * nx && j < 2 * nx + nt_;
                if (!is_x && !is_t) cand.push_back(j);
            }
            // this set's own V chunks first (keep what is already good)
            std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { const bool oa = a >= worst * per + nx && a < worst * per + 2 * nx, ob = b >= worst * per + nx && b < worst * per + 2 * nx; return oa > ob; });
            std::vector<char> used(cand.size(), 0);
            std::vector<int> jv(nx);
            for (int i = 0; i < nx; ++i) {
                int pick = -1; float best = want_fast ? 1e30f : -1.f; int bestk = -1, tried = 0;
                for (size_t k = 0; k < cand.size() && tried < 24; ++k) {
                    if (used[k]) continue;
                    const float m = pair(ix[i], cand[k]); ++tried;
                    if (want_fast ? m < best : m > best) { best = m; bestk = int(k); }
                    if (want_fast ? m < thr : m >= thr) { pick = int(k); break; }
                }
                if (pick < 0) pick = bestk;
                used[pick] = 1; jv[i] = cand[pick];
            }
            return jv;
        };
        int t0 = tests;
        const std::vector<int> fastv = arrange(true);
        printf("{\"probe\":\"pairs_arrange\",\"want\":\"fast\",\"set\":%d,\"pair_tests\":%d}\n", worst, tests - t0);
        run_set("all_fast", worst, fastv);
        t0 = tests;
        const std::vector<int> slowv = arrange(false);
        printf("{\"probe\":\"pairs_arrange\",\"want\":\"slow\",\"set\":%d,\"pair_tests\":%d}\n", worst, tests - t0);
        run_set("all_slow", worst, slowv);
        run_set("as_allocated_again", worst, iv);
        run_set("all_fast_again", worst, fastv);
        T.unmap();
        pool.release();
        malloc_ctl("hipmalloc_after");
        return 0;
    }
    if (mode == "chunks") {
        // Arrays from 1-GiB physical chunks: is a slow placement slow in particular chunks?  Per pool: the engine launch, the synthetic
        // writer on every (X chunk i, V chunk i) pair alone, then the slowest pairs exchanged for spare chunks and the launch again.
        const size_t S = size_t(argc > 2 ? atol(argv[2]) : 1024) << 20;
        const int R = argc > 3 ? atoi(argv[3]) : 3, spare = argc > 4 ? atoi(argv[4]) : 16;
        Mapped X, V, T;
        X.reserve(XB, S); V.reserve(XB, S); T.reserve(TB, S);
        const int nx = int(X.n), nt_ = int(T.n), npool = 2 * nx + nt_ + spare;
        const int64_t chains_per_chunk = int64_t(S / (NSK * D * 8));
        for (int r = 0; r < R; ++r) {
            Pool pool; pool.create(S, npool);
            std::vector<int> ix(nx), iv(nx), it(nt_);
            for (int i = 0; i < nx; ++i) { ix[i] = i; iv[i] = nx + i; }
            for (int i = 0; i < nt_; ++i) it[i] = 2 * nx + i;
            X.map(pool, ix); V.map(pool, iv); T.map(pool, it);
            std::string all; float e = run_engine((double *)X.base, (double *)V.base, (double *)T.base, &all);
            std::vector<float> per(nx);
            std::string pers;
            for (int i = 0; i < nx; ++i) {
                // chains whose rows lie wholly inside chunk i
                const int64_t c0 = (int64_t(i) * int64_t(S) + NSK * D * 8 - 1) / (NSK * D * 8), c1 = std::min<int64_t>(C, (int64_t(i + 1) * int64_t(S)) / (NSK * D * 8));
                per[i] = run_synth((double *)X.base + c0 * NSK * D, (double *)V.base + c0 * NSK * D, NSK * D, c1 - c0, int(NSK * D / 8), 1024, 2, 1, 2);
                char buf[32]; snprintf(buf, sizeof buf, "%s%.3f", i ? "," : "", per[i]); pers += buf;
            }
            printf("{\"probe\":\"chunks\",\"chunk_mib\":%zu,\"pool\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s],\"chains_per_chunk\":%lld,\"pair_ms\":[%s]}\n",
                   S >> 20, r, e, all.c_str(), (long long)chains_per_chunk, pers.c_str());
            fflush(stdout);
            // exchange the slowest pairs for spare chunks (both members of the pair), up to spare/2 pairs
            std::vector<int> order(nx); for (int i = 0; i < nx; ++i) order[i] = i;
            std::sort(order.begin(), order.end(), [&](int a, int b) { return per[a] > per[b]; });
            X.unmap(); V.unmap();
            int next = 2 * nx + nt_;
            for (int k = 0; k < spare / 2 && k < nx; ++k) { ix[order[k]] = next++; iv[order[k]] = next++; }
            X.map(pool, ix); V.map(pool, iv);
            std::string all2; float e2 = run_engine((double *)X.base, (double *)V.base, (double *)T.base, &all2);
            printf("{\"probe\":\"chunks_swapped\",\"pool\":%d,\"pairs_swapped\":%d,\"engine_ms\":%.2f,\"engine_all\":[%s]}\n", r, spare / 2, e2, all2.c_str());
            fflush(stdout);
            X.unmap(); V.unmap(); T.unmap();
            pool.release();
        }
        return 0;
    }
    // ---- vmm / synth
    const size_t S = size_t(argc > 2 ? atol(argv[2]) : 2048) << 20;
    const int R = argc > 3 ? atoi(argv[3]) : 3, P = argc > 4 ? atoi(argv[4]) : 3;
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    HIPCHECK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    HIPCHECK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("{\"probe\":\"granularity\",\"min\":%zu,\"recommended\":%zu,\"chunk\":%zu}\n", gmin, grec, S);
    Mapped X, V, T;
    X.reserve(XB, S); V.reserve(XB, S); T.reserve(TB, S);
    const int nx = int(X.n), nt_ = int(T.n), npool = 2 * nx + nt_;
    std::mt19937 prng(7);
    for (int r = 0; r < R; ++r) {
        Pool pool; pool.create(S, npool);
        std::vector<int> idx(npool); for (int i = 0; i < npool; ++i) idx[i] = i;
        for (int p = 0; p <= P; ++p) {
            std::vector<int> perm = idx;
            const char *name = "identity";
            if (p == 1) { std::reverse(perm.begin(), perm.begin() + 2 * nx); name = "reversed"; }
            else if (p == 2) { for (int i = 0; i < nx; ++i) { perm[i] = 2 * i; perm[nx + i] = 2 * i + 1; } name = "interleaved"; }   /* X and V chunks alternate */
            else if (p >= 3) { std::shuffle(perm.begin(), perm.begin() + 2 * nx, prng); name = "shuffled"; }
            std::vector<int> ix(perm.begin(), perm.begin() + nx), iv(perm.begin() + nx, perm.begin() + 2 * nx), it(perm.begin() + 2 * nx, perm.end());
            X.map(pool, ix); V.map(pool, iv); T.map(pool, it);
            std::string all; float e = run_engine((double *)X.base, (double *)V.base, (double *)T.base, &all);
            float s2 = run_synth((double *)X.base, (double *)V.base, NSK * D, C, int(NSK * D / 8), 1024, 2, 1);
            float s1 = run_synth((double *)X.base, (double *)V.base, NSK * D, C, int(NSK * D / 8), 1024, 1, 1);
            printf("{\"probe\":\"vmm\",\"chunk_mib\":%zu,\"pool\":%d,\"perm\":\"%s\",\"engine_ms\":%.2f,\"engine_all\":[%s],\"synth2_ms\":%.2f,\"synth1_ms\":%.2f,\"x_ids\":%s}\n",
                   S >> 20, r, name, e, all.c_str(), s2, s1, ids_str(ix, 24).c_str());
            fflush(stdout);
            if (mode == "synth" && p == 0) {
                // the writer on subsets: chains of the first / second half (different chunks), and X-only against the V chunks
                const int64_t half = C / 2;
                float a = run_synth((double *)X.base, (double *)V.base, NSK * D, half, int(NSK * D / 8), 1024, 2, 1);
                float b = run_synth((double *)X.base + half * NSK * D, (double *)V.base + half * NSK * D, NSK * D, half, int(NSK * D / 8), 1024, 2, 1);
                float c = run_synth((double *)V.base, (double *)X.base, NSK * D, C, int(NSK * D / 8), 1024, 1, 1);
                float pl = run_synth((double *)X.base, (double *)V.base, NSK * D, C, int(NSK * D / 8), 1024, 2, 0);
                printf("{\"probe\":\"synth_subsets\",\"pool\":%d,\"first_half_ms\":%.2f,\"second_half_ms\":%.2f,\"v_only_ms\":%.2f,\"plain_store_ms\":%.2f}\n", r, a, b, c, pl);
                fflush(stdout);
            }
            X.unmap(); V.unmap(); T.unmap();
        }
        pool.release();
    }
    return 0;
}
