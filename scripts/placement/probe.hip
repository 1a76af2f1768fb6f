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
    }
    void unmap() { for (size_t i = 0; i < n; ++i) HIPCHECK(hipMemUnmap(base + i * S, S)); }
    void free_va() { HIPCHECK(hipMemAddressFree(base, n * S)); base = nullptr; }
};

static std::string ids_str(const std::vector<int> &v, size_t lim = 64)
{
    std::string s = "[";
    for (size_t i = 0; i < v.size() && i < lim; ++i) s += (i ? "," : "") + std::to_string(v[i]);
    return s + (v.size() > lim ? ",...]" : "]");
}

int main(int argc, char **argv)
{
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
