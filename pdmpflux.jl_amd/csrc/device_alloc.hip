// device_alloc.hip -- output arrays of an ensemble built from independent physical chunks (HIP virtual memory management), and the
// streamed arrays of an ensemble drawn several times with the fastest set kept.
//
// Why (profiles/r4/placement/): the launch that streams the event rows of X and V side by side runs 14.3 ms or
// 17 - 19 ms depending on where the physical memory behind the two arrays lies -- chunks of the VRAM fall into a few classes (regions of
// tens of GiB), an X chunk and the V chunk written together with it are 50 % slower when they come from the same class (write requests
// then stall on DRAM credits 3 - 8 x as often; translation misses unchanged), and a hipMalloc'ed 40 GB array is a few huge blocks, so
// the relation is drawn once for the whole array.  Nothing in user space says which class a chunk is in, but a set of arrays can be
// JUDGED in 20 ms by a kernel with the launch's store pattern: pdmp_device_alloc_streams draws up to `draws` sets and keeps the fastest
// (three consecutive draws of one process: 15.3 / 14.4 / 17.0 ms, 14.5 / 14.3 / 14.4, 15.3 / 14.3 / 15.7, 16.0 / 14.8 / 15.0).  The chunks
// must not be small either: below ~32 MiB of physically contiguous memory per translation the kernel's 131 000 concurrent write streams
// miss in the translation caches (TCP_UTCL1_TRANSLATION_MISS 1.2e5 -> 7e7 .. 1.3e8 per launch; 16.0 ms at 16 MiB, 22.5 ms at 2 MiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/pdmp_hip.h"

namespace {

struct ChunkedAlloc {
    int device = 0;
    size_t va_bytes = 0, bytes = 0, chunk = 0;      // bytes / chunk: what the caller asked for (the pool's key)
    long long set_id = 0;                           // > 0: one of the arrays of a set pdmp_device_alloc_streams judged and kept; index in the set below
    int set_index = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
};
std::mutex g_mu;
std::map<void *, ChunkedAlloc> g_allocs;
// The pool (round 5): a freed allocation keeps its address range AND its mapping and waits here for the next request of the same size on
// the same device -- a caller that frees and re-allocates its output arrays (every fresh `alloc_outputs`) gets the very arrays back,
// in microseconds instead of the 3 - 6.5 s of mapping and judging 80 GB again, and because nothing is ever unmapped and mapped anew there
// is no stale translation to worry about.  Parked memory is released when an allocation would otherwise fail, by pdmp_device_pool_trim(),
// or never parked at all with PDMP_ALLOC_POOL=0 (read once).
std::map<void *, ChunkedAlloc> g_pool;
long long g_next_set = 1;
thread_local bool g_release_now = false;        // pdmp_device_alloc_streams frees the sets it did not keep for real (they must not wait in the pool beside the kept one)
bool pool_enabled() {
    static const bool on = [] { const char *e = getenv("PDMP_ALLOC_POOL"); return !(e && e[0] == '0'); }();
    return on;
}

thread_local char g_err[256];

// An address that has been unmapped (hipMemUnmap) and mapped to OTHER physical memory keeps its old translation for the compute queues
// on this stack: kernels go on reading and writing the memory that was mapped there first -- even after that memory has been released --
// while hipMemcpy sees the new mapping (scripts/placement/probe.hip, mode `stale2`; profiles/r4/placement/c15_stale2.jsonl).  The unmap
// of an ORDINARY allocation flushes the translation caches, so every mapping made here is followed by one hipMalloc + hipFree of 64 MiB
// (c16_stale2.jsonl: with it, kernel and copy agree in every variant).  Without this a pdmp_device_alloc that is handed the address range
// of an earlier, freed allocation would let kernels write into memory that is no longer theirs.
// Returns false when no trial allocation succeeded (memory completely full): the caller must then NOT hand out the mapping.
// Since round 5 the hazard itself is designed out as well: an address range released here is never given back to the driver
// (release_all keeps the reservation), so no later mapping can land on a range with old translations; the flush stays as the second fence.
bool flush_translations() {
    bool ok = false;
    for (size_t mib = 64; mib >= 2 && !ok; mib /= 4) {                      // (64 MiB is what was measured; smaller ones only when the memory is that full)
        void *t = nullptr;
        if (hipMalloc(&t, mib << 20) == hipSuccess) { (void)hipFree(t); ok = true; }
        else (void)hipGetLastError();
    }
    (void)hipGetLastError();
    return ok;
}

void release_all(char *base, ChunkedAlloc &a, size_t mapped) {
    size_t off = 0;
    for (size_t i = 0; i < a.handles.size(); ++i) {
        if (i < mapped) (void)hipMemUnmap(base + off, a.sizes[i]);
        (void)hipMemRelease(a.handles[i]);
        off += a.sizes[i];
    }
    // the address range stays reserved for the life of the process (it is only address space: 2^47 bytes of it): a range that is never
    // returned can never be handed out again with stale translations behind it
    (void)base;
}
void release_pool_locked() {
    for (auto &kv : g_pool) release_all((char *)kv.first, kv.second, kv.second.handles.size());
    g_pool.clear();
}

}  // namespace

extern "C" {

const char *pdmp_device_alloc_error(void) { return g_err; }

int pdmp_device_alloc(int32_t device, size_t bytes, size_t chunk_bytes, void **out) {
    g_err[0] = 0;
    if (!out || bytes == 0) { snprintf(g_err, sizeof g_err, "device_alloc: null pointer or zero size"); return PDMP_ERR_ARGUMENT; }
    *out = nullptr;
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { snprintf(g_err, sizeof g_err, "device_alloc: no current device"); return PDMP_ERR_RUNTIME; }
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    const size_t gran = (size_t)2 << 20;                                    // (the runtime reports 4 KiB; whole 2-MiB pages keep the translations large)
    size_t S = chunk_bytes ? chunk_bytes : ((size_t)1 << 30);
    S = (S + gran - 1) / gran * gran;
    if (pool_enabled()) {                                                    // the same request as an allocation parked in the pool: that allocation
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto it = g_pool.begin(); it != g_pool.end(); ++it)
            if (it->second.device == dev && it->second.bytes == bytes && it->second.chunk == chunk_bytes) {
                void *p = it->first;
                it->second.set_id = 0;                                       // (handed out on its own: no longer part of a judged set)
                g_allocs[p] = it->second;
                g_pool.erase(it);
                *out = p;
                return PDMP_OK;
            }
    }
    ChunkedAlloc a;
    a.device = dev; a.bytes = bytes; a.chunk = chunk_bytes;
    for (size_t left = bytes; left > 0;) {
        const size_t want = left < S ? left : S;
        const size_t sz = (want + gran - 1) / gran * gran;
        a.sizes.push_back(sz);
        a.va_bytes += sz;
        left -= want;
    }
#define DA_TRY(expr, mapped)                                                                                    \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            snprintf(g_err, sizeof g_err, "device_alloc: %s: %s", #expr, hipGetErrorString(e_));                \
            (void)hipGetLastError();                                                                            \
            release_all(base, a, (mapped));                                                                     \
            return PDMP_ERR_RUNTIME;                                                                            \
        }                                                                                                       \
    } while (0)
    char *base = nullptr;
    DA_TRY(hipMemAddressReserve((void **)&base, a.va_bytes, (S & (S - 1)) == 0 ? S : gran, nullptr, 0), 0);
    size_t off = 0;
    for (size_t i = 0; i < a.sizes.size(); ++i) {
        hipMemGenericAllocationHandle_t h;
        hipError_t ce = hipMemCreate(&h, a.sizes[i], &prop, 0);
        if (ce != hipSuccess) {                                              // out of memory with allocations parked in the pool: release them, once more
            (void)hipGetLastError();
            bool trimmed = false;
            { std::lock_guard<std::mutex> lk(g_mu); trimmed = !g_pool.empty(); if (trimmed) { (void)hipDeviceSynchronize(); release_pool_locked(); } }
            if (trimmed) ce = hipMemCreate(&h, a.sizes[i], &prop, 0);
        }
        DA_TRY(ce, i);
        a.handles.push_back(h);
        DA_TRY(hipMemMap(base + off, a.sizes[i], 0, h, 0), i);
        off += a.sizes[i];
    }
    hipMemAccessDesc desc;
    desc.location.type = hipMemLocationTypeDevice; desc.location.id = dev; desc.flags = hipMemAccessFlagsProtReadWrite;
    DA_TRY(hipMemSetAccess(base, a.va_bytes, &desc, 1), a.handles.size());
    {                                                                      // (the flush belongs to the device the memory lives on)
        int cur = dev;
        (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        const bool flushed = flush_translations();
        if (cur != dev) (void)hipSetDevice(cur);
        if (!flushed) {                                                     // (memory so full that not even 4 MiB could be allocated and freed)
            snprintf(g_err, sizeof g_err, "device_alloc: the translation caches could not be flushed after mapping (no memory left for the 4 MiB trial allocation)");
            release_all(base, a, a.handles.size());
            return PDMP_ERR_RUNTIME;
        }
    }
#undef DA_TRY
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_allocs[base] = a;
    }
    *out = base;
    return PDMP_OK;
}

// ---- the streamed arrays of an ensemble, drawn several times (include/pdmp_hip.h: pdmp_device_alloc_streams)
namespace {
// The launch's store pattern without its arithmetic: a wave owns 64 consecutive chains and writes, step by step, ONE 64-byte line of each
// chain's row in A and in B (eight lanes per line, eight chains per store instruction, streaming stores); chains `stride` doubles apart;
// batches of 64 chains handed out by an atomic counter to `gridDim.x` persistent waves.  Correlates with the lane kernel's launch time
// over placements to a few per cent (profiles/r4/placement: synth2_ms against engine_ms).
__global__ __launch_bounds__(64) void stream_pair_writer(double *A, double *B, long long stride, long long n_chains, int lines, int *queue, int n_batches) {
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
                const long long c = (long long)b * 64 + 8 * r + (lane >> 3);
                if (c < n_chains) {
                    const long long off = c * stride + (long long)l * 8 + (lane & 7);
                    const double val = (double)(l + lane);
                    __builtin_nontemporal_store(val, A + off);
                    __builtin_nontemporal_store(-val, B + off);
                }
            }
        }
    }
}
// minimum of two timed passes after one untimed pass; < 0 on an error
float time_stream_pair(void *A, void *B, size_t bytes, size_t stride_bytes) {
    const long long stride = (long long)(stride_bytes / 8), chains = (long long)(bytes / stride_bytes);
    const int lines = (int)(stride_bytes / 64);
    if (chains < 1 || lines < 1) return -1.f;
    int *queue = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    float best = -1.f;
    if (hipMalloc(&queue, sizeof(int)) != hipSuccess) return -1.f;
    if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) {
        const int nb = (int)((chains + 63) / 64);
        for (int i = 0; i < 3; ++i) {
            if (hipMemsetAsync(queue, 0, sizeof(int), 0) != hipSuccess || hipEventRecord(a, 0) != hipSuccess) { best = -1.f; break; }
            hipLaunchKernelGGL(stream_pair_writer, dim3(1024), dim3(64), 0, 0, (double *)A, (double *)B, stride, chains, lines, queue, nb);
            float ms = 0.f;
            if (hipEventRecord(b, 0) != hipSuccess || hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) { best = -1.f; break; }
            if (i >= 1 && (best < 0.f || ms < best)) best = ms;
        }
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(queue);
    (void)hipGetLastError();
    return best;
}
}  // namespace

int pdmp_device_alloc_streams(int32_t device, int32_t n, const size_t *bytes, size_t chunk_bytes, size_t stride_bytes, int32_t draws,
                              void **out_ptrs, float *draw_ms, int32_t *n_drawn) {
    g_err[0] = 0;
    if (!bytes || !out_ptrs || n < 1 || n > 16) { snprintf(g_err, sizeof g_err, "device_alloc_streams: null pointer or array count outside 1..16"); return PDMP_ERR_ARGUMENT; }
    if (n_drawn) *n_drawn = 0;
    size_t total = 0;
    for (int i = 0; i < n; ++i) { if (!bytes[i]) { snprintf(g_err, sizeof g_err, "device_alloc_streams: zero size"); return PDMP_ERR_ARGUMENT; } total += bytes[i]; }
    const bool measure = draws > 1 && n >= 2 && stride_bytes >= 64 && stride_bytes % 64 == 0 && bytes[0] == bytes[1] && bytes[0] >= stride_bytes;
    int dev = device, cur = 0;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { snprintf(g_err, sizeof g_err, "device_alloc_streams: no current device"); return PDMP_ERR_RUNTIME; }
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(dev);
    std::vector<void *> best(n, nullptr), cand(n, nullptr);
    float best_ms = -1.f;
    int rc = PDMP_OK, drawn = 0;
    if (pool_enabled()) {               // a set that was judged and kept before, then freed: all n of its arrays wait in the pool -> the same set again, nothing drawn
        std::lock_guard<std::mutex> lk(g_mu);
        std::map<long long, std::vector<void *>> sets;
        for (auto &kv : g_pool) {
            const ChunkedAlloc &q = kv.second;
            if (q.set_id > 0 && q.device == dev && q.set_index < n && q.bytes == bytes[q.set_index] && q.chunk == chunk_bytes) {
                auto &v = sets[q.set_id];
                if (v.empty()) v.assign(n, nullptr);
                v[q.set_index] = kv.first;
            }
        }
        for (auto &sv : sets) {
            bool whole = true;
            for (void *p : sv.second) whole = whole && p != nullptr;
            if (!whole) continue;
            for (int i = 0; i < n; ++i) { g_allocs[sv.second[i]] = g_pool[sv.second[i]]; g_pool.erase(sv.second[i]); out_ptrs[i] = sv.second[i]; }
            (void)hipSetDevice(cur);
            if (n_drawn) *n_drawn = 0;                                       // (0 draws: the judged set of an earlier call)
            return PDMP_OK;
        }
    }
    auto free_set = [&](std::vector<void *> &v) { g_release_now = true; for (auto &p : v) { if (p) (void)pdmp_device_free(p); p = nullptr; } g_release_now = false; };
    for (int k = 0; k < (measure ? draws : 1); ++k) {
        if (k > 0) {                                                        // another set must fit beside the best one so far
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < total + total / 8 + ((size_t)2 << 30)) break;
        }
        int e = PDMP_OK;
        for (int i = 0; i < n && e == PDMP_OK; ++i) e = pdmp_device_alloc(dev, bytes[i], chunk_bytes, &cand[i]);
        if (e != PDMP_OK) {                                                 // (pdmp_device_alloc has set the message)
            char keep[sizeof g_err]; memcpy(keep, g_err, sizeof keep);
            free_set(cand);
            if (k == 0) { memcpy(g_err, keep, sizeof keep); rc = e; }
            break;
        }
        float ms = -1.f;
        if (measure) {
            ms = time_stream_pair(cand[0], cand[1], bytes[0], stride_bytes);
            if (draw_ms) draw_ms[k] = ms;
        }
        ++drawn;
        if (!best[0] || (ms >= 0.f && (best_ms < 0.f || ms < best_ms))) { free_set(best); best.swap(cand); best_ms = ms; }
        else free_set(cand);
        if (!measure) break;
    }
    (void)hipSetDevice(cur);
    if (rc != PDMP_OK) return rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const long long id = g_next_set++;
        for (int i = 0; i < n; ++i) { auto it = g_allocs.find(best[i]); if (it != g_allocs.end()) { it->second.set_id = id; it->second.set_index = i; } }
    }
    for (int i = 0; i < n; ++i) out_ptrs[i] = best[i];
    if (n_drawn) *n_drawn = drawn;
    return PDMP_OK;
}

int pdmp_device_free(void *ptr) {
    g_err[0] = 0;
    if (!ptr) return PDMP_OK;
    ChunkedAlloc a;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_allocs.find(ptr);
        if (it == g_allocs.end()) { snprintf(g_err, sizeof g_err, "device_free: not a pointer of pdmp_device_alloc"); return PDMP_ERR_ARGUMENT; }
        a = it->second;
        g_allocs.erase(it);
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(a.device);
    (void)hipDeviceSynchronize();                                           // no kernel may still be writing through the mapping
    if (pool_enabled() && !g_release_now) {                                 // parked: address range and mapping intact, for the next request of this size
        std::lock_guard<std::mutex> lk(g_mu);
        g_pool[ptr] = a;
    } else {
        release_all((char *)ptr, a, a.handles.size());
        (void)flush_translations();                                         // (no translation of the released memory may outlive it)
    }
    (void)hipSetDevice(cur);
    return PDMP_OK;
}

int pdmp_device_pool_trim(void) {
    g_err[0] = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_pool) { (void)hipSetDevice(kv.second.device); (void)hipDeviceSynchronize(); }
    release_pool_locked();
    (void)flush_translations();
    (void)hipSetDevice(cur);
    return PDMP_OK;
}

}  // extern "C"
