// device_alloc.hip -- output arrays of an ensemble built from independent physical chunks (HIP virtual memory management).
//
// Why (profiles/r4/placement_*): a 40 GB array from hipMalloc is backed by a few huge physical blocks, and the
// launch that streams the event rows of X and V side by side runs 14.4 ms or 18 ms depending on how those blocks happen to lie relative
// to each other in the memory's channel / bank map (write requests then stall on DRAM credits 3 - 8 x as often: TCC_EA0_WRREQ_DRAM_
// CREDIT_STALL 2.6e8 against 3 - 8e7 per launch; translation misses unchanged) -- one draw for the whole array, fast or slow for as long
// as the allocation lives.  Built from 1-GiB chunks, every (X chunk, V chunk) pair draws on its own (3 % of the pairs are 6 - 13 % slower),
// so the launch time concentrates: 14.4 - 14.75 ms in every one of thirty pools and mappings measured, no slow mode.  The chunks must not
// be small either: below ~32 MiB of physically contiguous memory per translation the kernel's 131 000 concurrent write streams miss in the
// translation caches (TCP_UTCL1_TRANSLATION_MISS 1.2e5 -> 7e7 .. 1.3e8 per launch; 16.0 ms at 16 MiB, 18.9 ms at 4 MiB, 22.5 ms at 2 MiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/pdmp_hip.h"

namespace {

struct ChunkedAlloc {
    int device = 0;
    size_t va_bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
};
std::mutex g_mu;
std::map<void *, ChunkedAlloc> g_allocs;

thread_local char g_err[256];

void release_all(char *base, ChunkedAlloc &a, size_t mapped) {
    size_t off = 0;
    for (size_t i = 0; i < a.handles.size(); ++i) {
        if (i < mapped) (void)hipMemUnmap(base + off, a.sizes[i]);
        (void)hipMemRelease(a.handles[i]);
        off += a.sizes[i];
    }
    if (base) (void)hipMemAddressFree(base, a.va_bytes);
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
    ChunkedAlloc a;
    a.device = dev;
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
        DA_TRY(hipMemCreate(&h, a.sizes[i], &prop, 0), i);
        a.handles.push_back(h);
        DA_TRY(hipMemMap(base + off, a.sizes[i], 0, h, 0), i);
        off += a.sizes[i];
    }
    hipMemAccessDesc desc;
    desc.location.type = hipMemLocationTypeDevice; desc.location.id = dev; desc.flags = hipMemAccessFlagsProtReadWrite;
    DA_TRY(hipMemSetAccess(base, a.va_bytes, &desc, 1), a.handles.size());
#undef DA_TRY
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_allocs[base] = a;
    }
    *out = base;
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
    release_all((char *)ptr, a, a.handles.size());
    (void)hipSetDevice(cur);
    return PDMP_OK;
}

}  // extern "C"
