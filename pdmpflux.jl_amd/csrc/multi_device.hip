// multi_device.hip -- pdmp_multi_*: one host process, n GPUs (include/pdmp_hip.h).
// The ensemble's chains are independent (the reference's `seed = 42 + i` loop, test/test_property_based.jl:138), so devices
// take contiguous chain ranges, sample them with the global chain ids on their own streams and exchange nothing until the end;
// the optional final exchange is ONE collective per output array over RCCL (xGMI): an in-place ncclAllGather when the shards are
// equal, else one grouped ncclBroadcast per shard (all-gather-v).  RCCL is bound at run time (dlopen of librccl.so.1 -- the
// copy a PyTorch-ROCm process has already loaded, or /opt/rocm's): a single-GPU user never needs the library present.
#include "../../include/pdmp_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) int pdmp_set_error_message(int code, const char *msg);   // engine_api.hip: thread-local last error

namespace {

int failf(int code, const char *fmt, const char *a = "", const char *b = "") {
    char buf[1024];
    snprintf(buf, sizeof buf, fmt, a, b);
    return pdmp_set_error_message(code, buf);
}
#define M_TRY(expr)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return failf(PDMP_ERR_RUNTIME, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- the slice of the RCCL API this file uses (rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer, ncclFloat64 = 8)
typedef void *rcclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(rcclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kRcclFloat64 = 8;   // ncclFloat64 / ncclDouble, rccl.h
Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.lib) return PDMP_OK;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return failf(PDMP_ERR_RUNTIME, "cannot load librccl.so.1 (needed for n_devices > 1 with gather): %s", dlerror());
    Rccl r;
    r.lib = lib;
#define BIND(field, sym) *(void **)(&r.field) = dlsym(lib, sym); if (!r.field) return failf(PDMP_ERR_RUNTIME, "librccl.so.1 lacks %s", sym)
    BIND(CommInitAll, "ncclCommInitAll");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(AllGather, "ncclAllGather");
    BIND(Broadcast, "ncclBroadcast");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    g_rccl = r;
    return PDMP_OK;
}
#define R_TRY(expr)                                                                              \
    do {                                                                                         \
        int r_ = (expr);                                                                         \
        if (r_ != 0) return failf(PDMP_ERR_RUNTIME, "%s: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

}  // namespace

struct pdmp_multi {
    int n = 0;
    std::vector<int> dev;
    std::vector<pdmp_handle> h;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> e0, e1, e2;          // launch begin / kernel end / exchange end, per device
    std::vector<double *> dx, dv;                // per-device staging of the shard's xinit, vinit
    std::vector<size_t> init_cap;
    std::vector<rcclComm_t> comm;                // empty until the first gathering call with n > 1
    double kernel_ms = 0.0, exchange_ms = 0.0;
    bool force_rccl = false;                     // PDMP_MULTI_FORCE_RCCL at pdmp_multi_create (test hook: the collective with one device too)
};

extern "C" {

int pdmp_multi_shard_range(int64_t C, int32_t rank, int32_t n_ranks, int64_t *lo, int64_t *hi) {
    if (C < 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks || !lo || !hi) return failf(PDMP_ERR_ARGUMENT, "pdmp_multi_shard_range: bad argument");
    const int64_t base = C / n_ranks, rem = C % n_ranks;
    *lo = (int64_t)rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return PDMP_OK;
}

int pdmp_multi_destroy(pdmp_multi_handle m) {
    if (!m) return PDMP_OK;
    const int nd = (int)m->dev.size();               // (not m->n: a create that failed while validating the device list has fewer)
    for (int r = 0; r < nd; ++r) {
        (void)hipSetDevice(m->dev[r]);
        if (r < (int)m->comm.size() && m->comm[r] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm[r]);
        if (r < (int)m->h.size() && m->h[r]) (void)pdmp_destroy(m->h[r]);
        if (r < (int)m->stream.size() && m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
        if (r < (int)m->e0.size()) { (void)hipEventDestroy(m->e0[r]); (void)hipEventDestroy(m->e1[r]); (void)hipEventDestroy(m->e2[r]); }
        if (r < (int)m->dx.size()) { (void)hipFree(m->dx[r]); (void)hipFree(m->dv[r]); }
    }
    delete m;
    return PDMP_OK;
}

int pdmp_multi_create(const pdmp_config_t *cfg, int32_t n_devices, const int32_t *devices, pdmp_multi_handle *out) {
    if (!cfg || !out) return failf(PDMP_ERR_ARGUMENT, "null argument");
    *out = nullptr;
    const int visible = pdmp_device_count();
    if (n_devices <= 0 || n_devices > visible) {
        char a[32], b[32];
        snprintf(a, sizeof a, "%d", (int)n_devices); snprintf(b, sizeof b, "%d", visible);
        return failf(PDMP_ERR_ARGUMENT, "n_devices must be in 1..%s visible devices. Current value: %s", b, a);
    }
    pdmp_multi *m = new pdmp_multi();
    m->force_rccl = getenv("PDMP_MULTI_FORCE_RCCL") != nullptr;     // (policy of the handle: read once, here)
    for (int r = 0; r < n_devices; ++r) {
        const int d = devices ? devices[r] : r;
        for (int q = 0; q < r; ++q) if (m->dev[q] == d) { pdmp_multi_destroy(m); return failf(PDMP_ERR_ARGUMENT, "duplicate device in the device list"); }
        if (d < 0 || d >= visible) { pdmp_multi_destroy(m); return failf(PDMP_ERR_ARGUMENT, "device ordinal out of range"); }
        m->dev.push_back(d);
    }
    m->n = n_devices;                                // only now: the device list is complete and valid
    m->h.assign(n_devices, nullptr); m->stream.assign(n_devices, nullptr);
    m->dx.assign(n_devices, nullptr); m->dv.assign(n_devices, nullptr); m->init_cap.assign(n_devices, 0);
    for (int r = 0; r < n_devices; ++r) {
        pdmp_config_t c = *cfg;
        c.device = m->dev[r];
        if (int rc = pdmp_create(&c, &m->h[r])) { pdmp_multi_destroy(m); return rc; }
        hipEvent_t a, b, e;
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess || hipEventCreate(&e) != hipSuccess) {
            pdmp_multi_destroy(m);
            return failf(PDMP_ERR_RUNTIME, "pdmp_multi_create: stream / event creation failed");
        }
        m->e0.push_back(a); m->e1.push_back(b); m->e2.push_back(e);
    }
    *out = m;
    return PDMP_OK;
}

}  // extern "C"

// The sharded step.  Initial states either as ONE pair of host arrays (xh, vh: d x C, each shard staged to its device) or as
// per-device DEVICE arrays of the shards themselves (xd[r], vd[r]: d x len_r on device r -- nothing crosses PCIe).
static int multi_sample(pdmp_multi_handle m, int64_t C, int64_t n_sk, int64_t chain_id_offset, const double *xh, const double *vh,
                        const double *const *xd, const double *const *vd, uint64_t seed, int32_t gather, double *const *X,
                        double *const *V, double *const *t) {
    const double *xinit = xh, *vinit = vh;
    if (!m) return failf(PDMP_ERR_ARGUMENT, "null handle");
    if (n_sk <= 0) return failf(PDMP_ERR_ARGUMENT, "n_sk must be positive");                     // sample.jl:262-264
    if (C < 0) return failf(PDMP_ERR_ARGUMENT, "number of chains must be non-negative");
    if (C == 0) return PDMP_OK;
    if ((!xd && (!xinit || !vinit)) || (xd && !vd) || !X || !V || !t) return failf(PDMP_ERR_ARGUMENT, "null buffer");
    pdmp_config_t cfg;
    if (int rc = pdmp_get_config(m->h[0], &cfg)) return rc;
    const size_t d = (size_t)cfg.dim;
    const int n = m->n;
    std::vector<int64_t> lo(n), hi(n);
    bool equal = true;
    for (int r = 0; r < n; ++r) {
        (void)pdmp_multi_shard_range(C, r, n, &lo[r], &hi[r]);
        if (hi[r] - lo[r] != hi[0] - lo[0]) equal = false;
        if (!X[r] || !V[r] || !t[r]) return failf(PDMP_ERR_ARGUMENT, "null per-device buffer");
        if (xd && hi[r] > lo[r] && (!xd[r] || !vd[r])) return failf(PDMP_ERR_ARGUMENT, "null per-device initial-state buffer");
    }
    // ONE layout for every shard (results must not depend on which device a chain lands on): the occupancy choice for the largest shard
    const int G = pdmp_suggest_group_lanes(m->h[0], hi[0] - lo[0]);
    for (int r = 0; r < n; ++r) if (G > 0) { if (int rc = pdmp_set_group_lanes(m->h[r], G)) return rc; }
    // (PDMP_MULTI_FORCE_RCCL: test hook -- run the collective with one device too, which exercises the run-time binding of librccl,
    // the communicator and the grouped in-place all-gather on a one-GPU box)
    const bool exchange = gather != 0 && (n > 1 || m->force_rccl);
    if (exchange && m->comm.empty()) {
        if (int rc = load_rccl()) return rc;
        m->comm.assign(n, nullptr);
        if (const int r_ = g_rccl.CommInitAll(m->comm.data(), n, m->dev.data())) {
            m->comm.clear();                         // the next gathering call tries again instead of using null communicators
            return failf(PDMP_ERR_RUNTIME, "ncclCommInitAll: %s", g_rccl.GetErrorString(r_));
        }
    }
    // 1. every device: stage its shard of the inits, launch its kernel on its stream (asynchronous)
    for (int r = 0; r < n; ++r) {
        const int64_t len = hi[r] - lo[r];
        M_TRY(hipSetDevice(m->dev[r]));
        M_TRY(hipEventRecord(m->e0[r], m->stream[r]));
        if (len > 0) {
            const size_t need = (size_t)len * d * 8;
            const double *sx = xd ? xd[r] : nullptr, *sv = xd ? vd[r] : nullptr;
            if (!xd) {
            if (need > m->init_cap[r]) {
                M_TRY(hipStreamSynchronize(m->stream[r]));
                (void)hipFree(m->dx[r]); (void)hipFree(m->dv[r]); m->dx[r] = m->dv[r] = nullptr; m->init_cap[r] = 0;
                M_TRY(hipMalloc(&m->dx[r], need));
                M_TRY(hipMalloc(&m->dv[r], need));
                m->init_cap[r] = need;
            }
            M_TRY(hipMemcpyAsync(m->dx[r], xinit + (size_t)lo[r] * d, need, hipMemcpyHostToDevice, m->stream[r]));
            M_TRY(hipMemcpyAsync(m->dv[r], vinit + (size_t)lo[r] * d, need, hipMemcpyHostToDevice, m->stream[r]));
            sx = m->dx[r]; sv = m->dv[r];
            }
            const size_t off = gather ? (size_t)lo[r] * (size_t)n_sk : 0;         // gather: the shard sits at its chain offset of the full buffer
            if (int rc = pdmp_sample_skeleton_device(m->h[r], len, n_sk, chain_id_offset + lo[r], sx, sv, seed,
                                                     X[r] + off * d, V[r] + off * d, t[r] + off, nullptr, m->stream[r]))
                return rc;
        }
        M_TRY(hipEventRecord(m->e1[r], m->stream[r]));
    }
    // 2. the final exchange: one collective per output array, all devices in one group (single-process RCCL)
    if (exchange) {
        const size_t per_chain[3] = {(size_t)n_sk * d, (size_t)n_sk * d, (size_t)n_sk};
        for (int a = 0; a < 3; ++a) {
            double *const *buf = (a == 0) ? X : (a == 1 ? V : t);
            R_TRY(g_rccl.GroupStart());
            int gerr = 0;                            // an error inside the group: close the group before returning (never leave it open)
            for (int r = 0; r < n && !gerr; ++r) {
                if (equal) {
                    const size_t cnt = (size_t)(hi[0] - lo[0]) * per_chain[a];
                    gerr = g_rccl.AllGather(buf[r] + (size_t)r * cnt, buf[r], cnt, kRcclFloat64, m->comm[r], m->stream[r]);   // in place
                } else {
                    for (int root = 0; root < n && !gerr; ++root) {
                        const size_t cnt = (size_t)(hi[root] - lo[root]) * per_chain[a];
                        if (cnt == 0) continue;
                        double *p = buf[r] + (size_t)lo[root] * per_chain[a];
                        gerr = g_rccl.Broadcast(p, p, cnt, kRcclFloat64, root, m->comm[r], m->stream[r]);
                    }
                }
            }
            const int gend = g_rccl.GroupEnd();
            if (gerr) return failf(PDMP_ERR_RUNTIME, "RCCL collective inside the group: %s", g_rccl.GetErrorString(gerr));
            R_TRY(gend);
        }
    }
    // 3. wait, fold the timings, check every device's chain status
    int rc_all = PDMP_OK;
    m->kernel_ms = 0.0; m->exchange_ms = 0.0;
    for (int r = 0; r < n; ++r) {
        M_TRY(hipSetDevice(m->dev[r]));
        M_TRY(hipEventRecord(m->e2[r], m->stream[r]));
        M_TRY(hipStreamSynchronize(m->stream[r]));
        float k = 0.f, x = 0.f;
        M_TRY(hipEventElapsedTime(&k, m->e0[r], m->e1[r]));
        M_TRY(hipEventElapsedTime(&x, m->e1[r], m->e2[r]));
        if (k > m->kernel_ms) m->kernel_ms = k;
        if (x > m->exchange_ms) m->exchange_ms = x;
        if (hi[r] > lo[r]) { const int rc = pdmp_check_status(m->h[r], nullptr); if (rc != PDMP_OK && rc_all == PDMP_OK) rc_all = rc; }
    }
    return rc_all;
}

extern "C" {

int pdmp_multi_sample_skeleton(pdmp_multi_handle m, int64_t C, int64_t n_sk, int64_t chain_id_offset, const double *xinit,
                               const double *vinit, uint64_t seed, int32_t gather, double *const *X, double *const *V,
                               double *const *t) {
    return multi_sample(m, C, n_sk, chain_id_offset, xinit, vinit, nullptr, nullptr, seed, gather, X, V, t);
}

int pdmp_multi_sample_skeleton_device(pdmp_multi_handle m, int64_t C, int64_t n_sk, int64_t chain_id_offset, const double *const *xinit,
                                      const double *const *vinit, uint64_t seed, int32_t gather, double *const *X, double *const *V,
                                      double *const *t) {
    if (!xinit || !vinit) return failf(PDMP_ERR_ARGUMENT, "null buffer");
    return multi_sample(m, C, n_sk, chain_id_offset, nullptr, nullptr, xinit, vinit, seed, gather, X, V, t);
}

int pdmp_multi_init_states(pdmp_multi_handle m, int64_t C, int64_t chain_id_offset, uint64_t seed, double x_mean, double x_scale,
                           int32_t v_kind, double *const *xinit, double *const *vinit) {
    if (!m || !xinit || !vinit) return failf(PDMP_ERR_ARGUMENT, "null argument");
    if (C <= 0) return failf(PDMP_ERR_ARGUMENT, "C must be positive");
    for (int r = 0; r < m->n; ++r) {
        int64_t lo, hi;
        (void)pdmp_multi_shard_range(C, r, m->n, &lo, &hi);
        if (hi <= lo) continue;
        if (!xinit[r] || !vinit[r]) return failf(PDMP_ERR_ARGUMENT, "null per-device initial-state buffer");
        M_TRY(hipSetDevice(m->dev[r]));
        if (int rc = pdmp_init_states_device(m->h[r], hi - lo, chain_id_offset + lo, seed, x_mean, x_scale, v_kind, xinit[r], vinit[r], m->stream[r]))
            return rc;
    }
    for (int r = 0; r < m->n; ++r) { M_TRY(hipSetDevice(m->dev[r])); M_TRY(hipStreamSynchronize(m->stream[r])); }
    return PDMP_OK;
}

int pdmp_multi_get_timing(pdmp_multi_handle m, double *kernel_ms, double *exchange_ms) {
    if (!m) return failf(PDMP_ERR_ARGUMENT, "null handle");
    if (kernel_ms) *kernel_ms = m->kernel_ms;
    if (exchange_ms) *exchange_ms = m->exchange_ms;
    return PDMP_OK;
}

}  // extern "C"
