// engine_api.hip -- C ABI of libpdmphip.so (include/pdmp_hip.h): handle management, argument checks
// that mirror the reference constructors, kernel dispatch, sample_from_skeleton kernel, hipEvent timing.
#include "../../include/pdmp_hip.h"
#include "group_kernel.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace pdmp;
PDMP_NS_BEGIN
hipError_t launch_lane_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);   // lane_kernel.hpp: d = 49 .. 52
hipError_t launch_lane_zigzag_a(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream); // d = 25 .. 36
hipError_t launch_lane_zigzag_b(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream); // d = 37 .. 48
hipError_t launch_lane_check(const double *vinit, int64_t n, int32_t *flag, hipStream_t stream);
hipError_t launch_lane2_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);   // a PAIR of lanes per chain: d = 97 .. 104
hipError_t launch_lane2_zigzag_a(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream); // d = 53 .. 72
hipError_t launch_lane2_zigzag_b(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream); // d = 73 .. 96
hipError_t launch_lane2q_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);  // a pair of lanes per chain at d = 25 .. 52 (mid-sized launches)
hipError_t launch_lane_zigzag_c(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);   // d = 9 .. 24 (round 4)
hipError_t launch_lane2q_zigzag_c(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream); // the pair form at d = 9 .. 24
hipError_t launch_lane_zigzag_d(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);   // d = 1 .. 8 (round 4: rows shorter than a line)
hipError_t launch_lane2q_zigzag_d(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream); // the pair form at d = 1 .. 8
hipError_t launch_group_reg_zigzag(int target, bool fd, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_suzz_exact(int target, const GroupParams &gp, hipStream_t stream);        // round 4: SpeedUpZigZag, the slot counts that do not spill (inst_reg_suzz_k*.hip)
hipError_t launch_group_reg_sticky_exact(int target, const GroupParams &gp, hipStream_t stream);      // round 4: StickyZigZag, K <= 7 (inst_reg_sticky_k*.hip)
hipError_t launch_group_reg_bps(int target, bool fd, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_boomerang(int target, bool fd, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_rhmc(int target, const GroupParams &gp, hipStream_t stream);                        // inst_reg_rhmc.hip
hipError_t launch_group_reg_zigzag_exact(int target, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_bps_exact(int target, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_fecmc_exact(int target, const GroupParams &gp, hipStream_t stream);                 // inst_reg_fecmc_k*.hip
hipError_t launch_group_reg_boomerang_exact(int target, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_boomerang_exact_fd(int target, const GroupParams &gp, hipStream_t stream);
// tape replay (pdmp_set_tape): the same kernel forms with the tapes as random source (inst_tape_*.hip)
hipError_t launch_lane_zigzag_tape(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);
hipError_t launch_group_reg_tape_zigzag(int target, bool fd, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_tape_bps(int target, bool fd, const GroupParams &gp, hipStream_t stream);
hipError_t launch_group_reg_tape_boomerang(int target, bool fd, const GroupParams &gp, hipStream_t stream);
PDMP_NS_END
// The arithmetic contract v1 (pdmp_config_t.rng_version = 1: Philox4x32-10, the fdlibm forms without fma -- rounds 1-4 and the fixtures
// frozen then): the LDS-form group kernels and the wave-per-chain kernel compiled a second time with -DPDMP_RNG_VERSION=1 (csrc/Makefile,
// v1/*.o; inline namespace pdmp::av1), reached through these two entry points of v1_bridge.hip.  GroupParams / SkelParams are the same
// plain structs in both namespaces.
namespace pdmp {
hipError_t v1_launch_group(int sampler, int target, const void *group_params, hipStream_t stream);
hipError_t v1_launch_skeleton(int sampler, int target, const void *skel_params, hipStream_t stream);
}

namespace {

thread_local char g_err[4096] = "";
#ifdef PDMP_STAMP
unsigned long long *g_stamps = nullptr;
#endif

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(PDMP_ERR_RUNTIME, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct EventPair { hipEvent_t a, b; };

}  // namespace

extern "C" int pdmp_dense_batched_run(int device, int d, int signed_bound, int adaptive, double tmax, int64_t cap,
                                      const double *P_dev, int64_t C, int64_t n_sk, int64_t chain0, uint64_t seed,
                                      const double *xinit, const double *vinit, double *X, double *V, double *t,
                                      double *H, double *AR, int32_t *EB, double *EV, int32_t *RJ, int32_t *HH,
                                      int32_t *status, void *stream, int full_pv, int gemm_impl, char *err, int errlen);
extern "C" int pdmp_dense_incremental_run(int device, int d, int signed_bound, int adaptive, double tmax, int64_t cap,
                                      const double *P_dev, int64_t C, int64_t n_sk, int64_t chain0, uint64_t seed,
                                      const double *xinit, const double *vinit, double *X, double *V, double *t,
                                      double *H, double *AR, int32_t *EB, double *EV, int32_t *RJ, int32_t *HH,
                                      int32_t *status, void *stream, void **ws, size_t *ws_cap, int64_t resync_events, int gemm_impl,
                                      double t_stop, int resume, int64_t n_launch, const int64_t *chain_map, int64_t *count, char *err, int errlen);
extern "C" int pdmp_dense_bps_run(int device, int boomerang, int d, int signed_bound, int adaptive, int gaussian_velocity, double tmax, double refresh,
                                  int64_t cap, const double *P_dev, int64_t C, int64_t n_sk, int64_t chain0, uint64_t seed,
                                  const double *xinit, const double *vinit, double *X, double *V, double *t,
                                  double *H, double *AR, int32_t *EB, double *EV, int32_t *RJ, int32_t *HH,
                                  int32_t *status, void *stream, void **ws, size_t *ws_cap, int64_t resync_events, int gemm_impl, int lanes,
                                  double t_stop, int resume, int64_t n_launch, const int64_t *chain_map, int64_t *count, char *err, int errlen);   // dense_batched.hip: BPS, one GEMM per event

extern "C" void pdmp_dense_bind_context(void **slot);        // dense_batched.hip: the rocBLAS handles / side stream of the batched dense forms, per engine handle
extern "C" void pdmp_dense_destroy_context(void *ctx);
extern "C" int pdmp_user_compile(int device, int sampler, int fd, int kind, int rng_version, int rtgrid, const char *source, void **fn_out, char *err, int errlen);
extern "C" int pdmp_user_launch(void *fn, const GroupParams *gp, void *stream);

// Launch policy of a handle: which kernel form serves a launch, and the tuning numbers of the forms.  Filled ONCE, when the handle is
// created, from the defaults below and the PDMP_* environment overrides (test hooks and tuning scripts); no launch reads the
// environment.  pdmp_reload_policy() re-reads it for a caller that changes the environment between launches of one handle.
// The defaults were measured on one MI355X at n_sk = 1000 (profiles/r3/r3q1_*); the form never changes results
// (bit-identical kernels), only speed.
struct LaunchPolicy {
    int reg = 1;                   // PDMP_REG=0: keep x, v in LDS tiles (the LDS form) although a register form exists
    int reg_exact = 1;             // PDMP_REG_EXACT=0: the 13-slot register kernels with per-slot skips instead of the exact-K ones
    int lane_mode = -1;            // PDMP_LANE: 0 never a lane form, 1 one lane per chain whenever the configuration allows, 2 the pair form
                                   //   at d <= 52 too; -1 (unset): from the number of chains, thresholds below
    int64_t lane_min_chains = -1;  // PDMP_LANE_MIN_CHAINS: one lane per chain from this many chains on (-1: lane_from_default())
    int64_t pair_min_chains = 16385;   // PDMP_LANE_PAIR_MIN_CHAINS: up to 16 384 chains = 1 024 group-kernel waves, one per SIMD: 3.9 - 4.0 ms flat
    int lane_rounds = 24;          // PDMP_LANE_ROUNDS: work units per resident wave of a lane kernel (8 -> 24: 18.2 -> 16.8 ms at config 2)
    int cell_budget = 0;           // PDMP_CELL_BUDGET: cells of the lazy Zig-Zag bound per pass (0: the form's default: 9 / 3 group, 5 lane)
    int force_segments = 0;        // PDMP_FORCE_SEGMENTS: segments per chain (0: from the schedule); test hook for the hand-off
    int dense_mode = 0;            // PDMP_DENSE_MODE: 0 default (incr for >= 64 chains and d <= 1024, gemm beyond), 1 incr, 2 gemm, 3 matvec
    int tape_wave = 0;             // PDMP_TAPE_WAVE=1: tape replay on the wave-per-chain kernel
    int from_skeleton_rows = 0;    // PDMP_FROM_SKELETON_ROWS: the row-per-iteration sample_from_skeleton kernel (A/B runs)
    int64_t dense_resync = 0;      // PDMP_DENSE_RESYNC: events between two re-basing products of the batched dense forms (0: 256 Zig-Zag / 64 BPS, Boomerang)
    int dense_full_pv = 0;         // PDMP_DENSE_FULL_PV: the GEMM-per-phase form multiplies the velocity half in every phase
    int dense_gemm = -1;           // PDMP_DENSE_GEMM=mfma | rocblas: the product P [x | v] of the batched dense forms by the hand-written fp64 MFMA kernel
                                   //   (csrc/dgemm_mfma.hip) or by rocBLAS.  -1 (unset): the hand-written kernel for the Zig-Zag forms (one product of 2 C
                                   //   columns per 256 events: 57 - 58 TFLOP/s, on a par with the library), rocBLAS for BPS / Boomerang, whose per-event
                                   //   products of C / 2 columns in two concurrent lanes it serves better (65 against 53 TFLOP/s at 5 000 columns, and its
                                   //   kernel overlaps with the other lane's step kernel where ours does not: 92.8 against 108 ms, profiles/r4)
    int dense_lanes = 2;           // PDMP_DENSE_LANES=1 | 2: BPS / Boomerang on a dense precision as ONE ensemble, or as two half-ensembles on two
                                   //   streams (the product of one half beside the step kernel of the other)
    int no_lean = 0;               // PDMP_NO_LEAN: the full 13-slot register kernels even where a lean throughput instantiation exists (A/B, tests)
    int debug_launch = 0;          // PDMP_DEBUG_LAUNCH: print the launch geometry of the lane kernels
    int lane_debug = 0;            // PDMP_LANE_DEBUG (measurements only, the results are WRONG): 1 = the lane kernels without their stores of t, 2 = of V, 4 = of X
    int dense_debug = 0;           // PDMP_DENSE_DEBUG (measurements only, the results are WRONG): 1 = the incremental dense form's segment kernel without its row
                                   //   stores, 2 = without its loads of P's columns -- what a launch costs beside them (bench.py --ab-env)
    // one lane per chain from this many chains on: the pair form (GL = 2) at d = 53 .. 104 fills the machine at 32 768 chains; at d <= 52
    // the one-lane kernel takes 8.2 - 9.7 ms whatever the count up to 65 536 chains and overtakes the pair form at ~37 000 (d >= 45) / ~40 000
    // (d <= 52: the pair form is the best form exactly while its waves -- 32 chains each -- are alone on their SIMDs: 32 768 chains; measured at the end of
    //  round 4, d = 16 / 32 / 50, 33 500 .. 39 000 chains: pair 6.2 / 7.0-7.2 / 8.4-8.7 ms, one lane 5.45 / 6.6 / 8.2-8.3 -- scripts/r4_lane_from.sh; before: 37 000 / 40 000)
    // (round 5, the history length: the lane forms write 64 row streams per wave, n_sk d 8 bytes apart, and slow down faster than linearly as those
    //  grow -- n_sk = 2000 / 4000 at 33 000 chains: group kernel 15.4 / 32.3 ms, one lane 19.2 / 41.5, pair 18.1 / 37.1; from 50 000 chains on one lane
    //  per chain wins again (18.5 / 39.9 against 23.9 / 46.5) -- scripts/r5_thresholds.sh, profiles/r5/thresholds_nsk.jsonl: from n_sk = 1500 on
    //  the one-lane form starts at 45 000 chains, and the pair form never runs beyond the 32 768 chains its waves are alone on their SIMDs with)
    int64_t lane_from_default(bool pair_dims, int dim, int64_t n_sk) const { (void)dim; return pair_dims ? 28000 : (n_sk >= 1500 ? 45000 : 32769); }
};

static LaunchPolicy policy_from_env() {
    LaunchPolicy q;
    auto geti = [](const char *name, long long dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; };
    q.reg = (int)geti("PDMP_REG", 1) != 0;
    q.reg_exact = (int)geti("PDMP_REG_EXACT", 1) != 0;
    q.lane_mode = (int)geti("PDMP_LANE", -1);
    q.lane_min_chains = geti("PDMP_LANE_MIN_CHAINS", -1);
    q.pair_min_chains = geti("PDMP_LANE_PAIR_MIN_CHAINS", q.pair_min_chains);
    { const int v = (int)geti("PDMP_LANE_ROUNDS", q.lane_rounds); if (v >= 1 && v <= 64) q.lane_rounds = v; }
    { const int b = (int)geti("PDMP_CELL_BUDGET", 0); if (b >= 1 && b <= 9) q.cell_budget = b; }
    q.force_segments = (int)geti("PDMP_FORCE_SEGMENTS", 0);
    if (const char *m = getenv("PDMP_DENSE_MODE")) q.dense_mode = !strcmp(m, "incr") ? 1 : !strcmp(m, "gemm") ? 2 : !strcmp(m, "matvec") ? 3 : 0;
    q.tape_wave = (int)geti("PDMP_TAPE_WAVE", 0) == 1;
    q.from_skeleton_rows = getenv("PDMP_FROM_SKELETON_ROWS") != nullptr;
    { const long long r = geti("PDMP_DENSE_RESYNC", 0); if (r >= 1) q.dense_resync = r; }
    q.dense_full_pv = getenv("PDMP_DENSE_FULL_PV") != nullptr;
    if (const char *m = getenv("PDMP_DENSE_GEMM")) q.dense_gemm = !strcmp(m, "mfma") ? 1 : (!strcmp(m, "rocblas") ? 0 : -1);
    { const int v = (int)geti("PDMP_DENSE_LANES", q.dense_lanes); if (v == 1 || v == 2) q.dense_lanes = v; }
    q.no_lean = getenv("PDMP_NO_LEAN") != nullptr;
    q.debug_launch = getenv("PDMP_DEBUG_LAUNCH") != nullptr;
    { const int v = (int)geti("PDMP_DENSE_DEBUG", 0); if (v >= 0 && v <= 3) q.dense_debug = v; }
    { const int v = (int)geti("PDMP_LANE_DEBUG", 0); if (v >= 0 && v <= 7) q.lane_debug = v; }
    if (q.lane_debug || q.dense_debug)      // (never silently: these knobs drop stores)
        fprintf(stderr, "pdmp: PDMP_LANE_DEBUG=%d PDMP_DENSE_DEBUG=%d -- measurement knobs: stores / loads are dropped, the results of this handle are WRONG\n", q.lane_debug, q.dense_debug);
    return q;
}

struct pdmp_engine {
    pdmp_config_t cfg;
    LaunchPolicy policy;
    void *user_fn = nullptr;          // PDMP_T_USER: the runtime-compiled group kernel (user_jit.hip)
    int group_lanes = 0;              // 0: wave-per-chain kernel; else lanes per chain of the group kernel
    int last_form = 0;                // PDMP_FORM_* of the last launch (PDMP_FORM_LANE: provided the unit-speed flag stayed 0)
    int lane_fallback = PDMP_FORM_GROUP_REG;   // the group kernel queued behind a lane kernel (it runs when the unit-speed check fails): register or LDS form
    int device = 0;
    double *d_tparams = nullptr;
    int32_t *d_status = nullptr;      // [0] itercap chains, [1] guard events, [2] lost hand-offs, [3] spare
    int32_t *d_sched = nullptr;       // group kernel: [0] unit counter, [1..nb] per-batch progress
    size_t sched_cap = 0;
    double *d_state = nullptr;        // group kernel: per-chain hand-off state (2d+4 doubles)
    int tflags = 0;                   // SkelParams::tflags: bit 0 = diagonal Gaussian with every precision >= 0 (looked at once, when the handle is made)
    void *dense_ctx = nullptr;        // batched dense forms: this handle's rocBLAS handles, side stream and events (dense_batched.hip: DenseCtx)
    void *dense_ws = nullptr;         // incremental dense form: [x | v], [P x | P v], per-chain scalars (dense_batched.hip)
    size_t dense_ws_cap = 0;
    size_t state_cap = 0;
    double *d_tape[3] = {nullptr, nullptr, nullptr};   // tape source: exp, uni, nrm (chain-major)
    int64_t tape_n[3] = {0, 0, 0}, tape_chains = 0;
    hipStream_t last_stream = nullptr;
    bool until_valid = false;         // d_state holds the hand-off states of a first chunk of `until_chains` chains (time-horizon variant)
    int64_t until_chains = 0;
    uint64_t until_seed = 0;          // ... of the run (seed, first global chain id, lanes per chain) a resumed chunk must belong to
    int64_t until_chain0 = 0;
    int until_lanes = 0;
    bool launched = false;            // a launch has used the per-handle scheduler / hand-off / status buffers
    hipEvent_t order_ev = nullptr;    // orders launches of this handle that arrive on different streams (they share those buffers)
    std::vector<EventPair> pending;   // launches not yet folded into `timing`
    std::vector<EventPair> pool;
    pdmp_timing_t timing = {0.0, 0.0, 0};
};

namespace {

// ---- flows of the samplers whose flow is not x + v t / a rotation, in the wave layout (lane owns coordinates lane, lane + 64, ...);
// shared by sample_from_skeleton, the fused moments and the time-horizon cut.
// SpeedUpZigZag (SpeedUpZigZagSamplers.jl:71-79): x_t = (x - av v) + kv v; dot products in the wave order of the checker.
__device__ __forceinline__ void suzz_scalars(int d, int lane, const double *x0, const double *v0, double ta, double &av, double &kv, int rv) {   // rv: arithmetic contract of exp
    const double x1 = x0[0], v1 = v0[0], dd = (double)d;
    av = v1 * x1;
    double yv = 0.0, yy = 0.0;
    for (int q = lane; q < d; q += WAVE) { const double y = x0[q] - av * v0[q]; yv += y * v0[q]; yy += y * y; }
    yv = wave_sum(yv);
    yy = wave_sum(yy);
    const double c = v1 * yv, a = (1.0 + yy) / dd - (c * c) / (dd * dd), Y0 = x1 + c / dd;
    const double ea = __dsqrt_rn(dd) * v1 * ta;
    const double bt = (Y0 + __dsqrt_rn(Y0 * Y0 + a)) * (rv == 1 ? det_exp_v1(ea) : det_exp_v2(ea));
    const double X1 = (bt * bt - a) / (2.0 * bt) - c / dd;
    kv = v1 * X1;
}
// RHMC (RandomizedHamiltonianMonteCarlo.jl:97-130): velocity-Verlet from (x0, v0) by ta; up to 4 coordinates per lane (d <= 256,
// checked by the host), built-in targets.
__device__ __forceinline__ void rhmc_integrate(int d, int lane, const double *x0, const double *v0, double ta, double h, int target,
                                               const double *tparams, double (&xq)[4], double (&vq)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int q = lane + k * WAVE; xq[k] = (q < d) ? x0[q] : 0.0; vq[k] = (q < d) ? v0[q] : 0.0; }
    auto half_kick = [&](double hh) {                                 // v <- v - (0.5 hh) grad-U(x)
        double g0 = 0.0, g1 = 0.0;
        if (target == PDMP_T_BANANA && d >= 2) {                      // test/test_config.jl:33-36, as GTarget<T_BANANA>
            const double b0 = readlane_d(xq[0], 0), b1 = readlane_d(xq[0], 1);
            const double sq = b0 * b0, m = sq - 1.0, r = b1 - m, a2 = 2.0 * b0, pr = a2 * r;
            g0 = b0 - pr; g1 = r;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = lane + k * WAVE;
            if (q < d) {
                double gq = xq[k];
                if (target == PDMP_T_GAUSS_DIAG) gq = tparams[d + q] * (xq[k] - tparams[q]);
                else if (target == PDMP_T_BANANA && d >= 2 && q < 2) gq = (q == 0) ? g0 : g1;
                vq[k] = vq[k] - (0.5 * hh) * gq;
            }
        }
    };
    const long long n = (ta == 0.0) ? 0 : (long long)floor(ta / h);
    const double r = ta - (double)n * h;
    const long long steps = (ta == 0.0) ? 0 : n + ((r > 0.0) ? 1 : 0);
    for (long long k2 = 0; k2 < steps; ++k2) {
        const double hh = (k2 < n) ? h : r;
        half_kick(hh);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (lane + k * WAVE < d) xq[k] = xq[k] + hh * vq[k];
        half_kick(hh);
    }
}


// ---------------------------------------------------------------- sample_from_skeleton, sample.jl:475-513
// One wave per (chain, tile of 64 output samples): lane s finds the skeleton segment of sample
// tile*64+s by binary search in t (equivalent to the reference's monotone pointer walk :493-495 for
// non-decreasing t), then the wave writes the 64 samples one after the other, lanes over coordinates
// (coalesced d*8-byte rows).
// ia_mode: 0 = velocities as recorded; 1 = StickyPDMP methods (sample.jl:516-561, :588-596 with use_active): v_used = active ? v : 0.0;
// 2 = the (N, dt) method (:649-682): v0 .* active (a frozen negative velocity becomes -0.0, as in the reference).
// strict: the (N, dt) method's previous index searchsortedfirst(t, tm) - 1 = last i with t[i] < tm (clamped to the first column)
// instead of the pointer walk's last i with t[i] <= tm (:493-495).  n_use <= n_sk: only the first n_use columns of every chain
// are history (the (N, dt) method's views 1:N, :650-653; ragged histories of the time-horizon variant pass count[c] via `lens`).
// counts != nullptr (dt methods on ensembles): samples of chain c = floor(t_end / dt), at most N; the others are left untouched.
template <int BOOMERANG>   // 0: linear flow; 1 / 2: the Boomerang's rotation in arithmetic contract v2 / v1 (pdmp_config_t.rng_version)
__global__ __launch_bounds__(256) void from_skeleton_kernel(int d, int64_t C, int64_t n_sk, const double *X,
                                                            const double *V, const double *t, int64_t N,
                                                            double dt_in, int discard_vt, double *out, int suzz = 0,
                                                            int rhmc_target = -1, double rhmc_h = 0.0, const double *tparams = nullptr,
                                                            const uint8_t *IA = nullptr, int ia_mode = 0, int strict = 0,
                                                            int64_t n_use = 0, const int64_t *lens = nullptr, int64_t *counts = nullptr, int arv = 2) {
    const int lane = lane_id();
    const int64_t tiles = (N + WAVE - 1) / WAVE;
    const int64_t w = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w >= C * tiles) return;
    const int64_t c = w / tiles, tile = w % tiles;
    const double *tc = t + (size_t)c * n_sk;
    int64_t nh = (n_use > 0 && n_use < n_sk) ? n_use : n_sk;          // history columns of this chain
    if (lens) { const int64_t l = lens[c]; nh = (l > 0 && l < nh) ? l : nh; }
    const double dt = dt_in > 0.0 ? dt_in : tc[nh - 1] / (double)N;   // :483-484; dt_in: the dt::Float64 methods, :573-646, :649-682
    int64_t nmax = N;
    if (dt_in > 0.0 && counts) {                                      // n_skel = Int(floor(T_end / dt)), :608
        const double q = floor(tc[nh - 1] / dt_in);
        nmax = (q < (double)N) ? (int64_t)q : N;
        if (tile == 0 && lane == 0) counts[c] = nmax;
    }
    const int64_t j = tile * WAVE + lane + 1;                         // 1-based sample index
    const double tm = (double)j * dt;                                 // :492
    int64_t lo = 0, hi = nh - 1;
    if (j <= nmax) {
        while (lo < hi) {                                             // last index with t[i] <= tm (strict: t[i] < tm)
            const int64_t mid = (lo + hi + 1) >> 1;
            const bool le = strict ? (tc[mid] < tm) : (tc[mid] <= tm);
            if (le) lo = mid; else hi = mid - 1;
        }
    }
    const double tau = tm - tc[lo];                                   // :497
    const int seg_lo = (int)lo, seg_hi = (int)(lo >> 32);
    const size_t rows = discard_vt ? (size_t)d : 2 * (size_t)d + 1;   // :487
    int64_t nvalid = (nmax - tile * WAVE) < WAVE ? (nmax - tile * WAVE) : WAVE;
    if (nvalid < 0) nvalid = 0;
    for (int s = 0; s < (int)nvalid; ++s) {
        const int64_t i = ((int64_t)__builtin_amdgcn_readlane(seg_hi, s) << 32) |
                          (uint32_t)__builtin_amdgcn_readlane(seg_lo, s);
        const double ta = readlane_d(tau, s), tms = readlane_d(tm, s);
        const double *x0 = X + ((size_t)c * n_sk + (size_t)i) * d, *v0 = V + ((size_t)c * n_sk + (size_t)i) * d;
        const uint8_t *a0 = (IA && ia_mode) ? IA + ((size_t)c * n_sk + (size_t)i) * d : nullptr;
        double *o = out + ((size_t)c * N + (size_t)(tile * WAVE + s)) * rows;
        double sn = 0.0, cs = 1.0;
        if (BOOMERANG) { if (BOOMERANG == 2) det_sincos_v1(ta, sn, cs); else det_sincos_v2(ta, sn, cs); }
        if (rhmc_target >= 0) {                                       // RHMC: velocity-Verlet from the event's (x, v)
            double xq[4], vq[4];
            rhmc_integrate(d, lane, x0, v0, ta, rhmc_h, rhmc_target, tparams, xq, vq);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int q = lane + k * WAVE; if (q < d) { o[q] = xq[k]; if (!discard_vt) o[d + q] = vq[k]; } }
        } else if (suzz) {                                            // SpeedUpZigZagSamplers.jl:71-79: x_t = y + (v1 X1(t)) v
            double av, kv;
            suzz_scalars(d, lane, x0, v0, ta, av, kv, arv);
            for (int q = lane; q < d; q += WAVE) {
                const double vq = v0[q], y = x0[q] - av * vq;
                o[q] = y + kv * vq;
                if (!discard_vt) o[d + q] = vq;
            }
        } else
        for (int q = lane; q < d; q += WAVE) {
            const double xq = x0[q];
            double vq = v0[q];
            if (a0) vq = (ia_mode == 2) ? vq * (a0[q] ? 1.0 : 0.0) : (a0[q] ? vq : 0.0);   // :547-549 / :667
            if (BOOMERANG) {                                          // BoomerangSamplers.jl:31
                o[q] = xq * cs + vq * sn;
                if (!discard_vt) o[d + q] = (-xq) * sn + vq * cs;
            } else {                                                  // ZigZagSamplers.jl:80
                o[q] = xq + vq * ta;
                if (!discard_vt) o[d + q] = vq;
            }
        }
        if (!discard_vt && lane == 0) o[2 * d] = tms;                 // :505-508
    }
}

// The same samples, streamed: the common case (linear flow or Boomerang rotation) as a flat copy-like kernel.  One wave per (chain,
// tile of 64 samples) as above, but instead of 64 serial row copies the tile's 64 x rows output elements are walked in memory
// order -- element e = 64 k + lane belongs to sample e / rows, row entry e % rows -- so every store instruction writes 512
// contiguous bytes and every lane works (a d = 50 row fills 50 of 64 lanes in the row-per-iteration form); the per-sample segment
// index, residual time and sin / cos sit in LDS.  Four elements per lane in flight.  HBM-bound: 8(2d+1) B read per touched event +
// 8 rows B written per sample.
template <int BOOMERANG>   // 0: linear flow; 1 / 2: the Boomerang's rotation in arithmetic contract v2 / v1 (pdmp_config_t.rng_version)
__global__ __launch_bounds__(256) void from_skeleton_flat_kernel(int d, int64_t C, int64_t n_sk, const double *X, const double *V,
                                                                 const double *t, int64_t N, double dt_in, int discard_vt, double *out,
                                                                 const uint8_t *IA, int ia_mode, int strict, int64_t n_use,
                                                                 const int64_t *lens, int64_t *counts) {
    __shared__ int seg_s[4][WAVE];
    __shared__ double tau_s[4][WAVE], tm_s[4][WAVE], sn_s[4][WAVE], cs_s[4][WAVE];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const int64_t tiles = (N + WAVE - 1) / WAVE;
    const int64_t w = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wv);
    if (w >= C * tiles) return;
    const int64_t c = w / tiles, tile = w % tiles;
    const double *tc = t + (size_t)c * n_sk;
    int64_t nh = (n_use > 0 && n_use < n_sk) ? n_use : n_sk;
    if (lens) { const int64_t l = lens[c]; nh = (l > 0 && l < nh) ? l : nh; }
    const double dt = dt_in > 0.0 ? dt_in : tc[nh - 1] / (double)N;   // :483-484
    int64_t nmax = N;
    if (dt_in > 0.0 && counts) {                                      // n_skel = Int(floor(T_end / dt)), :608
        const double q = floor(tc[nh - 1] / dt_in);
        nmax = (q < (double)N) ? (int64_t)q : N;
        if (tile == 0 && lane == 0) counts[c] = nmax;
    }
    const int64_t j = tile * WAVE + lane + 1;                         // 1-based sample index
    const double tm = (double)j * dt;                                 // :492
    int lo = 0, hi = (int)nh - 1;
    if (j <= nmax) {
        while (lo < hi) {                                             // last index with t[i] <= tm (strict: t[i] < tm)
            const int mid = (lo + hi + 1) >> 1;
            const bool le = strict ? (tc[mid] < tm) : (tc[mid] <= tm);
            if (le) lo = mid; else hi = mid - 1;
        }
    }
    const double tau = tm - tc[lo];                                   // :497
    double sn = 0.0, cs = 1.0;
    if (BOOMERANG) { if (BOOMERANG == 2) det_sincos_v1(tau, sn, cs); else det_sincos_v2(tau, sn, cs); }
    seg_s[wv][lane] = lo; tau_s[wv][lane] = tau; tm_s[wv][lane] = tm; sn_s[wv][lane] = sn; cs_s[wv][lane] = cs;
    __builtin_amdgcn_wave_barrier();                                  // (LDS traffic of one wave is ordered; the barrier only stops reordering by the compiler)
    int64_t nvalid = (nmax - tile * WAVE) < WAVE ? (nmax - tile * WAVE) : WAVE;
    if (nvalid <= 0) return;
    const int rows = discard_vt ? d : 2 * d + 1;                      // :487
    const int total = (int)nvalid * rows;
    const double *Xc = X + (size_t)c * n_sk * d, *Vc = V + (size_t)c * n_sk * d;
    const uint8_t *Ac = (IA && ia_mode) ? IA + (size_t)c * n_sk * d : nullptr;
    double *o = out + ((size_t)c * N + (size_t)tile * WAVE) * rows;
    const unsigned long long magic = 0x100000000ull / (unsigned)rows;        // floor(2^32 / rows): (e * magic) >> 32 is e / rows or one less
    constexpr int U = 4;
    for (int e0 = lane; e0 < total; e0 += U * WAVE) {
        double xv[U], vv[U], ta[U], s1[U], c1[U], tmv[U];
        int r[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * WAVE;
            ok[u] = e < total;
            const int ee = ok[u] ? e : 0;
            int smp = (int)(((unsigned long long)(unsigned)ee * magic) >> 32);   // (rows = 1: magic = 2^32, exact)
            if ((smp + 1) * rows <= ee) ++smp;
            r[u] = ee - smp * rows;
            const int q = r[u] < d ? r[u] : (r[u] < 2 * d ? r[u] - d : 0);
            const size_t off = (size_t)seg_s[wv][smp] * d + q;
            xv[u] = Xc[off];                                         // (plain loads: a row is read by ~1.6 samples; streaming loads measured 10 % slower)
            double vq = Vc[off];
            if (Ac) { const uint8_t a = Ac[off]; vq = (ia_mode == 2) ? vq * (a ? 1.0 : 0.0) : (a ? vq : 0.0); }   // :547-549 / :667
            vv[u] = vq;
            ta[u] = tau_s[wv][smp]; tmv[u] = tm_s[wv][smp];
            s1[u] = sn_s[wv][smp]; c1[u] = cs_s[wv][smp];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            double val;
            if (r[u] < d) val = BOOMERANG ? xv[u] * c1[u] + vv[u] * s1[u] : xv[u] + vv[u] * ta[u];             // BoomerangSamplers.jl:31 / ZigZagSamplers.jl:80
            else if (r[u] < 2 * d) val = BOOMERANG ? (-xv[u]) * s1[u] + vv[u] * c1[u] : vv[u];
            else val = tmv[u];                                                                                      // :505-508
#ifdef PDMP_FS_PLAIN_STORES        // A/B builds
            o[e0 + u * WAVE] = val;
#else
            __builtin_nontemporal_store(val, &o[e0 + u * WAVE]);      // samples: consecutive lanes, consecutive doubles, written once -- streaming stores leave the L2 to the skeleton rows
#endif
        }
    }
}

// Closing point of the time-horizon variant sample_skeleton(sampler, T::Float64, ...) (src/sample.jl:323-439), on the device:
// per chain, count[c] = number of history columns once the chain is cut at T -- the events with t <= T (:372) plus, when the
// last of them lies strictly before T, one closing column at exactly t = T obtained by flow from it (:385-420), which
// overwrites the overshooting event in place; count[c] = -(events) - 1 when the chain has not passed T yet (capacity too
// small: the caller reruns with a larger n_sk, as _grow_history does, Composites.jl:172-191).  One wave per chain.
template <int BOOMERANG>   // 0: linear flow; 1 / 2: the Boomerang's rotation in arithmetic contract v2 / v1 (pdmp_config_t.rng_version)
__global__ __launch_bounds__(256) void close_at_kernel(int d, int64_t C, int64_t n_sk_cap, double *X, double *V, double *t, double T,
                                                       int64_t *count, int mode = 0, int target = 0, double rhmc_h = 0.0,
                                                       const double *tparams = nullptr, pdmp_diag_t dg = {}, const int64_t *valid = nullptr, int arv = 2) {
    const int lane = lane_id();
    const int64_t c = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (c >= C) return;
    double *tc = t + (size_t)c * n_sk_cap;
    // valid: the columns chain c recorded (a launch that stops chains at T leaves the rest of the capacity unwritten); the column
    // stride stays the capacity
    const int64_t n_sk = valid ? valid[c] : n_sk_cap;
    int64_t lo = 0, hi = n_sk - 1;                                    // last index with t[i] <= T (t[0] <= T: a chunk starts before T)
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (tc[mid] <= T) lo = mid; else hi = mid - 1;
    }
    const int64_t k = lo + 1;                                         // events recorded by the reference loop
    if (T > 0.0 && lo == n_sk - 1 && tc[lo] <= T) {                   // the chain has not provably passed T within this capacity
        if (tc[lo] < T) { if (lane == 0) count[c] = -k - 1; return; }
    }
    if (T == 0.0 || tc[lo] == T) { if (lane == 0) count[c] = (T == 0.0) ? 1 : k; return; }   // `while state.t < T` ends on t == T: no closing column
    const double tau = T - tc[lo];
    const double *x0 = X + ((size_t)c * n_sk_cap + (size_t)lo) * d, *v0 = V + ((size_t)c * n_sk_cap + (size_t)lo) * d;
    double *x1 = X + ((size_t)c * n_sk_cap + (size_t)k) * d, *v1 = V + ((size_t)c * n_sk_cap + (size_t)k) * d;
    // StickyPDMP: v_used = active_prev ? v_prev : 0 (sample.jl:386-390), is_active of the closing column = active_prev (:399-401)
    const uint8_t *a0 = (mode == 3 && dg.is_active) ? dg.is_active + ((size_t)c * n_sk_cap + (size_t)lo) * d : nullptr;
    uint8_t *a1 = a0 ? dg.is_active + ((size_t)c * n_sk_cap + (size_t)k) * d : nullptr;
    double sn = 0.0, cs = 1.0;
    if (BOOMERANG) { if (BOOMERANG == 2) det_sincos_v1(tau, sn, cs); else det_sincos_v2(tau, sn, cs); }
    if (mode == 2) {                                                                      // RHMC: velocity-Verlet to T
        double rx[4], rv[4];
        rhmc_integrate(d, lane, x0, v0, tau, rhmc_h, target, tparams, rx, rv);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) { const int q = lane + k2 * WAVE; if (q < d) { x1[q] = rx[k2]; v1[q] = rv[k2]; } }
    } else if (mode == 1) {                                                               // SpeedUpZigZag
        double av, kv;
        suzz_scalars(d, lane, x0, v0, tau, av, kv, arv);
        for (int q = lane; q < d; q += WAVE) { const double vq = v0[q]; x1[q] = (x0[q] - av * vq) + kv * vq; v1[q] = vq; }
    } else
    for (int q = lane; q < d; q += WAVE) {
        const double xq = x0[q];
        double vq = v0[q];
        if (a0) { const uint8_t a = a0[q]; vq = a ? vq : 0.0; a1[q] = a; }
        if (BOOMERANG) { x1[q] = xq * cs + vq * sn; v1[q] = (-xq) * sn + vq * cs; }      // BoomerangSamplers.jl:31
        else { x1[q] = xq + vq * tau; v1[q] = vq; }                                       // ZigZagSamplers.jl:80
    }
    if (lane == 0) {
        tc[k] = T; count[c] = k + 1;
        // statistics of a non-event point are zeros (:402-408); horizon stays the overshooting state's (record! of the same state object)
        const size_t e = (size_t)c * n_sk_cap + (size_t)k;
        if (dg.ar) dg.ar[e] = 0.0;
        if (dg.errored_bound) dg.errored_bound[e] = 0;
        if (dg.rejected) dg.rejected[e] = 0;
        if (dg.hitting_horizon) dg.hitting_horizon[e] = 0;
        if (dg.error_value_ar) for (int q = 0; q < 5; ++q) dg.error_value_ar[5 * e + q] = 0.0;
    }
}

// _trim_history (Composites.jl:198-214) for an ensemble: the ragged histories (count[c] columns of chain c inside a capacity of
// n_sk columns) packed back to back.  Pass 1 (one block): offsets[c] = exclusive prefix sum of count -- per wave a shuffle scan,
// the wave totals combined through LDS, a running carry across 1024-chain tiles.  Pass 2: one wave per (chain, 64-column tile)
// copies the rows (coalesced on both sides).
__global__ __launch_bounds__(1024) void compact_offsets_kernel(int64_t C, const int64_t *count, int64_t *offsets) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < C; base += 1024) {
        const int64_t c = base + threadIdx.x;
        int64_t v = (c < C) ? count[c] : 0;
        if (v < 0) v = 0;                                             // an unfinished chain (negative count) contributes nothing
        int64_t inc = v;
        for (int off = 1; off < WAVE; off <<= 1) {                    // inclusive scan inside the wave
            const int64_t o = (int64_t)__shfl_up((long long)inc, off, WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == WAVE - 1) wsum[wave] = inc;
        __syncthreads();
        int64_t before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (c < C) offsets[c] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[C] = carry_s;
}
__global__ __launch_bounds__(256) void compact_rows_kernel(int d, int64_t C, int64_t n_sk, const int64_t *count, const int64_t *offsets,
                                                           const double *X, const double *V, const double *t, pdmp_diag_t dg,
                                                           double *Xo, double *Vo, double *to, pdmp_diag_t og) {
    const int lane = lane_id();
    const int64_t tiles = (n_sk + WAVE - 1) / WAVE;
    const int64_t w = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w >= C * tiles) return;
    const int64_t c = w / tiles, k0 = (w % tiles) * WAVE;
    int64_t n = count[c];
    if (n < 0) n = 0;
    if (k0 >= n) return;
    const int64_t nk = (n - k0) < WAVE ? (n - k0) : WAVE;
    const size_t src = (size_t)c * n_sk + (size_t)k0, dst = (size_t)offsets[c] + (size_t)k0;
    const size_t words = (size_t)nk * d;
    for (size_t q = lane; q < words; q += WAVE) { Xo[dst * d + q] = X[src * d + q]; Vo[dst * d + q] = V[src * d + q]; }
    if (lane < nk) {
        to[dst + lane] = t[src + lane];
        if (dg.horizon && og.horizon) og.horizon[dst + lane] = dg.horizon[src + lane];
        if (dg.ar && og.ar) og.ar[dst + lane] = dg.ar[src + lane];
        if (dg.errored_bound && og.errored_bound) og.errored_bound[dst + lane] = dg.errored_bound[src + lane];
        if (dg.rejected && og.rejected) og.rejected[dst + lane] = dg.rejected[src + lane];
        if (dg.hitting_horizon && og.hitting_horizon) og.hitting_horizon[dst + lane] = dg.hitting_horizon[src + lane];
    }
    if (dg.error_value_ar && og.error_value_ar)
        for (size_t q = lane; q < (size_t)nk * 5; q += WAVE) og.error_value_ar[dst * 5 + q] = dg.error_value_ar[src * 5 + q];
    if (dg.is_active && og.is_active)
        for (size_t q = lane; q < words; q += WAVE) og.is_active[dst * d + q] = dg.is_active[src * d + q];
}

// ---- Time-horizon variant as a kernel feature (pdmp_sample_skeleton_until_device): bookkeeping between chunks.
// After a chunk has been sampled and closed, for every launch chain c (original chain o = map_in ? map_in[c] : c):
//   kept = the columns of this chunk that belong to the chain's history = |count[c]| columns (count >= 0: finished, the closing
//          column included; count < 0: -count - 1 columns, all events) minus `skip` (1 for a resumed chunk: its column 0 repeats
//          the previous chunk's last event);
//   start[c] = total[o] (the chain's history length so far = where this chunk's columns go), total[o] += kept;
//   the chains that have not reached T are appended to map_out in launch order -- the north star's "wavefront ballot / prefix-sum to
//   compact the branchy stream": per wave a ballot + popcount rank, wave totals through LDS, a running carry across tiles of 1024.
__global__ __launch_bounds__(1024) void until_advance_kernel(int64_t C, const int64_t *count, int64_t skip, const int64_t *map_in,
                                                             int64_t *total, int64_t *start, int64_t *map_out, int64_t *n_active) {
    __shared__ int wsum[16];
    __shared__ int64_t carry_s;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < C; base += 1024) {
        const int64_t c = base + threadIdx.x;
        bool act = false;
        int64_t o = 0;
        if (c < C) {
            const int64_t cnt = count[c];
            o = map_in ? map_in[c] : c;
            const int64_t cols = cnt >= 0 ? cnt : -cnt - 1;
            const int64_t kept = cols > skip ? cols - skip : 0;
            const int64_t before = total[o];
            start[c] = before;
            total[o] = before + kept;
            act = cnt < 0;
        }
        const unsigned long long ball = __ballot(act);
        const int rank = __builtin_popcountll(ball & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __builtin_popcountll(ball);
        __syncthreads();
        int64_t before = carry_s;
        int tile_total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) before += wsum[w]; tile_total += wsum[w]; }
        if (act) map_out[before + rank] = o;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tile_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_active = carry_s;
}
// One chunk's columns [skip, |count[c]|) of every launch chain into the dense buffers: column k of launch chain c (original chain o)
// goes to dense column offsets[o] + start[c] + (k - skip).  One wave per (chain, 64-column tile), coalesced on both sides.
__global__ __launch_bounds__(256) void compact_chunk_kernel(int d, int64_t C, int64_t cap, const int64_t *count, int64_t skip,
                                                            const int64_t *map, const int64_t *start, const int64_t *offsets,
                                                            const double *X, const double *V, const double *t, pdmp_diag_t dg,
                                                            double *Xo, double *Vo, double *to, pdmp_diag_t og) {
    const int lane = lane_id();
    const int64_t tiles = (cap + WAVE - 1) / WAVE;
    const int64_t w = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w >= C * tiles) return;
    const int64_t c = w / tiles, k0 = skip + (w % tiles) * WAVE;
    const int64_t cnt = count[c], n = cnt >= 0 ? cnt : -cnt - 1;
    if (k0 >= n) return;
    const int64_t nk = (n - k0) < WAVE ? (n - k0) : WAVE;
    const int64_t o = map ? map[c] : c;
    const size_t src = (size_t)c * cap + (size_t)k0, dst = (size_t)(offsets[o] + start[c] + (k0 - skip));
    const size_t words = (size_t)nk * d;
    for (size_t q = lane; q < words; q += WAVE) { Xo[dst * d + q] = X[src * d + q]; Vo[dst * d + q] = V[src * d + q]; }
    if (lane < nk) {
        to[dst + lane] = t[src + lane];
        if (dg.horizon && og.horizon) og.horizon[dst + lane] = dg.horizon[src + lane];
        if (dg.ar && og.ar) og.ar[dst + lane] = dg.ar[src + lane];
        if (dg.errored_bound && og.errored_bound) og.errored_bound[dst + lane] = dg.errored_bound[src + lane];
        if (dg.rejected && og.rejected) og.rejected[dst + lane] = dg.rejected[src + lane];
        if (dg.hitting_horizon && og.hitting_horizon) og.hitting_horizon[dst + lane] = dg.hitting_horizon[src + lane];
    }
    if (dg.error_value_ar && og.error_value_ar)
        for (size_t q = lane; q < (size_t)nk * 5; q += WAVE) og.error_value_ar[dst * 5 + q] = dg.error_value_ar[src * 5 + q];
    if (dg.is_active && og.is_active)
        for (size_t q = lane; q < words; q += WAVE) og.is_active[dst * d + q] = dg.is_active[src * d + q];
}

// Fused sample_from_skeleton + moment sums: per chain the sums of x_j and x_j^2 over the N
// equispaced samples of sample.jl:475-513, without materialising the d x N x C sample array.  One wave per chain,
// samples in order, each lane accumulates its coordinates left to right: deterministic (no atomics), equal to
// a sequential sum over the samples of sample_from_skeleton.
template <int BOOMERANG>   // 0: linear flow; 1 / 2: the Boomerang's rotation in arithmetic contract v2 / v1 (pdmp_config_t.rng_version)
__global__ __launch_bounds__(256) void moments_kernel(int d, int64_t C, int64_t n_sk, const double *X, const double *V,
                                                      const double *t, int64_t N, double *sum_x, double *sum_x2,
                                                      int mode = 0, int target = 0, double rhmc_h = 0.0, const double *tparams = nullptr,
                                                      const uint8_t *IA = nullptr, const int64_t *lens = nullptr, int arv = 2) {
    const int lane = lane_id();
    const int64_t c = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (c >= C) return;
    const double *tc = t + (size_t)c * n_sk;
    int64_t nh = n_sk;                                                // history columns of this chain (ragged: the time-horizon variant)
    if (lens) { const int64_t l = lens[c]; nh = (l > 0 && l < nh) ? l : nh; }
    const double dt = tc[nh - 1] / (double)N;
    const int64_t tiles = (N + WAVE - 1) / WAVE;
    constexpr int M = 4;
    for (int q0 = 0; q0 < d; q0 += M * WAVE) {
        double s1[M], s2[M];
#pragma unroll
        for (int m = 0; m < M; ++m) { s1[m] = 0.0; s2[m] = 0.0; }
        for (int64_t tile = 0; tile < tiles; ++tile) {
            const int64_t j = tile * WAVE + lane + 1;
            const double tm = (double)j * dt;
            int64_t lo = 0, hi = nh - 1;
            if (j <= N) {
                while (lo < hi) {
                    const int64_t mid = (lo + hi + 1) >> 1;
                    if (tc[mid] <= tm) lo = mid; else hi = mid - 1;
                }
            }
            const double tau = tm - tc[lo];
            const int seg_lo = (int)lo, seg_hi = (int)(lo >> 32);
            const int nvalid = (int)((N - tile * WAVE) < WAVE ? (N - tile * WAVE) : WAVE);
            for (int s = 0; s < nvalid; ++s) {
                const int64_t i = ((int64_t)__builtin_amdgcn_readlane(seg_hi, s) << 32) | (uint32_t)__builtin_amdgcn_readlane(seg_lo, s);
                const double ta = readlane_d(tau, s);
                const double *x0 = X + ((size_t)c * n_sk + (size_t)i) * d, *v0 = V + ((size_t)c * n_sk + (size_t)i) * d;
                const uint8_t *a0 = (IA && mode == 3) ? IA + ((size_t)c * n_sk + (size_t)i) * d : nullptr;   // StickyPDMP, sample.jl:547-549
                double sn = 0.0, cs = 1.0;
                if (BOOMERANG) { if (BOOMERANG == 2) det_sincos_v1(ta, sn, cs); else det_sincos_v2(ta, sn, cs); }
                double av = 0.0, kv = 0.0, rx[4] = {0, 0, 0, 0}, rv[4];
                if (mode == 1) suzz_scalars(d, lane, x0, v0, ta, av, kv, arv);                          // SpeedUpZigZag: the sampler's own flow
                else if (mode == 2) rhmc_integrate(d, lane, x0, v0, ta, rhmc_h, target, tparams, rx, rv);   // RHMC (d <= 256: q0 == 0)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const int q = q0 + m * WAVE + lane;
                    if (q < d) {
                        const double xq = x0[q], vq = (a0 && !a0[q]) ? 0.0 : v0[q];
                        const double xs = (mode == 1) ? (xq - av * vq) + kv * vq : (mode == 2) ? rx[m]
                                        : BOOMERANG ? xq * cs + vq * sn : xq + vq * ta;
                        s1[m] += xs;
                        s2[m] += xs * xs;
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int q = q0 + m * WAVE + lane;
            if (q < d) { sum_x[(size_t)c * d + q] = s1[m]; sum_x2[(size_t)c * d + q] = s2[m]; }
        }
    }
}

template <int RV>
__global__ void debug_rng_kernel(uint64_t seed, int64_t chain, int kind, int64_t n, double *out) {
    ChainRngT<RV> rng;
    rng.init(seed, (uint64_t)chain);
    const int lane = lane_id();
    if (kind == 2) {
        for (int64_t i = lane; i < n; i += WAVE) out[i] = rng.normal_coord((int)i);
        return;
    }
    for (int64_t i = 0; i < n; ++i) {
        const double r = (kind == 0) ? rng.uniform() : rng.exponential();
        if (lane == 0) out[i] = r;
    }
}

// Synthetic initial states (pdmp_init_states_device): one thread per chain, everything in the chain's own sequential order so that
// the checker's orc_init_states reproduces it bit for bit.  Stream 1 of a chain = its Philox stream with the top bit of the chain word set.
template <int RV>
__global__ void init_states_kernel(int64_t C, int64_t chain0, uint64_t seed, int d, double x_mean, double x_scale, int v_kind,
                                   double *__restrict__ x, double *__restrict__ v) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    ChainRngT<RV> rng;
    rng.init_lane(seed, ((uint64_t)(chain0 + c)) | (1ull << 63));
    double *xc = x + c * d, *vc = v + c * d;
    for (int p = 0; 2 * p < d; ++p) {
        double z0, z1;
        rng.normal_pair(p, z0, z1);
        xc[2 * p] = x_mean + x_scale * z0;
        if (2 * p + 1 < d) xc[2 * p + 1] = x_mean + x_scale * z1;
    }
    rng.normals_done(d);
    if (v_kind == PDMP_V_RADEMACHER) {
        for (int i = 0; i < d; ++i) vc[i] = (rng.uniform() < 0.5) ? -1.0 : 1.0;
        return;
    }
    double ss = 0.0;
    for (int p = 0; 2 * p < d; ++p) {
        double z0, z1;
        rng.normal_pair(p, z0, z1);
        vc[2 * p] = z0; ss = ss + z0 * z0;
        if (2 * p + 1 < d) { vc[2 * p + 1] = z1; ss = ss + z1 * z1; }
    }
    if (v_kind == PDMP_V_SPHERE) {
        const double nrm = __dsqrt_rn(ss);
        for (int i = 0; i < d; ++i) vc[i] = vc[i] / nrm;
    }
}

int normalize_config(pdmp_config_t &c, bool from_source = false) {
    if (c.dim <= 0) return fail(PDMP_ERR_ARGUMENT, "dimension dim must be positive. Current value: %d", c.dim);          // ZigZagSamplers.jl:62-64
    if (c.grid_size < 0) return fail(PDMP_ERR_ARGUMENT, "grid_size must be non-negative. Current value: %d", c.grid_size); // :66-68
    if (c.grid_size == 1 || c.grid_size > WAVE) return fail(PDMP_ERR_UNSUPPORTED, "grid_size must be in 2..64. Current value: %d", c.grid_size);
    if (c.sampler < PDMP_ZIGZAG || c.sampler > PDMP_STICKY) return fail(PDMP_ERR_ARGUMENT, "unknown sampler id %d", c.sampler);
    if (c.rng_version == 0) c.rng_version = 2;                                                       // the default arithmetic contract (DESIGN.md "Arithmetic contract v2")
    if (c.rng_version != 1 && c.rng_version != 2) return fail(PDMP_ERR_ARGUMENT, "rng_version must be 0 (default), 1 or 2. Current value: %d", c.rng_version);
    if (c.sampler == PDMP_STICKY) {
        if (c.target_id == PDMP_T_GAUSS_DENSE || c.target_id == PDMP_T_USER) return fail(PDMP_ERR_UNSUPPORTED, "StickyZigZag: built-in separable / banana targets only");
        const int64_t base_params = (c.target_id == PDMP_T_GAUSS_DIAG) ? 2 * (int64_t)c.dim : 0;
        if (c.n_target_params != base_params + c.dim || !c.target_params) return fail(PDMP_ERR_ARGUMENT, "StickyZigZag needs kappa[dim] as the last dim target_params");
    }
    if (c.sampler == PDMP_RHMC) {                                                                    // RandomizedHamiltonianMonteCarlo.jl:59-95
        if (!(c.refresh_rate > 0.0) || !std::isfinite(c.refresh_rate)) return fail(PDMP_ERR_ARGUMENT, "refresh_rate must be finite and positive. Current value: %g", c.refresh_rate);
        if (!(0.0 < c.rh_phi && c.rh_phi <= 1.5707963267948966)) return fail(PDMP_ERR_ARGUMENT, "phi must satisfy 0 < phi <= pi/2. Current value: %g", c.rh_phi);
        if (!(c.rh_step_size > 0.0) || !std::isfinite(c.rh_step_size)) return fail(PDMP_ERR_ARGUMENT, "step_size must be finite and positive. Current value: %g", c.rh_step_size);
        if (c.target_id == PDMP_T_GAUSS_DENSE) return fail(PDMP_ERR_UNSUPPORTED, "RHMC: no dense-precision target");
        c.grid_size = 0; c.vectorized_bound = 0; c.signed_bound = 0;                                  // :91-95
    }
    if (c.sampler == PDMP_SUZZ) {
        if (c.target_id == PDMP_T_GAUSS_DENSE || c.target_id == PDMP_T_USER) return fail(PDMP_ERR_UNSUPPORTED, "SpeedUpZigZag: built-in separable / banana targets only");
    }
    if (c.sampler == PDMP_FECMC) {                                                                   // ForwardEventChainMonteCarlo.jl:298-310
        if (c.dim < 2) return fail(PDMP_ERR_ARGUMENT, "The dimension must be at least 2 to use the ForwardEventChain. Got dimension %d", c.dim);
        if (c.dim == 2) c.fe_mix_p = 0.0;
        c.refresh_rate = 0.0;
        if (c.fe_speed_factor == 0.0) c.fe_speed_factor = 1.0;
        if (!std::isfinite(c.fe_speed_factor)) return fail(PDMP_ERR_ARGUMENT, "speed_factor must be finite");
        if (c.target_id == PDMP_T_GAUSS_DENSE) return fail(PDMP_ERR_UNSUPPORTED, "ForwardECMC: no dense-precision target");
    }
    if (c.target_id == PDMP_T_USER && !from_source) return fail(PDMP_ERR_ARGUMENT, "PDMP_T_USER handles are made by pdmp_create_from_source");
    if (c.target_id < 0 || c.target_id > PDMP_T_USER) return fail(PDMP_ERR_ARGUMENT, "unknown target id %d", c.target_id);
    if (c.target_id == PDMP_T_GAUSS_DENSE) {
        if (c.n_target_params != (int64_t)c.dim * c.dim || !c.target_params) return fail(PDMP_ERR_ARGUMENT, "GAUSS_DENSE needs target_params = P[dim*dim] (column-major)");
        if (c.deriv_mode != PDMP_DERIV_FORWARD)
            return fail(PDMP_ERR_UNSUPPORTED, "dense-precision target: forward-mode slopes only");
        // Boomerang: the batched form (dense_batched.hip: dr_boom_kernel, one GEMM per event over all chains of a launch) up to dim = 1536, the
        // per-chain mat-vec kernel for small ensembles, tape replays and beyond (round 4).  The batched forms serve grid_size = 10; every
        // other grid_size (round 5: a free parameter of the group kernel, UpperBound.jl:92,203) runs the per-chain mat-vec kernel.
        if (group_waves_per_block((c.dim + WAVE - 1) / WAVE, 0, 6) < 1) return fail(PDMP_ERR_UNSUPPORTED, "dense-precision target: dim too large for the LDS tiles");
    }
    if (c.deriv_mode != PDMP_DERIV_FORWARD && c.deriv_mode != PDMP_DERIV_FINITE_DIFF) return fail(PDMP_ERR_ARGUMENT, "Unsupported AD_backend id %d", c.deriv_mode);
    if (!std::isfinite(c.tmax) || c.tmax < 0.0) return fail(PDMP_ERR_ARGUMENT, "tmax must be finite and non-negative");
    if (c.tmax == 0.0) { c.tmax = 1.0; c.adaptive = 1; }                                            // ZigZagSamplers.jl:73
    if (c.sampler == PDMP_ZIGZAG || c.sampler == PDMP_SUZZ || c.sampler == PDMP_STICKY) {
        if (c.signed_bound && !c.vectorized_bound) c.signed_bound = 0;                              // :75-78 (SpeedUpZigZagSamplers.jl:65-68)
    } else {
        c.vectorized_bound = 0;                                                                     // BPS :37, Boomerang :36
    }

    if (c.target_id == PDMP_T_GAUSS_DIAG && c.sampler != PDMP_STICKY && (c.n_target_params != 2 * (int64_t)c.dim || !c.target_params))
        return fail(PDMP_ERR_ARGUMENT, "GAUSS_DIAG needs target_params = mu[dim], prec[dim]");
    if (c.max_iter_per_event <= 0) c.max_iter_per_event = 1000000;
    if (c.max_iter_per_event > 0x7fffffff) c.max_iter_per_event = 0x7fffffff;
    return PDMP_OK;
}

typedef hipError_t (*launch_fn)(const SkelParams &, hipStream_t);
launch_fn pick_launcher(int sampler, int target) {
    static const launch_fn table[3][3] = {
        {launch_skeleton<S_ZIGZAG, T_GAUSS_ISO>, launch_skeleton<S_ZIGZAG, T_GAUSS_DIAG>, launch_skeleton<S_ZIGZAG, T_BANANA>},
        {launch_skeleton<S_BPS, T_GAUSS_ISO>, launch_skeleton<S_BPS, T_GAUSS_DIAG>, launch_skeleton<S_BPS, T_BANANA>},
        {launch_skeleton<S_BOOMERANG, T_GAUSS_ISO>, launch_skeleton<S_BOOMERANG, T_GAUSS_DIAG>, launch_skeleton<S_BOOMERANG, T_BANANA>},
    };
    if (sampler < 0 || sampler > 2 || target < 0 || target > 2) return nullptr;   // the wave-per-chain kernel serves the three BASELINE samplers only
    return table[sampler][target];
}

typedef hipError_t (*glaunch_fn)(const GroupParams &, hipStream_t);
glaunch_fn pick_group_launcher(int sampler, int target) {
    static const glaunch_fn table[7][4] = {
        {launch_group<S_ZIGZAG, T_GAUSS_ISO>, launch_group<S_ZIGZAG, T_GAUSS_DIAG>, launch_group<S_ZIGZAG, T_BANANA>, launch_group<S_ZIGZAG, T_GAUSS_DENSE>},
        {launch_group<S_BPS, T_GAUSS_ISO>, launch_group<S_BPS, T_GAUSS_DIAG>, launch_group<S_BPS, T_BANANA>, launch_group<S_BPS, T_GAUSS_DENSE>},
        {launch_group<S_BOOMERANG, T_GAUSS_ISO>, launch_group<S_BOOMERANG, T_GAUSS_DIAG>, launch_group<S_BOOMERANG, T_BANANA>, launch_group<S_BOOMERANG, T_GAUSS_DENSE>},
        {launch_group<S_FECMC, T_GAUSS_ISO>, launch_group<S_FECMC, T_GAUSS_DIAG>, launch_group<S_FECMC, T_BANANA>, nullptr},
        {launch_group<S_SUZZ, T_GAUSS_ISO>, launch_group<S_SUZZ, T_GAUSS_DIAG>, launch_group<S_SUZZ, T_BANANA>, nullptr},
        {launch_group<S_RHMC, T_GAUSS_ISO>, launch_group<S_RHMC, T_GAUSS_DIAG>, launch_group<S_RHMC, T_BANANA>, nullptr},
        {launch_group<S_STICKY, T_GAUSS_ISO>, launch_group<S_STICKY, T_GAUSS_DIAG>, launch_group<S_STICKY, T_BANANA>, nullptr},
    };
    if (sampler < 0 || sampler > 6 || target < 0 || target > 3) return nullptr;
    return table[sampler][target];
}

glaunch_fn pick_group_tape_launcher(int sampler, int target) {
    if (target == T_GAUSS_DENSE)                               // (round 4: the per-chain mat-vec kernel replays tapes for Zig-Zag and BPS)
        return sampler == S_ZIGZAG ? launch_group_tape<S_ZIGZAG, T_GAUSS_DENSE> : sampler == S_BPS ? launch_group_tape<S_BPS, T_GAUSS_DENSE>
             : sampler == S_BOOMERANG ? launch_group_tape<S_BOOMERANG, T_GAUSS_DENSE> : nullptr;
    static const glaunch_fn table[7][3] = {
        {launch_group_tape<S_ZIGZAG, T_GAUSS_ISO>, launch_group_tape<S_ZIGZAG, T_GAUSS_DIAG>, launch_group_tape<S_ZIGZAG, T_BANANA>},
        {launch_group_tape<S_BPS, T_GAUSS_ISO>, launch_group_tape<S_BPS, T_GAUSS_DIAG>, launch_group_tape<S_BPS, T_BANANA>},
        {launch_group_tape<S_BOOMERANG, T_GAUSS_ISO>, launch_group_tape<S_BOOMERANG, T_GAUSS_DIAG>, launch_group_tape<S_BOOMERANG, T_BANANA>},
        {launch_group_tape<S_FECMC, T_GAUSS_ISO>, launch_group_tape<S_FECMC, T_GAUSS_DIAG>, launch_group_tape<S_FECMC, T_BANANA>},
        {launch_group_tape<S_SUZZ, T_GAUSS_ISO>, launch_group_tape<S_SUZZ, T_GAUSS_DIAG>, launch_group_tape<S_SUZZ, T_BANANA>},
        {launch_group_tape<S_RHMC, T_GAUSS_ISO>, launch_group_tape<S_RHMC, T_GAUSS_DIAG>, launch_group_tape<S_RHMC, T_BANANA>},
        {launch_group_tape<S_STICKY, T_GAUSS_ISO>, launch_group_tape<S_STICKY, T_GAUSS_DIAG>, launch_group_tape<S_STICKY, T_BANANA>},
    };
    if (sampler < 0 || sampler > 6 || target < 0 || target > 2) return nullptr;
    return table[sampler][target];
}

// Lanes per chain for the group kernel: the smallest power of two that keeps a wave's LDS tile
// (2 * ceil(d/G) * 512 B) within `budget` bytes, i.e. at least 160 KiB / (4 * budget) blocks of four
// waves per CU.  0 = the configuration is served by the wave-per-chain kernel instead.
int choose_group_lanes(const pdmp_config_t &c) {
    if (c.target_id == PDMP_T_GAUSS_DENSE) return WAVE;        // one chain per wave: the gradient is a mat-vec
    // (round 5: grid_size is a run-time parameter of the LDS-form group kernel, 2 .. 64 -- UpperBound.jl:92,203; rounds 1-4 sent every
    //  grid_size other than 10 / 0 to the wave-per-chain kernel, which PDMP_GROUP_LANES=0 still selects)
    if (const char *e = getenv("PDMP_GROUP_LANES")) {
        const int G = atoi(e);
        if (G == 0) return 0;
        if (G >= 1 && G <= WAVE && (G & (G - 1)) == 0 && group_waves_per_block((c.dim + G - 1) / G, 2 * c.dim) >= 1) return G;
    }
    if ((c.sampler == PDMP_FECMC || (c.sampler == PDMP_STICKY && c.vectorized_bound && c.signed_bound)) &&   // (StickyZigZag: x, v, the active mask in VGPRs; 13 slots spill)
        c.deriv_mode == PDMP_DERIV_FORWARD && c.grid_size == GNG && c.target_id <= PDMP_T_BANANA) {
        // register form (round 4, inst_reg_fecmc_k*.hip): x, v in VGPRs, five LDS tiles of K x 64 doubles per wave (exchange, n, v_orth,
        // e1, e2) -- K <= 7 keeps two blocks of four waves per CU; exact-K kernels for K = 2 .. 7 at one lane per chain, 4 .. 7 otherwise
        for (int G = 1; G <= 16; G <<= 1) {
            const int K = (c.dim + G - 1) / G;
            if (K <= 7 && K >= (G == 1 ? 2 : 4)) return G;
        }
    }
    if (c.sampler == PDMP_SUZZ && c.vectorized_bound && c.signed_bound && c.deriv_mode == PDMP_DERIV_FORWARD && c.grid_size == GNG && c.target_id <= PDMP_T_BANANA) {
        // register form (round 4, inst_reg_suzz_k*.hip): exact-K kernels where they fit 256 registers -- K <= 7 (isotropic Gaussian), 6 (diagonal), 4 (banana)
        const int kmax = c.target_id == PDMP_T_GAUSS_ISO ? 7 : (c.target_id == PDMP_T_GAUSS_DIAG ? 6 : 4);
        // (measured at d = 50, 10^5 chains x 500, against the LDS form at G = 4: iso G = 8 / K = 7 55.9 -> 48.9 ms; banana G = 16 / K = 4 107.4 -> 76.6 ms;
        //  the diagonal Gaussian would need G = 16 / K = 4 there and loses, 59.3 -> 70.3 ms: at most 8 lanes per chain for it)
        const int gmax = c.target_id == PDMP_T_GAUSS_DIAG ? 8 : 16;
        for (int G = 1; G <= gmax; G <<= 1) {
            const int K = (c.dim + G - 1) / G;
            if (K <= kmax && K >= 2) return G;
        }
    }
    if (c.sampler == PDMP_FECMC) {         // six LDS tiles per wave (x, v, n, v_orth, e1, e2): 20 KB keep two blocks per CU (measured
        for (int G = 1; G <= WAVE; G <<= 1)   // at d = 50: G = 4 / 8 / 16 -> 467 / 566 / 306 ms); 40 KB for very large d
            if (6 * (size_t)((c.dim + G - 1) / G) * WAVE * sizeof(double) <= 20 * 1024) return G;
        return (6 * (size_t)((c.dim + WAVE - 1) / WAVE) * WAVE * sizeof(double) <= 40 * 1024) ? WAVE : 0;
    }
    // Zig-Zag at d <= 26 in the configuration the lane kernel serves (1 .. 7 slots of four): the G = 4 layout, so that large ensembles
    // run one lane per chain there as well (round 4: d = 26, 10^5 chains: 14.8 ms with G = 2 -> 10.3 ms; d <= 8 later in the round)
    // (grid_size = 10 only: the lane kernels serve no other grid, and at d <= 8 the G = 4 layout leaves lanes idle for nothing)
    if (c.sampler == PDMP_ZIGZAG && c.dim >= 1 && c.dim <= 26 && c.deriv_mode == PDMP_DERIV_FORWARD && c.vectorized_bound && c.signed_bound && c.grid_size == GNG &&
        (c.target_id == PDMP_T_GAUSS_ISO || c.target_id == PDMP_T_BANANA)) return 4;
    for (int G = 1; G <= 16; G <<= 1)      // register form (inst_reg_*.hip): at most KREG slots per lane, G <= 16
        if ((c.dim + G - 1) / G <= KREG) return G;
    if (c.sampler == PDMP_ZIGZAG && c.deriv_mode == PDMP_DERIV_FORWARD && c.vectorized_bound && c.signed_bound && c.grid_size == GNG && c.target_id <= PDMP_T_BANANA)
        for (int G = 32; G <= WAVE; G <<= 1)   // Zig-Zag, d = 209 .. 832 (round 4, last): register kernels with half a wave / a wave per chain
            if ((c.dim + G - 1) / G <= KREG) return G;
    const size_t budget = 20 * 1024;      // two blocks (8 waves) per CU
    for (int G = 1; G <= WAVE; G <<= 1) {
        const size_t K = (size_t)(c.dim + G - 1) / G;
        if (2 * K * WAVE * sizeof(double) <= budget) return G;
    }
    return 0;
}

// Which kernel family replays a tape for this handle: the group family (LDS form, register form, one lane per chain -- whichever
// the same launch would run with the Philox stream) for every sampler on the built-in separable / banana targets, else the
// wave-per-chain kernel (grid sizes other than 10: Zig-Zag / BPS / Boomerang), else none.
bool tape_on_group(const pdmp_engine *h) {
    if (h->cfg.target_id == PDMP_T_GAUSS_DENSE) return h->group_lanes > 0 && h->cfg.sampler <= PDMP_BOOMERANG;
    return h->group_lanes > 0 && h->cfg.target_id <= PDMP_T_BANANA && !h->policy.tape_wave;
}
bool tape_on_wave(const pdmp_engine *h) {
    return h->cfg.sampler <= PDMP_BOOMERANG && h->cfg.target_id <= PDMP_T_BANANA && h->cfg.grid_size >= 2 && h->cfg.dim <= 4 * WAVE;
}

int begin_timed(pdmp_engine *h, hipStream_t s, EventPair &ev) {
    if (!h->pool.empty()) { ev = h->pool.back(); h->pool.pop_back(); }
    else { HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); }
    HIP_TRY(hipEventRecord(ev.a, s));
    return PDMP_OK;
}
int end_timed(pdmp_engine *h, hipStream_t s, EventPair &ev) {
    HIP_TRY(hipEventRecord(ev.b, s));
    h->pending.push_back(ev);
    h->last_stream = s;
    return PDMP_OK;
}
int fold_timing(pdmp_engine *h) {
    for (auto &ev : h->pending) {
        HIP_TRY(hipEventSynchronize(ev.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        h->timing.kernel_ms_total += ms;
        h->timing.kernel_ms_last = ms;
        h->timing.launches++;
        h->pool.push_back(ev);
    }
    h->pending.clear();
    return PDMP_OK;
}

}  // namespace

extern "C" {

int pdmp_abi_version(void) { return PDMP_ABI_VERSION; }
// for the library's other translation units (multi_device.hip): one thread-local last-error buffer for the whole ABI
__attribute__((visibility("hidden"))) int pdmp_set_error_message(int code, const char *msg) { return fail(code, "%s", msg); }
const char *pdmp_last_error(void) { return g_err; }
int pdmp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int create_impl(const pdmp_config_t *cfg, const char *source, int32_t kind, pdmp_handle *out) {
    if (!cfg || !out) return fail(PDMP_ERR_ARGUMENT, "null argument");
    *out = nullptr;
    pdmp_config_t c = *cfg;
    if (source) { c.target_id = PDMP_T_USER; c.target_params = nullptr; c.n_target_params = 0; }
    if (int rc = normalize_config(c, source != nullptr)) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(PDMP_ERR_RUNTIME, "no HIP device visible: the PDMP engine has no CPU fallback");
    int dev = c.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(PDMP_ERR_ARGUMENT, "device %d out of range (%d visible)", dev, ndev);
    HIP_TRY(hipSetDevice(dev));
    if (choose_group_lanes(c) == 0 && c.grid_size == 0)
        return fail(PDMP_ERR_UNSUPPORTED, "grid_size == 0 (Brent constant bound) is served by the group kernel only");
    if (choose_group_lanes(c) == 0 && c.sampler >= PDMP_FECMC)
        return fail(PDMP_ERR_UNSUPPORTED, "ForwardECMC / SpeedUpZigZag / RHMC: dim too large for the LDS tiles of the group kernel");
    if (choose_group_lanes(c) == 0 && c.dim > 4 * WAVE)
        return fail(PDMP_ERR_UNSUPPORTED, "dim > %d needs grid_size = 10 (group kernel); the wave-per-chain kernel holds at most 4 coordinates per lane", 4 * WAVE);
    pdmp_engine *h = new pdmp_engine();
    h->cfg = c;
    h->cfg.device = dev;
    h->device = dev;
    h->cfg.target_params = nullptr;
    h->group_lanes = choose_group_lanes(c);
    h->policy = policy_from_env();
    if (c.target_id == PDMP_T_GAUSS_DIAG && c.target_params && c.n_target_params >= 2 * (int64_t)c.dim) {
        bool nonneg = true;
        for (int i = 0; i < c.dim; ++i) nonneg = nonneg && (c.target_params[c.dim + i] >= 0.0);     // (a NaN precision fails the test)
        h->tflags = nonneg ? 1 : 0;
    }
#define HIP_TRY_H(expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { pdmp_destroy(h); return fail(PDMP_ERR_RUNTIME, "%s: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)
    if (c.n_target_params > 0 && c.target_params) {
        HIP_TRY_H(hipMalloc(&h->d_tparams, (size_t)c.n_target_params * 8));
        HIP_TRY_H(hipMemcpy(h->d_tparams, c.target_params, (size_t)c.n_target_params * 8, hipMemcpyHostToDevice));
    }
    HIP_TRY_H(hipMalloc(&h->d_status, 8 * sizeof(int32_t)));      // [4]: the lane kernel's unit-speed flag
    HIP_TRY_H(hipMemset(h->d_status, 0, 8 * sizeof(int32_t)));
#undef HIP_TRY_H
    if (source) {
        if (h->group_lanes <= 0) { pdmp_destroy(h); return fail(PDMP_ERR_UNSUPPORTED, "potentials given as source: dim too large for the LDS tiles"); }
        static thread_local char log[sizeof g_err];
        log[0] = '\0';
        const int rc = pdmp_user_compile(dev, c.sampler, c.deriv_mode == PDMP_DERIV_FINITE_DIFF, kind, c.rng_version, (c.grid_size != GNG && c.grid_size != 0) ? 1 : 0, source, &h->user_fn, log, (int)sizeof log);
        if (rc != 0) {
            pdmp_destroy(h);
            // a potential that does not compile is the caller's mistake: the ArgumentError of ADBackend.jl:68-70
            return fail(rc == -1 ? PDMP_ERR_ARGUMENT : PDMP_ERR_RUNTIME, "potential source: %s", log);
        }
    }
    *out = h;
    return PDMP_OK;
}

int pdmp_create(const pdmp_config_t *cfg, pdmp_handle *out) { return create_impl(cfg, nullptr, 0, out); }

int pdmp_create_from_source(const pdmp_config_t *cfg, const char *source, int32_t kind, pdmp_handle *out) {
    if (!source) return fail(PDMP_ERR_ARGUMENT, "null source");
    if (kind != PDMP_SRC_POTENTIAL && kind != PDMP_SRC_GRADIENT) return fail(PDMP_ERR_ARGUMENT, "source kind must be PDMP_SRC_POTENTIAL or PDMP_SRC_GRADIENT");
    return create_impl(cfg, source, kind, out);
}

int pdmp_last_launch_form(pdmp_handle h) {
    if (!h) return -1;
    if (h->last_form != PDMP_FORM_LANE && h->last_form != PDMP_FORM_LANE_PAIR) return h->last_form;
    int32_t flag = 0;                              // the lane kernel ran unless the check kernel found a speed other than +-1
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->last_stream) != hipSuccess ||
        hipMemcpy(&flag, h->d_status + 4, sizeof flag, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return flag == 0 ? h->last_form : h->lane_fallback;
}

int pdmp_get_group_lanes(pdmp_handle h) { return h ? ((h->tape_chains > 0 && !tape_on_group(h)) ? 0 : h->group_lanes) : -1; }

// Lanes per chain for launches of C chains.  Round 2 measured the obvious idea -- more lanes per chain when a launch cannot fill
// the 2 048 resident wave slots (a 12 500-chain shard of the 8-GPU run is 782 waves at G = 4) -- and it does not pay: Zig-Zag d = 50,
// 12 500 chains x 1000 events: G = 4 / 8 / 16 -> 4.88 / 5.19 / 7.27 ms (25 000 chains: 9.1 / 8.5 ms; 50 000: 14.0 / 16.5 ms).
// An under-filled launch is bound by the latency of ONE wave's event loop (n_sk sequential events), and that is dominated by the
// per-chain scalar work (random stream, log, bound inversion, Categorical scan), which more lanes per chain replicate instead of
// shrinking.  So the suggestion is the default layout for every C; pdmp_set_group_lanes stays as the explicit knob.
int pdmp_suggest_group_lanes(pdmp_handle h, int64_t C) {
    if (!h) return -1;
    (void)C;
    return choose_group_lanes(h->cfg);
}

int pdmp_set_group_lanes(pdmp_handle h, int32_t G) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    const pdmp_config_t &c = h->cfg;
    const int def = choose_group_lanes(c);
    if (G == 0 || G == def) { if (h->group_lanes != def) h->until_valid = false; h->group_lanes = def; return PDMP_OK; }
    if (def <= 0 || c.target_id == PDMP_T_GAUSS_DENSE)
        return fail(PDMP_ERR_UNSUPPORTED, "this configuration does not run on the group kernel with a free lane count");
    if (G < 1 || G > WAVE || (G & (G - 1)) != 0) return fail(PDMP_ERR_ARGUMENT, "lanes per chain must be a power of two in 1..64. Current value: %d", (int)G);
    const int tiles = (c.sampler == PDMP_FECMC) ? 6 : 2;
    const int tpar = ((c.target_id == PDMP_T_GAUSS_DIAG) ? 2 * c.dim : 0) + ((c.sampler == PDMP_STICKY) ? c.dim : 0);
    if (group_waves_per_block((c.dim + G - 1) / G, tpar, tiles) < 1)
        return fail(PDMP_ERR_UNSUPPORTED, "dim %d with %d lanes per chain does not fit the LDS tiles", c.dim, (int)G);
    HIP_TRY(hipSetDevice(h->device));
    if (h->launched) HIP_TRY(hipStreamSynchronize(h->last_stream));
    h->group_lanes = G;
    h->until_valid = false;           // (a first chunk's hand-off states have the layout of the lanes per chain they were written with)
    return PDMP_OK;
}

int pdmp_reload_policy(pdmp_handle h) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    h->policy = policy_from_env();
    h->until_valid = false;
    return PDMP_OK;
}

int pdmp_clear_tape(pdmp_handle h) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    for (int q = 0; q < 3; ++q) { if (h->d_tape[q]) (void)hipFree(h->d_tape[q]); h->d_tape[q] = nullptr; h->tape_n[q] = 0; }
    h->tape_chains = 0;
    h->until_valid = false;
    return PDMP_OK;
}

int pdmp_set_tape(pdmp_handle h, int64_t C, int64_t n_exp, int64_t n_uni, int64_t n_nrm, const double *exp_tape,
                  const double *uni_tape, const double *nrm_tape) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (C <= 0 || n_exp <= 0 || n_uni <= 0 || n_nrm <= 0 || !exp_tape || !uni_tape || !nrm_tape)
        return fail(PDMP_ERR_ARGUMENT, "pdmp_set_tape needs C > 0 chains and three non-empty tapes");
    if (!tape_on_group(h) && !tape_on_wave(h))
        return fail(PDMP_ERR_UNSUPPORTED, "tape replay: the built-in separable / banana targets (group kernels: every sampler, grid_size = 10 or 0; "
                                          "wave-per-chain kernel: Zig-Zag / BPS / Boomerang, grid_size >= 2, dim <= %d)", 4 * WAVE);
    if (int rc = pdmp_clear_tape(h)) return rc;
    const double *src[3] = {exp_tape, uni_tape, nrm_tape};
    const int64_t n[3] = {n_exp, n_uni, n_nrm};
    for (int q = 0; q < 3; ++q) {
        HIP_TRY(hipMalloc(&h->d_tape[q], (size_t)C * (size_t)n[q] * 8));
        HIP_TRY(hipMemcpy(h->d_tape[q], src[q], (size_t)C * (size_t)n[q] * 8, hipMemcpyHostToDevice));
        h->tape_n[q] = n[q];
    }
    h->tape_chains = C;
    return PDMP_OK;
}

int pdmp_get_config(pdmp_handle h, pdmp_config_t *out) {
    if (!h || !out) return fail(PDMP_ERR_ARGUMENT, "null argument");
    *out = h->cfg;
    return PDMP_OK;
}

int pdmp_destroy(pdmp_handle h) {
    if (!h) return PDMP_OK;
    (void)hipSetDevice(h->device);
    (void)fold_timing(h);
    for (auto &ev : h->pool) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    if (h->order_ev) (void)hipEventDestroy(h->order_ev);
    if (h->d_tparams) (void)hipFree(h->d_tparams);
    if (h->d_status) (void)hipFree(h->d_status);
    if (h->d_sched) (void)hipFree(h->d_sched);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->dense_ws) (void)hipFree(h->dense_ws);
    if (h->dense_ctx) { pdmp_dense_destroy_context(h->dense_ctx); h->dense_ctx = nullptr; }
    for (int q = 0; q < 3; ++q) if (h->d_tape[q]) (void)hipFree(h->d_tape[q]);
    delete h;
    return PDMP_OK;
}

// the time-horizon variant's launch arguments (pdmp_sample_skeleton_until_device); null: the fixed-count path
struct UntilArgs { double T; int resume; const int64_t *chain_map; int64_t *count; int form; };

static int sample_skeleton_launch(pdmp_handle h, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                                  const double *xinit, const double *vinit, uint64_t seed, double *X, double *V,
                                  double *t, const pdmp_diag_t *diag, void *stream, const UntilArgs *until) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (n_sk <= 0) return fail(PDMP_ERR_ARGUMENT, "n_sk must be positive. Current value: %lld", (long long)n_sk);   // sample.jl:262-264
    if (C < 0) return fail(PDMP_ERR_ARGUMENT, "number of chains must be non-negative");
    if (C == 0) return PDMP_OK;
    const bool resuming = until && until->resume;
    if ((!resuming && (!xinit || !vinit)) || !X || !V || !t) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    if (n_sk > 0x7fffffffLL || (C + 3) / 4 > 0x7fffffffLL || C * (n_sk / 16 + 1) > 0x7fffffffLL) return fail(PDMP_ERR_UNSUPPORTED, "too many chains x events for one launch");
    HIP_TRY(hipSetDevice(h->device));
    SkelParams p;
    memset(&p, 0, sizeof p);
    const pdmp_config_t &c = h->cfg;
    p.d = c.dim; p.n_grid = c.grid_size; p.signed_bound = c.signed_bound; p.adaptive = c.adaptive;
    p.gaussian_velocity = c.gaussian_velocity; p.deriv_mode = c.deriv_mode; p.vectorized = c.vectorized_bound;
    p.C = C; p.n_sk = n_sk; p.chain0 = chain_id_offset; p.cap = c.max_iter_per_event; p.seed = seed;
    p.tmax = c.tmax; p.refresh = c.refresh_rate;
    p.fe_mix_p = c.fe_mix_p; p.fe_ran_p = c.fe_ran_p; p.fe_switch = c.fe_switch; p.fe_positive = c.fe_positive;
    p.rh_step_size = c.rh_step_size; p.rh_phi = c.rh_phi;
    p.fe_speed = (c.fe_speed_factor == 0.0) ? 1.0 : c.fe_speed_factor;
    p.xinit = xinit; p.vinit = vinit; p.tparams = h->d_tparams;
    p.X = X; p.V = V; p.t = t;
    if (diag) {
        p.H = diag->horizon; p.AR = diag->ar; p.EV = diag->error_value_ar;
        p.EB = diag->errored_bound; p.RJ = diag->rejected; p.HH = diag->hitting_horizon;
        p.IA = diag->is_active;
    }
    p.status = h->d_status;
    p.tflags = h->tflags | (h->policy.lane_debug << 8);
    p.t_stop = __builtin_huge_val();
    if (until) { p.t_stop = until->T; p.resume = until->resume; p.chain_map = until->chain_map; p.count = until->count; }
    const bool taped = h->tape_chains > 0;
    const bool v1 = c.rng_version == 1;        // arithmetic contract v1: the LDS-form group kernels and the wave-per-chain kernel of the v1 build (v1_bridge.hip)
    if (taped && v1) return fail(PDMP_ERR_UNSUPPORTED, "tape replay runs in the default arithmetic contract (rng_version = 2)");
    if (taped) {
        if (chain_id_offset < 0 || chain_id_offset + C > h->tape_chains)
            return fail(PDMP_ERR_ARGUMENT, "tape holds %lld chains; requested chains [%lld, %lld)", (long long)h->tape_chains,
                        (long long)chain_id_offset, (long long)(chain_id_offset + C));
        if (!tape_on_group(h) && !tape_on_wave(h))
            return fail(PDMP_ERR_UNSUPPORTED, "tape replay: built-in separable / banana targets, and Zig-Zag / BPS / Boomerang on the dense-precision target");
        p.tape_exp = h->d_tape[0]; p.tape_uni = h->d_tape[1]; p.tape_nrm = h->d_tape[2];
        p.tape_n_exp = h->tape_n[0]; p.tape_n_uni = h->tape_n[1]; p.tape_n_nrm = h->tape_n[2];
    }
#ifdef PDMP_STAMP
    static unsigned long long *d_stamps = nullptr;
    if (!d_stamps) { HIP_TRY(hipMalloc(&d_stamps, 64)); }
    HIP_TRY(hipMemset(d_stamps, 0, 64));
    p.stamps = d_stamps;
    g_stamps = d_stamps;
#endif
    hipStream_t s = (hipStream_t)stream;
    if (h->launched && s != h->last_stream) {       // the scheduler, hand-off and status buffers belong to the handle: a launch on
        if (!h->order_ev) HIP_TRY(hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming));   // another stream waits for the previous one
        HIP_TRY(hipEventRecord(h->order_ev, h->last_stream));
        HIP_TRY(hipStreamWaitEvent(s, h->order_ev, 0));
    }
    h->launched = true;
    GroupParams gp;
    memset(&gp, 0, sizeof gp);
    const bool tgroup = taped && tape_on_group(h);           // a replay on the group family: same forms, tapes as random source
    if (h->group_lanes > 0 && (!taped || tgroup)) {
        gp.s = p;
        gp.G = h->group_lanes;
        gp.K = (c.dim + gp.G - 1) / gp.G;
        gp.tpar = ((c.target_id == PDMP_T_GAUSS_DIAG) ? 2 * c.dim : 0) + ((c.sampler == PDMP_STICKY) ? c.dim : 0);   // mu, prec | kappa staged in LDS
        gp.tiles = (c.target_id == PDMP_T_GAUSS_DENSE || c.sampler == PDMP_FECMC) ? 6 : 2;   // ForwardECMC: n, v_orth, e1, e2
        // cells of a lazily built bound per pass of the state machine (same bits whatever the number).  A full machine (two waves per SIMD
        // hiding each other's latency) is fastest with instalments of three; a launch that leaves every wave alone on its SIMD pays each
        // extra pass of the state machine in full: all nine (12 500 chains: 4.29 -> 3.91 ms, scripts/budget_sweep.sh)
        gp.cell_budget = ((C + WAVE / h->group_lanes - 1) / (WAVE / h->group_lanes) <= 1024) ? (c.grid_size == GNG ? GNG - 1 : WAVE) : 3;   // (WAVE: every cell of any grid_size)
        gp.host_flags = (h->policy.no_lean ? 1 : 0) | (h->policy.debug_launch ? 2 : 0);
        if (h->policy.cell_budget > 0) gp.cell_budget = h->policy.cell_budget;   // tuning / tests
        const int64_t nb = (C + WAVE / gp.G - 1) / (WAVE / gp.G);
        const size_t need_sched = (size_t)nb + 1, need_state = (size_t)((C + WAVE - 1) / WAVE * WAVE) * ((size_t)lane_state_words(c.dim) + (c.sampler == PDMP_STICKY ? (size_t)gp.G : 0));   // (whole batches of 64: lane_kernel.hpp; StickyZigZag: + the active masks of a chain's lanes)
        if (resuming && (!h->until_valid || !h->d_state || need_state > h->state_cap || C > h->until_chains))
            return fail(PDMP_ERR_ARGUMENT, "resume: this handle holds no hand-off state of a first chunk of at least %lld chains", (long long)C);
        // the hand-off buffer is per handle: a resumed chunk must continue THE run that left it there (another seed, chain range or
        // lane layout would read another run's states, or states in another layout, without any error)
        if (resuming && (seed != h->until_seed || chain_id_offset != h->until_chain0 || h->group_lanes != h->until_lanes))
            return fail(PDMP_ERR_ARGUMENT, "resume: the hand-off state of this handle belongs to another run (seed %llu, chain_id_offset %lld, %d lanes per chain)",
                        (unsigned long long)h->until_seed, (long long)h->until_chain0, h->until_lanes);
        if (!resuming) { h->until_valid = until != nullptr; h->until_chains = C; h->until_seed = seed; h->until_chain0 = chain_id_offset; h->until_lanes = h->group_lanes; }
        if (need_sched > h->sched_cap) {
            if (h->d_sched) { HIP_TRY(hipStreamSynchronize(h->last_stream)); HIP_TRY(hipFree(h->d_sched)); h->d_sched = nullptr; h->sched_cap = 0; }
            HIP_TRY(hipMalloc(&h->d_sched, need_sched * sizeof(int32_t)));
            h->sched_cap = need_sched;
        }
        if (need_state > h->state_cap) {
            if (h->d_state) { HIP_TRY(hipStreamSynchronize(h->last_stream)); HIP_TRY(hipFree(h->d_state)); h->d_state = nullptr; h->state_cap = 0; }
            HIP_TRY(hipMalloc(&h->d_state, need_state * sizeof(double)));
            h->state_cap = need_state;
        }
        HIP_TRY(hipMemsetAsync(h->d_sched, 0, need_sched * sizeof(int32_t), s));
        if (h->policy.force_segments) gp.nseg = h->policy.force_segments;       // test hook, see group_schedule
        if (tgroup) gp.nseg = 1;                                                // the hand-off state does not carry tape positions
        gp.ctr = h->d_sched;
        gp.progress = h->d_sched + 1;
        gp.state = h->d_state;
    }
    EventPair ev;
    if (int rc = begin_timed(h, s, ev)) return rc;
    // Dense-precision Zig-Zag on many chains: grad-U of all chains as one fp64 GEMM per phase (dense_batched.hip);
    // small ensembles and PDMP_DENSE_MODE=matvec keep the bit-exact per-chain mat-vec kernel.
    const LaunchPolicy &pol = h->policy;
    const int dense_mode = pol.dense_mode;                    // 1 incr | 2 gemm | 3 matvec (0, default: incr for >= 64 chains and d <= 1024, gemm beyond)
    // (time horizon on the dense-precision target, round 4: the batched forms below -- their workspace holds every chain's state between
    //  chunks; a resumed chunk stays in the form of its first chunk whatever its size; the per-chain mat-vec kernel has no horizon)
    const bool dense_resume = resuming && until->form == PDMP_FORM_DENSE;
    const int64_t C_ws = dense_resume ? h->until_chains : C;       // chains of the workspace: the first chunk's
    const bool batched = c.target_id == PDMP_T_GAUSS_DENSE && c.sampler == PDMP_ZIGZAG && c.vectorized_bound && c.grid_size == GNG &&
                         c.deriv_mode == PDMP_DERIV_FORWARD && !taped && !v1 && 2 * C_ws < 0x7fffffffLL && (!resuming || dense_resume) &&
                         (dense_resume || (dense_mode ? (dense_mode == 2 || dense_mode == 1) : C >= 64)) && !(until && dense_mode == 2);
    const bool bps_batched = c.target_id == PDMP_T_GAUSS_DENSE && (c.sampler == PDMP_BPS || c.sampler == PDMP_BOOMERANG) && c.grid_size == GNG && c.dim <= (c.sampler == PDMP_BOOMERANG ? 24 : 32) * WAVE &&
                             c.deriv_mode == PDMP_DERIV_FORWARD && !taped && !v1 && 2 * C_ws < 0x7fffffffLL && (!resuming || dense_resume) &&
                             (dense_resume || (dense_mode ? (dense_mode == 2 || dense_mode == 1) : C >= 64));
    if (until && (taped || h->group_lanes <= 0 || (c.target_id == PDMP_T_GAUSS_DENSE && !batched && !bps_batched) ||
                  (batched && (c.dim > 16 * WAVE || (!c.signed_bound && c.dim > 8 * WAVE))) || (bps_batched && c.dim > 16 * WAVE))) {   // (the largest kernels: no horizon)
        h->pool.push_back(ev);
        return fail(PDMP_ERR_UNSUPPORTED, "the in-kernel time horizon runs on the group / lane kernels (grid_size = 10 or 0) and on the batched dense-precision forms "
                                          "(>= 64 chains, d <= 1024); no tape: rerun the fixed-count path with a doubled capacity instead");
    }
    const double t_stop = until ? until->T : __builtin_huge_val();
    if (batched || bps_batched) pdmp_dense_bind_context(&h->dense_ctx);   // (this thread's batched dense runs use THIS handle's rocBLAS handles and side stream)
    if (batched) {
        char msg[256] = "";
        int rc = -5;
        if (dense_mode != 2)                                   // P x, P v kept current incrementally: O(d) per gradient (dense_batched.hip)
            rc = pdmp_dense_incremental_run(h->device, c.dim, c.signed_bound, c.adaptive, c.tmax, c.max_iter_per_event, h->d_tparams, C_ws, n_sk,
                                            chain_id_offset, seed, xinit, vinit, X, V, t, p.H, p.AR, p.EB, p.EV, p.RJ, p.HH,
                                            h->d_status, stream, &h->dense_ws, &h->dense_ws_cap, pol.dense_resync, (pol.dense_gemm < 0 ? 1 : pol.dense_gemm) | (pol.dense_debug << 8),
                                            t_stop, resuming ? 1 : 0, C, until ? until->chain_map : nullptr, until ? until->count : nullptr, msg, (int)sizeof msg);
        if (rc == -5)                                           // d > 1024 (or forced): one GEMM per phase
            rc = pdmp_dense_batched_run(h->device, c.dim, c.signed_bound, c.adaptive, c.tmax, c.max_iter_per_event, h->d_tparams, C, n_sk,
                                        chain_id_offset, seed, xinit, vinit, X, V, t, p.H, p.AR, p.EB, p.EV, p.RJ, p.HH,
                                        h->d_status, stream, pol.dense_full_pv, pol.dense_gemm < 0 ? 1 : pol.dense_gemm, msg, (int)sizeof msg);
        if (rc != 0) { h->pool.push_back(ev); return fail(rc == -4 ? PDMP_ERR_ITERCAP : PDMP_ERR_RUNTIME, "dense batched path: %s", msg); }
        h->last_form = PDMP_FORM_DENSE;
        return end_timed(h, s, ev);
    }
    // BPS / Boomerang on a dense precision, many chains: the rate along the flow is a closed form in a few dot products with P x and P v
    // (BPS: <P x, v> + t <P v, v>; Boomerang: a quadratic form in cos t, sin t), so only an accepted event needs products with P --
    // P [x | v'] of ALL chains as one fp64 GEMM per skeleton column (dense_batched.hip: dr_bps_kernel, dr_boom_kernel)
    if (bps_batched) {
        char msg[256] = "";
        const int rc = pdmp_dense_bps_run(h->device, c.sampler == PDMP_BOOMERANG ? 1 : 0, c.dim, c.signed_bound, c.adaptive, c.gaussian_velocity, c.tmax, c.refresh_rate, c.max_iter_per_event,
                                          h->d_tparams, C_ws, n_sk, chain_id_offset, seed, xinit, vinit, X, V, t, p.H, p.AR, p.EB, p.EV, p.RJ, p.HH,
                                          h->d_status, stream, &h->dense_ws, &h->dense_ws_cap, pol.dense_resync, pol.dense_gemm < 0 ? 0 : pol.dense_gemm, pol.dense_lanes,
                                          t_stop, resuming ? 1 : 0, C, until ? until->chain_map : nullptr, until ? until->count : nullptr, msg, (int)sizeof msg);
        if (rc != 0) { h->pool.push_back(ev); return fail(PDMP_ERR_RUNTIME, "dense BPS path: %s", msg); }
        h->last_form = PDMP_FORM_DENSE;
        return end_timed(h, s, ev);
    }
    if (c.target_id == PDMP_T_USER) {
        if (taped || h->group_lanes <= 0 || !h->user_fn) { h->pool.push_back(ev); return fail(PDMP_ERR_UNSUPPORTED, "potentials given as source: no tape replay"); }
        h->last_form = PDMP_FORM_USER;
        const int e = pdmp_user_launch(h->user_fn, &gp, s);
        if (e != 0) { h->pool.push_back(ev); return fail(PDMP_ERR_RUNTIME, "launching the compiled potential: %s", hipGetErrorString((hipError_t)e)); }
    } else if (h->group_lanes > 0 && (!taped || tgroup)) {
        hipError_t e = hipErrorNotSupported;
        h->last_form = PDMP_FORM_GROUP_REG;
        // register form: the default bound flags only (grid_size = 10; Zig-Zag: vectorized signed bound) -- GroupChain::brent() etc.
        const bool reg_flags = c.sampler == PDMP_RHMC ||        // (RHMC: its bound is the constant refresh clock whatever the flags)
                               (c.grid_size == GNG && ((c.sampler != PDMP_ZIGZAG && c.sampler != PDMP_STICKY && c.sampler != PDMP_SUZZ) || (c.vectorized_bound && c.signed_bound)));
        if (pol.reg && reg_flags && !v1) {
            const bool fd = c.deriv_mode != 0;
            // Large Zig-Zag ensembles at d = 49..52 on the isotropic Gaussian / banana: one LANE per chain (lane_kernel.hpp), bit-identical
            // to the G = 4 group kernel.  It needs unit speeds: a check kernel raises a device flag otherwise, the lane kernel then
            // returns at once and the group kernel below (which returns at once while the flag is 0) does the launch's work.
            // measured (one MI355X, n_sk = 1000): up to 65 536 chains the lane kernel takes 9.6 ms (d = 50 Gaussian) / 11.0 ms (banana) / 7.8 ms
            // (d = 32) whatever the count -- one wave per SIMD walks its 64 chains' events one after the other -- while the group kernel grows
            // with the count: it wins below ~41 000 chains at d = 50 (9.4 ms at 40 000, 12.8 ms at 45 000) and below ~48 000 at d = 32
            // (round 3: between 16 385 chains and these thresholds the PAIR form below serves the launch; with it in the middle the one-lane form starts at 37 000 for d >= 45)
            // d = 53 .. 104 (the group kernel's G = 8): a PAIR of lanes per chain, 32 chains per wave, bit-identical to the G = 8 group kernel
            const bool lane2 = gp.G == 8;
            // (the pair kernel, d = 100, n_sk = 1000: 10.6 ms up to 32 768 chains -- one wave per SIMD -- against 9.5 ms of the group kernel at 25 000, 37.9 ms at 10^5)
            const int64_t lane_from = pol.lane_min_chains >= 0 ? pol.lane_min_chains : pol.lane_from_default(lane2, c.dim, n_sk);   // (d = 32, 44 000 chains: one lane 6.42 ms, pair 7.34, group kernel 7.43; 30 000: pair 4.99, group 5.92)
            // d = 25 .. 52 again, launches between ~a fifth and ~two fifths of the machine (the 4-GPU shard of the headline ensemble): a pair of
            // lanes per chain with two coordinates of a slot each -- GROUP(4) still -- so that 25 000 chains are 782 waves instead of 391
            // measured (d = 50 Gaussian, n_sk = 1000; group kernel / one lane / pair): 12 500 chains 4.31 / 8.36 / 5.36 ms, 18 000 6.58 / 8.41 / 5.39,
            // 25 000 8.42 / 8.43 / 5.45, 32 000 10.27 / 8.47 / 7.02, 40 000 10.99 / 8.48 / 9.60, 50 000 11.53 / 8.57 / 11.50 (profiles/r3/r3q1_*)
            const int64_t pair_from = pol.pair_min_chains;   // (up to 16 384 chains = 1 024 waves of the group kernel, one per SIMD: 3.9 - 4.0 ms flat with the whole bound per pass; 17 000 chains: 6.1 ms against the pair form's 5.4)
            const bool lane_ok = c.sampler == PDMP_ZIGZAG && !fd && (c.target_id == PDMP_T_GAUSS_ISO || c.target_id == PDMP_T_BANANA) &&
                                 ((gp.G == 4 && c.dim >= 1 && c.dim <= 52) || (lane2 && c.dim >= 53 && c.dim <= 104 && !tgroup)) &&
                                 (int64_t)WAVE * n_sk * c.dim * 8 < (lane2 ? 2 : 1) * 0x7fffffffLL;   // (a batch's window of X: 31-bit byte offsets; 64 chains, or 32 of a pair form -- the quad pair is held to the window of 64 so that a resumed chunk may be of either size)
            // (round 3: no alignment preconditions any more -- a chain's rows may start anywhere inside a 64-byte line, lane_kernel.hpp: open_window)
            // a resumed chunk continues in the form whose state layout its first chunk left behind (until->form), whatever its size
            const int lane_mode = pol.lane_mode;                       // (2: the pair form at d <= 52 too, whatever the count -- tests)
            const bool pairq = !lane2 && !tgroup && (resuming ? until->form == PDMP_FORM_LANE_PAIR
                                                              : (lane_mode == 2 || (lane_mode < 0 && C >= pair_from && C < lane_from && C <= 32768)));   // (d <= 20, 10^5 chains: one lane 8.9 - 9.3 ms, pair 9.5 - 9.8)
            const bool lane_wanted = resuming ? ((until->form == PDMP_FORM_LANE || until->form == PDMP_FORM_LANE_PAIR) && c.sampler == PDMP_ZIGZAG)   // (BPS: its own lane form, below)
                                              : (lane_mode != 0 && (lane_mode == 1 || lane_mode == 2 || C >= lane_from || pairq));
            if (resuming && lane_wanted && !lane_ok) {
                h->pool.push_back(ev);
                return fail(PDMP_ERR_ARGUMENT, "resume: the first chunk ran one lane per chain; a chunk's capacity must keep 64 * capacity * dim * 8 below 2^31 bytes");
            }
            if (lane_ok && lane_wanted) {
                int32_t *flag = h->d_status + 4;
                hipError_t le = hipMemsetAsync(flag, 0, sizeof(int32_t), s);
                if (le == hipSuccess && !resuming) le = launch_lane_check(vinit, C * (int64_t)c.dim, flag, s);   // (a resumed chunk: the first chunk's check stands)
                const int rounds = pol.lane_rounds;          // work units per resident wave
                GroupParams lp = gp;
                if (pol.cell_budget == 0) lp.cell_budget = c.dim <= 16 ? GNG - 1 : 5;   // a pass of 64 chains costs ~2 000 instructions, a cell 240 at d = 50: wait for more cells;
                                                                                     // d <= 16, where a cell is a few dozen: the whole bound in one pass (d = 2 / 10 / 16, 10^5 chains: 10.4 / 8.5 / 9.7 -> 8.4 / 7.9 / 9.4 ms; scripts/r4_short_rows3.sh)
                if (tgroup) lp.nseg = 1;                     // (lane_schedule: a forced segment count)
                if (le == hipSuccess) le = tgroup ? launch_lane_zigzag_tape(c.target_id, lp, flag, rounds, s)   // (d = 49 .. 52; else the group kernel below replays)
                                      : pairq ? (c.dim > 24 ? launch_lane2q_zigzag(c.target_id, lp, flag, rounds, s) : c.dim > 8 ? launch_lane2q_zigzag_c(c.target_id, lp, flag, rounds, s) : launch_lane2q_zigzag_d(c.target_id, lp, flag, rounds, s))
                                      : c.dim > 96 ? launch_lane2_zigzag(c.target_id, lp, flag, rounds, s)
                                      : c.dim > 72 ? launch_lane2_zigzag_b(c.target_id, lp, flag, rounds, s)
                                      : c.dim > 52 ? launch_lane2_zigzag_a(c.target_id, lp, flag, rounds, s)
                                      : c.dim > 48 ? launch_lane_zigzag(c.target_id, lp, flag, rounds, s)
                                      : c.dim > 36 ? launch_lane_zigzag_b(c.target_id, lp, flag, rounds, s)
                                      : c.dim > 24 ? launch_lane_zigzag_a(c.target_id, lp, flag, rounds, s)
                                      : c.dim > 8 ? launch_lane_zigzag_c(c.target_id, lp, flag, rounds, s) : launch_lane_zigzag_d(c.target_id, lp, flag, rounds, s);
                if (le == hipSuccess) {                      // (a lane kernel that cannot be launched -- it asks for 145 KB of LDS per block --
                    h->lane_fallback = PDMP_FORM_GROUP_REG;
                    gp.only_if = flag;                       //  leaves the launch to the group kernel below)
                    h->last_form = (lane2 || pairq) ? PDMP_FORM_LANE_PAIR : PDMP_FORM_LANE;
                } else (void)hipGetLastError();
            }
            if (tgroup)                                     // a replay: the 13-slot register kernels with the tapes as random source
                e = c.sampler == PDMP_ZIGZAG    ? (fd ? hipErrorNotSupported : launch_group_reg_tape_zigzag(c.target_id, fd, gp, s))
                  : c.sampler == PDMP_BPS       ? launch_group_reg_tape_bps(c.target_id, fd, gp, s)
                  : c.sampler == PDMP_BOOMERANG ? launch_group_reg_tape_boomerang(c.target_id, fd, gp, s)
                                                : hipErrorNotSupported;
            else if (pol.reg_exact)                         // a kernel of exactly K slots where one exists
                e = c.sampler == PDMP_BOOMERANG    ? (fd ? launch_group_reg_boomerang_exact_fd(c.target_id, gp, s)
                                                         : launch_group_reg_boomerang_exact(c.target_id, gp, s))
                  : fd                             ? hipErrorNotSupported
                  : c.sampler == PDMP_ZIGZAG       ? launch_group_reg_zigzag_exact(c.target_id, gp, s)
                  : c.sampler == PDMP_BPS          ? launch_group_reg_bps_exact(c.target_id, gp, s)
                  : c.sampler == PDMP_FECMC        ? launch_group_reg_fecmc_exact(c.target_id, gp, s)           // round 4: K <= 7
                  : c.sampler == PDMP_STICKY       ? launch_group_reg_sticky_exact(c.target_id, gp, s)
                  : c.sampler == PDMP_SUZZ         ? launch_group_reg_suzz_exact(c.target_id, gp, s) : hipErrorNotSupported;
            if (e == hipErrorNotSupported && !tgroup)
                e = c.sampler == PDMP_ZIGZAG    ? launch_group_reg_zigzag(c.target_id, fd, gp, s)
                  : c.sampler == PDMP_BPS       ? launch_group_reg_bps(c.target_id, fd, gp, s)
                  : c.sampler == PDMP_BOOMERANG ? launch_group_reg_boomerang(c.target_id, fd, gp, s)
                  : c.sampler == PDMP_RHMC      ? launch_group_reg_rhmc(c.target_id, gp, s)      // round 4: FULL 13-slot kernels (iso, banana), partial ones (diag)
                                                : hipErrorNotSupported;
        }
        if (e == hipErrorNotSupported) {
            if (gp.only_if) h->lane_fallback = PDMP_FORM_GROUP;      // (behind a lane kernel: last_form stays the lane form; d <= 4 at G = 4 has no register form)
            else h->last_form = PDMP_FORM_GROUP;
            const glaunch_fn f = tgroup ? pick_group_tape_launcher(c.sampler, c.target_id) : pick_group_launcher(c.sampler, c.target_id);
            if (!f) { h->pool.push_back(ev); return fail(PDMP_ERR_UNSUPPORTED, "no group kernel for sampler %d on target %d", c.sampler, c.target_id); }
            e = v1 ? v1_launch_group(c.sampler, c.target_id, &gp, s) : f(gp, s);
        }
        if (e != hipSuccess) { h->pool.push_back(ev); return fail(PDMP_ERR_RUNTIME, "group kernel launch: %s", hipGetErrorString(e)); }
    } else {
        h->last_form = PDMP_FORM_WAVE;
        const launch_fn f = pick_launcher(c.sampler, c.target_id);
        if (!f) { h->pool.push_back(ev); return fail(PDMP_ERR_UNSUPPORTED, "the wave-per-chain kernel serves Zig-Zag / BPS / Boomerang on the built-in separable / banana targets"); }
        const hipError_t e = v1 ? v1_launch_skeleton(c.sampler, c.target_id, &p, s) : f(p, s);
        if (e != hipSuccess) { h->pool.push_back(ev); return fail(PDMP_ERR_RUNTIME, "wave kernel launch: %s", hipGetErrorString(e)); }
    }
    return end_timed(h, s, ev);
}

int pdmp_sample_skeleton_device(pdmp_handle h, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                                const double *xinit, const double *vinit, uint64_t seed, double *X, double *V,
                                double *t, const pdmp_diag_t *diag, void *stream) {
    return sample_skeleton_launch(h, C, n_sk, chain_id_offset, xinit, vinit, seed, X, V, t, diag, stream, nullptr);
}

// ---- sample_skeleton(sampler, T::Float64, ...) (src/sample.jl:323-439) as a kernel feature: one chunk of the history.
int pdmp_sample_skeleton_until_device(pdmp_handle h, int64_t C, int64_t capacity, int64_t chain_id_offset, const double *xinit,
                                      const double *vinit, uint64_t seed, double T, int32_t resume, const int64_t *chain_map,
                                      double *X, double *V, double *t, const pdmp_diag_t *diag, int64_t *n_recorded, void *stream) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (!(T >= 0.0) || !std::isfinite(T)) return fail(PDMP_ERR_ARGUMENT, "T must be finite and non-negative. Current value: %g", T);   // sample.jl:335-337
    if (capacity < 2) return fail(PDMP_ERR_ARGUMENT, "a chunk needs a capacity of at least 2 columns. Current value: %lld", (long long)capacity);
    if (!n_recorded) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    if (!resume && chain_map) return fail(PDMP_ERR_ARGUMENT, "a chain map belongs to a resumed chunk");
    UntilArgs u = {T, resume ? 1 : 0, chain_map, n_recorded, 0};
    // a resumed chunk continues in the kernel form whose state layout the previous chunk left behind: the lane kernel's, unless its
    // unit-speed check sent that launch to the group kernel (known once it has run: pdmp_last_launch_form waits for it -- the host
    // has waited for the previous chunk anyway, to learn how many chains are left)
    if (resume) u.form = pdmp_last_launch_form(h);
    const int rc = sample_skeleton_launch(h, C, capacity, chain_id_offset, xinit, vinit, seed, X, V, t, diag, stream, &u);
    return rc;
}

int pdmp_close_chunk_device(pdmp_handle h, int64_t C, int64_t capacity, double *X, double *V, double *t, const pdmp_diag_t *diag,
                            double T, const int64_t *n_recorded, int64_t *count, void *stream) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (h->cfg.sampler == PDMP_RHMC && (h->cfg.dim > 4 * WAVE || h->cfg.target_id > PDMP_T_BANANA))
        return fail(PDMP_ERR_UNSUPPORTED, "RHMC: the flow is integrated with up to 4 coordinates per lane (dim <= 256) on the built-in targets");
    if (!(T >= 0.0) || !std::isfinite(T)) return fail(PDMP_ERR_ARGUMENT, "T must be finite and non-negative. Current value: %g", T);
    if (capacity <= 0 || C < 0) return fail(PDMP_ERR_ARGUMENT, "capacity must be positive and C non-negative");
    if (C == 0) return PDMP_OK;
    if (!X || !V || !t || !count || !n_recorded) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(256), grid((unsigned)((C + 3) / 4));
    pdmp_diag_t dg = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (diag) dg = *diag;
    if (h->cfg.sampler == PDMP_BOOMERANG)
        hipLaunchKernelGGL((h->cfg.rng_version == 1 ? close_at_kernel<2> : close_at_kernel<1>), grid, block, 0, s, h->cfg.dim, C, capacity, X, V, t, T, count, 0, 0, 0.0, (const double *)nullptr, dg, n_recorded, 2);
    else
        hipLaunchKernelGGL(close_at_kernel<0>, grid, block, 0, s, h->cfg.dim, C, capacity, X, V, t, T, count,
                           h->cfg.sampler == PDMP_SUZZ ? 1 : (h->cfg.sampler == PDMP_RHMC ? 2 : (h->cfg.sampler == PDMP_STICKY ? 3 : 0)),
                           h->cfg.target_id, h->cfg.rh_step_size, (const double *)h->d_tparams, dg, n_recorded, (h->cfg.rng_version == 1 ? 1 : 2));
    HIP_TRY(hipGetLastError());
    h->last_stream = s;
    return PDMP_OK;
}

int pdmp_until_advance_device(pdmp_handle h, int64_t C, const int64_t *count, int64_t skip, const int64_t *map_in, int64_t *total,
                              int64_t *start, int64_t *map_out, int64_t *n_active, void *stream) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (C < 0 || skip < 0) return fail(PDMP_ERR_ARGUMENT, "C and skip must be non-negative");
    if (!count || !total || !start || !map_out || !n_active) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(until_advance_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, C, count, skip, map_in, total, start, map_out, n_active);
    HIP_TRY(hipGetLastError());
    h->last_stream = (hipStream_t)stream;
    return PDMP_OK;
}

int pdmp_compact_chunk_device(pdmp_handle h, int64_t C, int64_t capacity, const int64_t *count, int64_t skip, const int64_t *chain_map,
                              const int64_t *start, const int64_t *offsets, const double *X, const double *V, const double *t,
                              const pdmp_diag_t *diag, double *X_out, double *V_out, double *t_out, const pdmp_diag_t *diag_out, void *stream) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (capacity <= 0 || C < 0 || skip < 0) return fail(PDMP_ERR_ARGUMENT, "capacity must be positive, C and skip non-negative");
    if (C == 0) return PDMP_OK;
    if (!count || !start || !offsets || !X || !V || !t || !X_out || !V_out || !t_out) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    HIP_TRY(hipSetDevice(h->device));
    const int64_t tiles = (capacity + WAVE - 1) / WAVE, blocks = (C * tiles + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail(PDMP_ERR_UNSUPPORTED, "too many columns for one launch");
    pdmp_diag_t dg = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, og = dg;
    if (diag && diag_out) { dg = *diag; og = *diag_out; }
    hipStream_t s = (hipStream_t)stream;
    EventPair ev;
    if (int rc = begin_timed(h, s, ev)) return rc;
    hipLaunchKernelGGL(compact_chunk_kernel, dim3((unsigned)blocks), dim3(256), 0, s, h->cfg.dim, C, capacity, count, skip, chain_map, start, offsets,
                       X, V, t, dg, X_out, V_out, t_out, og);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) { h->pool.push_back(ev); return fail(PDMP_ERR_RUNTIME, "compact_chunk_kernel: %s", hipGetErrorString(e)); }
    return end_timed(h, s, ev);
}

int pdmp_check_status(pdmp_handle h, int64_t *n_guard) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    int32_t st[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(st, h->d_status, sizeof st, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(h->d_status, 0, sizeof st));
    if (n_guard) *n_guard = st[1];
    if (st[2] > 0) return fail(PDMP_ERR_RUNTIME, "%d work unit(s) lost their predecessor's hand-off", st[2]);
    if (st[3] > 0) return fail(PDMP_ERR_ARGUMENT, "%d chain(s) exhausted their random-variate tape", st[3]);
    if (st[0] > 0) return fail(PDMP_ERR_ITERCAP, "%d chain(s) hit the per-event iteration cap (%lld)", st[0],
                               (long long)h->cfg.max_iter_per_event);
    return PDMP_OK;
}

int pdmp_sample_skeleton(pdmp_handle h, int64_t C, int64_t n_sk, int64_t chain_id_offset, const double *xinit,
                         const double *vinit, uint64_t seed, double *X, double *V, double *t,
                         const pdmp_diag_t *diag) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (n_sk <= 0) return fail(PDMP_ERR_ARGUMENT, "n_sk must be positive. Current value: %lld", (long long)n_sk);
    if (C < 0) return fail(PDMP_ERR_ARGUMENT, "number of chains must be non-negative");
    if (C == 0) return PDMP_OK;
    if (!xinit || !vinit || !X || !V || !t) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    HIP_TRY(hipSetDevice(h->device));
    const size_t d = (size_t)h->cfg.dim, ne = (size_t)C * (size_t)n_sk;
    double *dx = nullptr, *dv = nullptr, *dX = nullptr, *dV = nullptr, *dt = nullptr;
    double *dH = nullptr, *dAR = nullptr, *dEV = nullptr;
    int32_t *dEB = nullptr, *dRJ = nullptr, *dHH = nullptr;
    uint8_t *dIA = nullptr;
    int rc = PDMP_OK;
    auto cleanup = [&]() {
        (void)hipFree(dx); (void)hipFree(dv); (void)hipFree(dX); (void)hipFree(dV); (void)hipFree(dt);
        (void)hipFree(dH); (void)hipFree(dAR); (void)hipFree(dEV); (void)hipFree(dEB); (void)hipFree(dRJ); (void)hipFree(dHH); (void)hipFree(dIA);
    };
#define TRY_OR_CLEAN(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(PDMP_ERR_RUNTIME, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
    TRY_OR_CLEAN(hipMalloc(&dx, (size_t)C * d * 8));
    TRY_OR_CLEAN(hipMalloc(&dv, (size_t)C * d * 8));
    TRY_OR_CLEAN(hipMalloc(&dX, ne * d * 8));
    TRY_OR_CLEAN(hipMalloc(&dV, ne * d * 8));
    TRY_OR_CLEAN(hipMalloc(&dt, ne * 8));
    pdmp_diag_t dd = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (diag) {
        if (diag->horizon) { TRY_OR_CLEAN(hipMalloc(&dH, ne * 8)); dd.horizon = dH; }
        if (diag->ar) { TRY_OR_CLEAN(hipMalloc(&dAR, ne * 8)); dd.ar = dAR; }
        if (diag->error_value_ar) { TRY_OR_CLEAN(hipMalloc(&dEV, ne * 40)); dd.error_value_ar = dEV; }
        if (diag->errored_bound) { TRY_OR_CLEAN(hipMalloc(&dEB, ne * 4)); dd.errored_bound = dEB; }
        if (diag->rejected) { TRY_OR_CLEAN(hipMalloc(&dRJ, ne * 4)); dd.rejected = dRJ; }
        if (diag->hitting_horizon) { TRY_OR_CLEAN(hipMalloc(&dHH, ne * 4)); dd.hitting_horizon = dHH; }
        if (diag->is_active) { TRY_OR_CLEAN(hipMalloc(&dIA, ne * d)); TRY_OR_CLEAN(hipMemset(dIA, 1, ne * d)); dd.is_active = dIA; }
    }
    TRY_OR_CLEAN(hipMemcpy(dx, xinit, (size_t)C * d * 8, hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(dv, vinit, (size_t)C * d * 8, hipMemcpyHostToDevice));
    rc = pdmp_sample_skeleton_device(h, C, n_sk, chain_id_offset, dx, dv, seed, dX, dV, dt, diag ? &dd : nullptr, nullptr);
    if (rc == PDMP_OK) rc = pdmp_check_status(h, nullptr);
    if (rc == PDMP_OK || rc == PDMP_ERR_ITERCAP) {
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        TRY_OR_CLEAN(hipMemcpy(X, dX, ne * d * 8, hipMemcpyDeviceToHost));
        TRY_OR_CLEAN(hipMemcpy(V, dV, ne * d * 8, hipMemcpyDeviceToHost));
        TRY_OR_CLEAN(hipMemcpy(t, dt, ne * 8, hipMemcpyDeviceToHost));
        if (dH) TRY_OR_CLEAN(hipMemcpy(diag->horizon, dH, ne * 8, hipMemcpyDeviceToHost));
        if (dAR) TRY_OR_CLEAN(hipMemcpy(diag->ar, dAR, ne * 8, hipMemcpyDeviceToHost));
        if (dEV) TRY_OR_CLEAN(hipMemcpy(diag->error_value_ar, dEV, ne * 40, hipMemcpyDeviceToHost));
        if (dEB) TRY_OR_CLEAN(hipMemcpy(diag->errored_bound, dEB, ne * 4, hipMemcpyDeviceToHost));
        if (dRJ) TRY_OR_CLEAN(hipMemcpy(diag->rejected, dRJ, ne * 4, hipMemcpyDeviceToHost));
        if (dHH) TRY_OR_CLEAN(hipMemcpy(diag->hitting_horizon, dHH, ne * 4, hipMemcpyDeviceToHost));
        if (dIA) TRY_OR_CLEAN(hipMemcpy(diag->is_active, dIA, ne * d, hipMemcpyDeviceToHost));
        memcpy(g_err, keep, sizeof keep);
    }
    cleanup();
    return rc;
}

// ---- sample_from_skeleton: one launcher behind the three reference methods
// method 0: N::Int (sample.jl:475-513 / StickyPDMP :516-561); 1: dt::Float64 (:573-646); 2: (N::Int, dt::Float64) (:649-682)
static int from_skeleton_launch(pdmp_handle h, int method, int64_t C, int64_t n_sk, const double *X, const double *V, const double *t,
                                const uint8_t *is_active, const int64_t *lens, int64_t n_prefix, int64_t N, double dt, int32_t discard_vt,
                                double *out, int64_t *counts, void *stream) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (h->cfg.sampler == PDMP_RHMC && (h->cfg.dim > 4 * WAVE || h->cfg.target_id > PDMP_T_BANANA))
        return fail(PDMP_ERR_UNSUPPORTED, "RHMC: sample_from_skeleton integrates the flow with up to 4 coordinates per lane (dim <= 256) on the built-in targets");
    if (method == 0 && dt != 0.0) return fail(PDMP_ERR_ARGUMENT, "the N::Int method takes no dt");
    if (method != 0 && (!(dt > 0.0) || !std::isfinite(dt))) return fail(PDMP_ERR_ARGUMENT, "dt must be finite and positive. Current value: %g", dt);
    if (N <= 0) return fail(PDMP_ERR_ARGUMENT, "N must be positive. Current value: %lld", (long long)N);   // sample.jl:476-478
    if (n_sk <= 0) return fail(PDMP_ERR_ARGUMENT, "empty history");
    if (method == 2 && (n_prefix <= 0 || n_prefix > n_sk)) return fail(PDMP_ERR_ARGUMENT, "the (N, dt) method needs 1 <= N <= length(history). Current value: %lld", (long long)n_prefix);
    if (C < 0) return fail(PDMP_ERR_ARGUMENT, "number of chains must be non-negative");
    if (C == 0) return PDMP_OK;
    if (!X || !V || !t || !out) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    if (h->cfg.sampler == PDMP_STICKY && !is_active)                                                         // StickyPDMP methods read history.is_active
        return fail(PDMP_ERR_ARGUMENT, "StickyZigZag: sample_from_skeleton needs PDMPHistory.is_active (sample.jl:516-561)");
    HIP_TRY(hipSetDevice(h->device));
    const int64_t tiles = (N + WAVE - 1) / WAVE, waves = C * tiles, blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail(PDMP_ERR_UNSUPPORTED, "too many samples for one launch");
    // is_active is consulted by the StickyPDMP methods (select) and by the (N, dt) method of every sampler (v .* active, :667)
    const int ia_mode = !is_active ? 0 : (method == 2 ? 2 : (h->cfg.sampler == PDMP_STICKY ? 1 : 0));
    hipStream_t s = (hipStream_t)stream;
    EventPair ev;
    if (int rc = begin_timed(h, s, ev)) return rc;
    const bool flat = h->cfg.sampler != PDMP_SUZZ && h->cfg.sampler != PDMP_RHMC && (2 * (int64_t)h->cfg.dim + 1) * WAVE < (1ll << 31) / (2 * (int64_t)h->cfg.dim + 1) &&
                      !h->policy.from_skeleton_rows;                 // (policy: the row-per-iteration kernel, for A/B runs)
    if (flat && h->cfg.sampler == PDMP_BOOMERANG)
        hipLaunchKernelGGL((h->cfg.rng_version == 1 ? from_skeleton_flat_kernel<2> : from_skeleton_flat_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, h->cfg.dim, C, n_sk, X, V, t, N, dt, (int)discard_vt, out,
                           is_active, ia_mode, (int)(method == 2), n_prefix, lens, counts);
    else if (flat)
        hipLaunchKernelGGL(from_skeleton_flat_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, h->cfg.dim, C, n_sk, X, V, t, N, dt, (int)discard_vt, out,
                           is_active, ia_mode, (int)(method == 2), n_prefix, lens, counts);
    else if (h->cfg.sampler == PDMP_BOOMERANG)
        hipLaunchKernelGGL((h->cfg.rng_version == 1 ? from_skeleton_kernel<2> : from_skeleton_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, h->cfg.dim, C, n_sk, X, V, t, N, dt, (int)discard_vt, out, 0,
                           -1, 0.0, (const double *)nullptr, is_active, ia_mode, (int)(method == 2), n_prefix, lens, counts, 2);
    else
        hipLaunchKernelGGL(from_skeleton_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, h->cfg.dim, C, n_sk, X, V, t, N, dt, (int)discard_vt, out,
                           (int)(h->cfg.sampler == PDMP_SUZZ), h->cfg.sampler == PDMP_RHMC ? h->cfg.target_id : -1, h->cfg.rh_step_size,
                           (const double *)h->d_tparams, is_active, ia_mode, (int)(method == 2), n_prefix, lens, counts, (h->cfg.rng_version == 1 ? 1 : 2));
    if (hipError_t e = hipGetLastError(); e != hipSuccess) { h->pool.push_back(ev); return fail(PDMP_ERR_RUNTIME, "from_skeleton_kernel: %s", hipGetErrorString(e)); }
    return end_timed(h, s, ev);
}

int pdmp_sample_from_skeleton_device(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                                     const double *t, const uint8_t *is_active, const int64_t *lens, int64_t N, int32_t discard_vt,
                                     double *out, void *stream) {
    return from_skeleton_launch(h, 0, C, n_sk, X, V, t, is_active, lens, 0, N, 0.0, discard_vt, out, nullptr, stream);
}

int pdmp_sample_from_skeleton_dt_device(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                                        const double *t, const uint8_t *is_active, const int64_t *lens, int64_t N, double dt,
                                        int32_t discard_vt, double *out, int64_t *counts, void *stream) {
    return from_skeleton_launch(h, 1, C, n_sk, X, V, t, is_active, lens, 0, N, dt, discard_vt, out, counts, stream);
}

int pdmp_sample_from_skeleton_prefix_device(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                                            const double *t, const uint8_t *is_active, int64_t n_prefix, double dt, int64_t M,
                                            int32_t discard_vt, double *out, int64_t *counts, void *stream) {
    return from_skeleton_launch(h, 2, C, n_sk, X, V, t, is_active, nullptr, n_prefix, M, dt, discard_vt, out, counts, stream);
}

// host-pointer forms: stage the history on the device, run the device form, copy the samples back
static int from_skeleton_host(pdmp_handle h, int method, int64_t C, int64_t n_sk, const double *X, const double *V, const double *t,
                              const uint8_t *is_active, int64_t n_prefix, int64_t N, double dt, int32_t discard_vt, double *out,
                              int64_t *counts) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (N <= 0) return fail(PDMP_ERR_ARGUMENT, "N must be positive. Current value: %lld", (long long)N);
    if (n_sk <= 0) return fail(PDMP_ERR_ARGUMENT, "empty history");
    if (C <= 0) return C == 0 ? PDMP_OK : fail(PDMP_ERR_ARGUMENT, "number of chains must be non-negative");
    if (!X || !V || !t || !out) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    HIP_TRY(hipSetDevice(h->device));
    const size_t d = (size_t)h->cfg.dim, ne = (size_t)C * (size_t)n_sk;
    const size_t rows = discard_vt ? d : 2 * d + 1, no = (size_t)C * (size_t)N * rows;
    double *dX = nullptr, *dV = nullptr, *dt_buf = nullptr, *dout = nullptr;
    uint8_t *dIA = nullptr;
    int64_t *dcnt = nullptr;
    auto cleanup = [&]() { (void)hipFree(dX); (void)hipFree(dV); (void)hipFree(dt_buf); (void)hipFree(dout); (void)hipFree(dIA); (void)hipFree(dcnt); };
    TRY_OR_CLEAN(hipMalloc(&dX, ne * d * 8));
    TRY_OR_CLEAN(hipMalloc(&dV, ne * d * 8));
    TRY_OR_CLEAN(hipMalloc(&dt_buf, ne * 8));
    TRY_OR_CLEAN(hipMalloc(&dout, no * 8));
    TRY_OR_CLEAN(hipMemcpy(dX, X, ne * d * 8, hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(dV, V, ne * d * 8, hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(dt_buf, t, ne * 8, hipMemcpyHostToDevice));
    if (is_active) { TRY_OR_CLEAN(hipMalloc(&dIA, ne * d)); TRY_OR_CLEAN(hipMemcpy(dIA, is_active, ne * d, hipMemcpyHostToDevice)); }
    if (counts) {
        TRY_OR_CLEAN(hipMalloc(&dcnt, (size_t)C * 8));
        TRY_OR_CLEAN(hipMemset(dout, 0, no * 8));                    // columns beyond a chain's count stay zero
    }
    int rc = from_skeleton_launch(h, method, C, n_sk, dX, dV, dt_buf, dIA, nullptr, n_prefix, N, dt, discard_vt, dout, dcnt, nullptr);
    if (rc == PDMP_OK) TRY_OR_CLEAN(hipMemcpy(out, dout, no * 8, hipMemcpyDeviceToHost));
    if (rc == PDMP_OK && counts) TRY_OR_CLEAN(hipMemcpy(counts, dcnt, (size_t)C * 8, hipMemcpyDeviceToHost));
    cleanup();
    return rc;
}

int pdmp_sample_from_skeleton(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                              const double *t, const uint8_t *is_active, int64_t N, int32_t discard_vt, double *out) {
    return from_skeleton_host(h, 0, C, n_sk, X, V, t, is_active, 0, N, 0.0, discard_vt, out, nullptr);
}

int pdmp_sample_from_skeleton_dt(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                                 const double *t, const uint8_t *is_active, int64_t N, double dt, int32_t discard_vt, double *out,
                                 int64_t *counts) {
    return from_skeleton_host(h, 1, C, n_sk, X, V, t, is_active, 0, N, dt, discard_vt, out, counts);
}

int pdmp_sample_from_skeleton_prefix(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                                     const double *t, const uint8_t *is_active, int64_t n_prefix, double dt, int64_t M,
                                     int32_t discard_vt, double *out, int64_t *counts) {
    return from_skeleton_host(h, 2, C, n_sk, X, V, t, is_active, n_prefix, M, dt, discard_vt, out, counts);
}

int pdmp_close_at_time_device(pdmp_handle h, int64_t C, int64_t n_sk, double *X, double *V, double *t, const pdmp_diag_t *diag,
                              double T, int64_t *count, void *stream) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (h->cfg.sampler == PDMP_RHMC && (h->cfg.dim > 4 * WAVE || h->cfg.target_id > PDMP_T_BANANA))
        return fail(PDMP_ERR_UNSUPPORTED, "RHMC: the flow is integrated with up to 4 coordinates per lane (dim <= 256) on the built-in targets");
    if (!(T >= 0.0) || !std::isfinite(T)) return fail(PDMP_ERR_ARGUMENT, "T must be finite and non-negative. Current value: %g", T);   // sample.jl:335-337
    if (n_sk <= 0 || C < 0) return fail(PDMP_ERR_ARGUMENT, "n_sk must be positive and C non-negative");
    if (C == 0) return PDMP_OK;
    if (!X || !V || !t || !count) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    if (h->cfg.sampler == PDMP_STICKY && !(diag && diag->is_active))
        return fail(PDMP_ERR_ARGUMENT, "StickyZigZag: the closing point needs PDMPHistory.is_active (sample.jl:386-392)");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(256), grid((unsigned)((C + 3) / 4));
    pdmp_diag_t dg = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (diag) dg = *diag;
    if (h->cfg.sampler == PDMP_BOOMERANG)
        hipLaunchKernelGGL((h->cfg.rng_version == 1 ? close_at_kernel<2> : close_at_kernel<1>), grid, block, 0, s, h->cfg.dim, C, n_sk, X, V, t, T, count, 0, 0, 0.0, (const double *)nullptr, dg, (const int64_t *)nullptr, 2);
    else
        hipLaunchKernelGGL(close_at_kernel<0>, grid, block, 0, s, h->cfg.dim, C, n_sk, X, V, t, T, count,
                           h->cfg.sampler == PDMP_SUZZ ? 1 : (h->cfg.sampler == PDMP_RHMC ? 2 : (h->cfg.sampler == PDMP_STICKY ? 3 : 0)),
                           h->cfg.target_id, h->cfg.rh_step_size, (const double *)h->d_tparams, dg, (const int64_t *)nullptr, (h->cfg.rng_version == 1 ? 1 : 2));
    HIP_TRY(hipGetLastError());
    h->last_stream = s;
    return PDMP_OK;
}

int pdmp_compact_offsets_device(pdmp_handle h, int64_t C, const int64_t *count, int64_t *offsets, void *stream) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (C < 0) return fail(PDMP_ERR_ARGUMENT, "number of chains must be non-negative");
    if (!count || !offsets) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(compact_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, C, count, offsets);
    HIP_TRY(hipGetLastError());
    h->last_stream = (hipStream_t)stream;
    return PDMP_OK;
}

int pdmp_compact_history_device(pdmp_handle h, int64_t C, int64_t n_sk, const int64_t *count, const int64_t *offsets,
                                const double *X, const double *V, const double *t, const pdmp_diag_t *diag,
                                double *X_out, double *V_out, double *t_out, const pdmp_diag_t *diag_out, void *stream) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (n_sk <= 0 || C < 0) return fail(PDMP_ERR_ARGUMENT, "n_sk must be positive and C non-negative");
    if (C == 0) return PDMP_OK;
    if (!count || !offsets || !X || !V || !t || !X_out || !V_out || !t_out) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    HIP_TRY(hipSetDevice(h->device));
    const int64_t tiles = (n_sk + WAVE - 1) / WAVE, blocks = (C * tiles + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail(PDMP_ERR_UNSUPPORTED, "too many columns for one launch");
    pdmp_diag_t dg = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, og = dg;
    if (diag && diag_out) { dg = *diag; og = *diag_out; }
    hipStream_t s = (hipStream_t)stream;
    EventPair ev;
    if (int rc = begin_timed(h, s, ev)) return rc;
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, h->cfg.dim, C, n_sk, count, offsets, X, V, t, dg, X_out, V_out, t_out, og);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) { h->pool.push_back(ev); return fail(PDMP_ERR_RUNTIME, "compact_rows_kernel: %s", hipGetErrorString(e)); }
    return end_timed(h, s, ev);
}

int pdmp_sample_moments_device(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V, const double *t,
                               const uint8_t *is_active, const int64_t *lens, int64_t N, double *sum_x, double *sum_x2, void *stream) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    if (h->cfg.sampler == PDMP_RHMC && (h->cfg.dim > 4 * WAVE || h->cfg.target_id > PDMP_T_BANANA))
        return fail(PDMP_ERR_UNSUPPORTED, "RHMC: the flow is integrated with up to 4 coordinates per lane (dim <= 256) on the built-in targets");
    if (N <= 0) return fail(PDMP_ERR_ARGUMENT, "N must be positive. Current value: %lld", (long long)N);   // sample.jl:476-478
    if (n_sk <= 0) return fail(PDMP_ERR_ARGUMENT, "empty history");
    if (C < 0) return fail(PDMP_ERR_ARGUMENT, "number of chains must be non-negative");
    if (C == 0) return PDMP_OK;
    if (!X || !V || !t || !sum_x || !sum_x2) return fail(PDMP_ERR_ARGUMENT, "null buffer");
    if (h->cfg.sampler == PDMP_STICKY && !is_active)
        return fail(PDMP_ERR_ARGUMENT, "StickyZigZag: the samples need PDMPHistory.is_active (sample.jl:516-561)");
    if ((C + 3) / 4 > 0x7fffffffLL) return fail(PDMP_ERR_UNSUPPORTED, "too many chains for one launch");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    EventPair ev;
    if (int rc = begin_timed(h, s, ev)) return rc;
    const dim3 grid((unsigned)((C + 3) / 4)), block(256);
    if (h->cfg.sampler == PDMP_BOOMERANG)
        hipLaunchKernelGGL((h->cfg.rng_version == 1 ? moments_kernel<2> : moments_kernel<1>), grid, block, 0, s, h->cfg.dim, C, n_sk, X, V, t, N, sum_x, sum_x2, 0, 0, 0.0, (const double *)nullptr,
                           (const uint8_t *)nullptr, lens, 2);
    else
        hipLaunchKernelGGL(moments_kernel<0>, grid, block, 0, s, h->cfg.dim, C, n_sk, X, V, t, N, sum_x, sum_x2,
                           h->cfg.sampler == PDMP_SUZZ ? 1 : (h->cfg.sampler == PDMP_RHMC ? 2 : (h->cfg.sampler == PDMP_STICKY ? 3 : 0)),
                           h->cfg.target_id, h->cfg.rh_step_size, (const double *)h->d_tparams, is_active, lens, (h->cfg.rng_version == 1 ? 1 : 2));
    if (hipError_t e = hipGetLastError(); e != hipSuccess) { h->pool.push_back(ev); return fail(PDMP_ERR_RUNTIME, "moments_kernel: %s", hipGetErrorString(e)); }
    return end_timed(h, s, ev);
}

#ifdef PDMP_STAMP
int pdmp_debug_stamps(unsigned long long *out6) {       // diagnostic build only
    if (!g_stamps) return -1;
    (void)hipDeviceSynchronize();
    return hipMemcpy(out6, g_stamps, 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif

int pdmp_get_timing(pdmp_handle h, pdmp_timing_t *out) {
    if (!h || !out) return fail(PDMP_ERR_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = fold_timing(h)) return rc;
    *out = h->timing;
    return PDMP_OK;
}
int pdmp_reset_timing(pdmp_handle h) {
    if (!h) return fail(PDMP_ERR_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = fold_timing(h)) return rc;
    h->timing = {0.0, 0.0, 0};
    return PDMP_OK;
}

int pdmp_init_states_device(pdmp_handle h, int64_t C, int64_t chain_id_offset, uint64_t seed, double x_mean, double x_scale,
                            int32_t v_kind, double *xinit, double *vinit, void *stream) {
    if (!h || !xinit || !vinit) return fail(PDMP_ERR_ARGUMENT, "init_states: null argument");
    if (C <= 0 || chain_id_offset < 0) return fail(PDMP_ERR_ARGUMENT, "init_states: C must be positive and chain_id_offset non-negative");
    if (v_kind < PDMP_V_RADEMACHER || v_kind > PDMP_V_SPHERE) return fail(PDMP_ERR_ARGUMENT, "init_states: unknown v_kind %d", (int)v_kind);
    HIP_TRY(hipSetDevice(h->device));
    const int threads = 64;
    hipLaunchKernelGGL((h->cfg.rng_version == 1 ? init_states_kernel<1> : init_states_kernel<2>), dim3((unsigned)((C + threads - 1) / threads)), dim3(threads), 0, (hipStream_t)stream,
                       C, chain_id_offset, seed, h->cfg.dim, x_mean, x_scale, (int)v_kind, xinit, vinit);
    HIP_TRY(hipGetLastError());
    return PDMP_OK;
}

int pdmp_debug_rng(pdmp_handle h, uint64_t seed, int64_t chain, int32_t kind, int64_t n, double *out) {
    if (!h || !out || n <= 0 || kind < 0 || kind > 2) return fail(PDMP_ERR_ARGUMENT, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    double *d = nullptr;
    HIP_TRY(hipMalloc(&d, (size_t)n * 8));
    hipLaunchKernelGGL((h->cfg.rng_version == 1 ? debug_rng_kernel<1> : debug_rng_kernel<2>), dim3(1), dim3(WAVE), 0, nullptr, seed, chain, (int)kind, n, d);
    hipError_t e = hipMemcpy(out, d, (size_t)n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PDMP_ERR_RUNTIME, "debug_rng: %s", hipGetErrorString(e));
    return PDMP_OK;
}

}  // extern "C"
