// dense_batched.hip -- Zig-Zag on a dense-precision Gaussian, U = x'Px/2 (BASELINE config 4), for MANY chains:
// the only O(d^2) operation of the path, grad-U(x_t) = P x_t, is taken out of the per-chain state machine and
// done for all chains at once as ONE fp64 GEMM per phase on the matrix cores (rocBLAS dgemm -> v_mfma_f64 on
// gfx950), instead of C independent mat-vecs that each stream the 8 MB of P out of L2 / Infinity Cache
// (group_kernel.hpp, GTarget<T_GAUSS_DENSE>: 7e4 events/s at d = 1000).
//
// The thinning state machine of src/SamplingLoopInplace.jl:27-217 becomes a coroutine per chain that yields
// every time it needs a gradient: each phase every live chain posts exactly one request vector (x + v t for
// the next grid time of its bound, UpperBound.jl:203-247 built lazily cell by cell, or for the proposed event
// time, SamplingLoopInplace.jl:113-129); the phase's GEMM multiplies P with [requests | velocities] (the
// second half gives P v, the time-derivative of grad-U along the flow, for the forward-mode slopes); the step
// kernel consumes the products, advances every chain to its next request and records accepted events.
// Chains advance at their own pace (no lock-step waste: a chain builds only the cells it reads).
//
// Arithmetic: everything except the GEMM is the engine's WAVE64 order (wave_math.hpp).  The GEMM's accumulation
// order is rocBLAS' (MFMA k-blocks), so this path is checked against the oracle to a tolerance (1e-9 relative on
// event times / positions over short runs), not bit for bit; PDMP_DENSE_MODE=matvec selects the bit-exact kernel.
#include "skeleton_kernel.hpp"

#include <rocblas/rocblas.h>
extern "C" int pdmp_dgemm_mfma(int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc, void *stream);   // dgemm_mfma.hip
extern "C" int pdmp_dgemm_mfma_polite(int max_wg, int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc, void *stream);

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <mutex>

PDMP_NS_BEGIN

enum : int { DB_LEFT = 0, DB_CELL = 1, DB_RATE = 2 };
enum : int { DB_FRESH = 0, DB_ERR = 1, DB_REUSE = 2 };

struct DbState {                 // per-chain coroutine state (one 256-byte record per chain)
    double t, horizon, ts, tp, E, lb, ar, hB, harg, h, acc, cprev, bj, ev[5];
    unsigned long long rng_n;
    int32_t phase, mode, j, k, it, errored, rejected, hits, done, pad;
};

struct DbParams {
    int32_t d, signed_bound, adaptive;
    int64_t C, n_sk, chain0, cap;
    uint64_t seed;
    double tmax;
    const double *xinit, *vinit;
    double *Xs;                  // C x d chain-major positions
    double *In;                  // d x 2C column-major: [requests | velocities]   (GEMM input)
    double *Out;                 // d x 2C column-major: [P requests | P velocities] (GEMM output)
    double *Gprev;               // C x d: gradient at the left grid time of the cell under construction
    const double *P;             // d x d column-major precision matrix (for the rank-one update of P v at a flip)
    DbState *st;
    double *X, *V, *t, *H, *AR, *EV;
    int32_t *EB, *RJ, *HH;
    int32_t *status;             // [0] itercap, [1] guard
    int32_t *remaining;          // live chains
};

__device__ __forceinline__ void db_record(const DbParams &p, int64_t c, int lane, const double *x, const double *v,
                                          const DbState &s) {
    const size_t e = (size_t)c * (size_t)p.n_sk + (size_t)s.k;
    double *Xe = p.X + e * (size_t)p.d, *Ve = p.V + e * (size_t)p.d;
    for (int i = lane; i < p.d; i += WAVE) { Xe[i] = x[i]; Ve[i] = v[i]; }
    if (lane == 0) {
        p.t[e] = s.t;
        if (p.H) p.H[e] = s.horizon;
        if (p.AR) p.AR[e] = s.ar;
        if (p.EB) p.EB[e] = s.errored;
        if (p.RJ) p.RJ[e] = s.rejected;
        if (p.HH) p.HH[e] = s.hits;
        if (p.EV) for (int q = 0; q < 5; ++q) p.EV[5 * e + q] = s.ev[q];
    }
}

// request = x + v * t  (ZigZagSamplers.jl:80), fused as in the group kernel
__device__ __forceinline__ void db_request(const DbParams &p, int lane, const double *x, const double *v, double t, double *req) {
    for (int i = lane; i < p.d; i += WAVE) req[i] = fma(v[i], t, x[i]);
}
__device__ __forceinline__ void db_flow(const DbParams &p, int lane, double *x, const double *v, double t) {
    for (int i = lane; i < p.d; i += WAVE) x[i] = fma(v[i], t, x[i]);
}

// init_state (AbstractPDMP.jl:93-153) + record of the initial state (sample.jl:271-272) + first request (t = 0)
__global__ __launch_bounds__(256) void db_init_kernel(const DbParams p) {
    const int lane = lane_id();
    const int64_t c = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (c >= p.C) return;
    double *x = p.Xs + (size_t)c * p.d, *req = p.In + (size_t)c * p.d, *v = p.In + ((size_t)p.C + (size_t)c) * p.d;
    for (int i = lane; i < p.d; i += WAVE) {
        const double xi = p.xinit[(size_t)c * p.d + i], vi = p.vinit[(size_t)c * p.d + i];
        x[i] = xi; v[i] = vi; req[i] = xi;
    }
    DbState s;
    s.t = 0.0; s.horizon = p.tmax; s.ts = 0.0; s.tp = 0.0; s.E = 0.0; s.lb = 0.0; s.ar = 0.0; s.hB = p.tmax;
    s.harg = 0.0; s.h = 0.0; s.acc = 0.0; s.cprev = 0.0; s.bj = 0.0;
    for (int q = 0; q < 5; ++q) s.ev[q] = 0.0;
    s.rng_n = 0; s.phase = DB_LEFT; s.mode = DB_FRESH; s.j = 0; s.k = 0; s.it = 0;
    s.errored = 0; s.rejected = 0; s.hits = 0; s.done = (p.n_sk <= 1) ? 1 : 0; s.pad = 0;
    db_record(p, c, lane, x, v, s);
    s.k = 1;
    if (lane == 0) {
        p.st[c] = s;
        if (!s.done) atomicAdd(p.remaining, 1);
    }
}

// Start a bound over [0, harg]: one_step_of_thinning! (SamplingLoopInplace.jl:65-85), erroneous_acceptance_rate!
// (:131-151) or the rebuild after a rejection (if_reject!, :188-203).  The exponential is drawn first (the bound
// consumes no random numbers).  Posts the request for grid time t_1.
__device__ __forceinline__ void db_begin_bound(const DbParams &p, int lane, DbState &s, ChainRng &rng, const double *x,
                                               const double *v, double *req) {
    const double harg = (s.mode == DB_ERR) ? s.horizon / 2 : ((s.mode == DB_REUSE) ? s.hB : s.horizon);
    s.hB = harg; s.harg = harg; s.h = harg / 9.0;
    const double e1 = rng.exponential();
    s.E = (s.mode == DB_REUSE) ? s.E + e1 : e1;
    s.acc = 0.0; s.cprev = 0.0; s.bj = 0.0; s.j = 0;
    db_request(p, lane, x, v, s.h, req);                         // t_1 = 1 * h
    s.phase = DB_CELL;
}

__global__ __launch_bounds__(256) void db_step_kernel(const DbParams p) {
    const int lane = lane_id();
    const int64_t c = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (c >= p.C) return;
    DbState s = p.st[c];
    if (s.done) return;
    const int d = p.d;
    double *x = p.Xs + (size_t)c * d, *req = p.In + (size_t)c * d, *v = p.In + ((size_t)p.C + (size_t)c) * d;
    const double *G = p.Out + (size_t)c * d;                     // P * (last request)
    const double *B = p.Out + ((size_t)p.C + (size_t)c) * d;     // P * v
    double *Gp = p.Gprev + (size_t)c * d;
    ChainRng rng;
    rng.init(p.seed, (uint64_t)(p.chain0 + c));
    rng.restore(s.rng_n);
    const bool adaptive = p.adaptive != 0, sgn = p.signed_bound != 0;

    if (++s.it > p.cap * 4) {                                    // phases, not thinning steps: generous cap
        s.done = 1;
        if (lane == 0) { atomicAdd(&p.status[0], 1); atomicSub(p.remaining, 1); p.st[c] = s; }
        return;
    }

    if (s.phase == DB_LEFT) {
        // G = grad-U at the current state (grid time 0 of a new bound)
        for (int i = lane; i < d; i += WAVE) Gp[i] = G[i];
        db_begin_bound(p, lane, s, rng, x, v, req);
    } else if (s.phase == DB_CELL) {
        // cell j from grid times j (Gp) and j+1 (G); slopes from B = P v: upper_bound_grid_vect, UpperBound.jl:208-246
        const double h = s.h, t0 = (double)s.j * h, t1 = (s.j == 8) ? s.harg : (double)(s.j + 1) * h;
        double part = 0.0;
        for (int i = lane; i < d; i += WAVE) {
            const double vi = v[i], bi = B[i];
            double v0 = Gp[i] * vi, v1 = G[i] * vi, g = bi * vi;          // signed_rate_vect and its t-derivative
            double g0 = g, g1 = g;
            if (!sgn) { g0 = (0.0 < v0) ? g : 0.0; g1 = (0.0 < v1) ? g : 0.0; v0 = fmax(0.0, v0); v1 = fmax(0.0, v1); }
            const double num = fma(-g0, t0, fma(g1, t1, v0 - v1));         // :229
            const double den = g1 - g0;
            const double q = (den == 0.0) ? ((num > 0.0) ? h : 0.0) : fmin(fmax(num / den, 0.0), h);   // :234-235
            const double inter = fma(g0, q, v0);                           // :237
            part += fmax(fmax(fmax(v0, v1), inter), 0.0);                  // :239-241
        }
        const double bj = wave_sum(part);                                  // :246
        s.bj = bj;
        s.acc += bj;
        const double cj = s.acc * h;
        bool fin = false;
        if (!(cj < s.E)) {                                                 // next_event, :264-273
            s.tp = t0 + (s.E - s.cprev) / (cj - s.cprev) * h;
            s.lb = bj;
            fin = true;
        } else {
            s.cprev = cj;
            if (s.j == 8) { s.tp = __builtin_huge_val(); s.lb = bj; fin = true; }
        }
        if (!fin) {
            for (int i = lane; i < d; i += WAVE) Gp[i] = G[i];             // right grid time becomes the left one
            s.j++;
            db_request(p, lane, x, v, (s.j == 8) ? s.harg : (double)(s.j + 1) * h, req);
        } else {
            bool propose;
            if (s.mode == DB_ERR) {                                        // :131-151
                if (adaptive) s.horizon = s.hB;
                s.errored++;
                { const int slot = s.errored % 5;                          // :148
#pragma unroll
                  for (int q = 0; q < 5; ++q) if (q == slot) s.ev[q] = s.ar; }
                propose = s.tp < s.horizon;
            } else if (s.mode == DB_REUSE) {                               // :188-203
                if (adaptive) s.horizon = s.horizon / 1.04;
                s.rejected++;
                if (s.tp > s.horizon) { db_flow(p, lane, x, v, s.horizon); s.ts += s.horizon; s.hits++; }
                propose = s.tp < s.horizon;
            } else if (s.tp > s.horizon) {                                 // move_to_horizon!, :87-101
                db_flow(p, lane, x, v, s.horizon);
                s.ts += s.horizon;
                s.hits++;
                if (adaptive) s.horizon = s.horizon * 1.01;
                propose = false;
            } else {
                propose = s.tp < s.horizon;
            }
            s.mode = DB_FRESH;
            if (propose) { db_request(p, lane, x, v, s.tp, req); s.phase = DB_RATE; }
            else { db_request(p, lane, x, v, 0.0, req); s.phase = DB_LEFT; }    // new bound from the (possibly moved) state
        }
    } else {  // DB_RATE: G = grad-U at x + v tp
        double part = 0.0;
        for (int i = lane; i < d; i += WAVE) part += fmax(0.0, G[i] * v[i]);      // ZigZagSamplers.jl:83-86
        const double lam = wave_sum(part);
        s.ar = lam / s.lb;
        if (s.ar > 1.0) {
            s.mode = DB_ERR;
            db_request(p, lane, x, v, 0.0, req);
            s.phase = DB_LEFT;
        } else {
            const double u = rng.uniform();
            if (u < s.ar) {                                                // if_accept!, :170-186
                db_flow(p, lane, x, v, s.tp);
                // velocity_jump (ZigZagSamplers.jl:101-107): Categorical over lambda_i / S, un-normalised walk against u2*S
                const double u2 = rng.uniform();
                if (lam > 0.0) {
                    const double theta = u2 * lam;
                    double carry = 0.0;
                    int m = -1;
                    for (int base = 0; base < d && m < 0; base += WAVE) {
                        const int i = base + lane;
                        const double li = (i < d) ? fmax(0.0, G[i] * v[i]) : 0.0;
                        const double inc = wave_scan_incl(li, lane);
                        const unsigned long long hit = __ballot((i < d) && (carry + inc > theta));
                        if (hit != 0ull) m = base + (int)__builtin_ctzll(hit);
                        carry = carry + readlane_d(inc, WAVE - 1);
                    }
                    if (m < 0) m = d - 1;
                    // v[m] *= -1, and P v follows by a rank-one update, P (v - 2 v_m e_m) = P v - 2 v_m P[:, m], instead of
                    // being recomputed by the next GEMM (the host loop re-syncs it with a full product every 32 phases)
                    const double vm_old = v[m];
                    {
                        double *Bv = p.Out + ((size_t)p.C + (size_t)c) * (size_t)d;
                        const double *Pm = p.P + (size_t)m * (size_t)d;
                        const double f = 2.0 * vm_old;
                        for (int i = lane; i < d; i += WAVE) Bv[i] = Bv[i] - f * Pm[i];
                    }
                    if (lane == (m & (WAVE - 1))) v[m] = -v[m];
                } else if (lane == 0) {
                    atomicAdd(&p.status[1], 1);                            // Q11 guard
                }
                s.t = s.t + s.tp + s.ts;
                s.ts = 0.0;
                s.tp = 0.0;
                db_record(p, c, lane, x, v, s);
                s.errored = 0; s.rejected = 0; s.hits = 0;
                for (int q = 0; q < 5; ++q) s.ev[q] = 0.0;
                s.it = 0;
                if (++s.k >= p.n_sk) {
                    s.done = 1;
                    if (lane == 0) atomicSub(p.remaining, 1);
                } else {
                    // grad-U at the new state is the product just consumed: it is the left grid time of the next bound
                    for (int i = lane; i < d; i += WAVE) Gp[i] = G[i];
                    db_begin_bound(p, lane, s, rng, x, v, req);            // P v_new arrives with the next GEMM
                }
            } else {
                s.mode = DB_REUSE;                                         // if_reject!: rebuild from the unchanged state
                db_request(p, lane, x, v, 0.0, req);
                s.phase = DB_LEFT;
            }
        }
    }
    s.rng_n = rng.n;
    if (lane == 0) p.st[c] = s;
}


// ---------------------------------------------------------------- incremental form (the default for Zig-Zag on a dense precision)
// grad-U is linear and the Zig-Zag flow is x + v t, so along a chain's trajectory
//     P (x + v t) = P x + t P v,        a flow by tau:  P x <- P x + tau P v,        a flip of v_m:  P v <- P v - 2 v_m P[:, m],
// i.e. once gx = P x and gv = P v are known, every gradient the thinning loop asks for -- nine grid times per bound, the
// proposal -- is an O(d) fma, and an event costs one column of P instead of ~12 products with all of P.  One wave per chain,
// x, v, gx, gv in registers (coordinate i = s*64 + lane), control flow wave-uniform, the loop of src/SamplingLoopInplace.jl
// written straight down (same state machine, random-stream order and cell arithmetic as db_step_kernel above).  The recurrences
// only accumulate rounding (one fma per flow, one per flip), and they are re-based every `resync` events on exact products:
// the host runs the chains in segments and recomputes [P x | P v] of ALL chains with one fp64 GEMM (matrix cores) between
// segments.  Checked against the oracle to the same tolerance as the GEMM-per-phase path.
struct DiState { double t, horizon; unsigned long long rng_n; int32_t dead, pad; };

struct DiParams {
    int32_t d, signed_bound, adaptive;
    int64_t C, n_sk, chain0, cap, k_begin, k_end;   // this launch records skeleton columns k_begin .. k_end-1
    uint64_t seed;
    double tmax;
    double *In;                  // d x 2C column-major: [x | v] of every chain (state between segments, GEMM input)
    double *Out;                 // d x 2C: [P x | P v] (GEMM output; the BPS / Boomerang kernels carry P x across events in its first half)
    const double *P;
    DiState *st;
    double *X, *V, *t, *H, *AR, *EV;
    int32_t *EB, *RJ, *HH;
    int32_t *status;             // [0] itercap, [1] guard
    double refresh;              // BPS (dr_bps_kernel): refresh_rate
    int32_t gaussian_velocity, pad;
    int64_t c_lo, c_hi;          // the output slots this launch advances: [c_lo, c_hi) (two half-ensembles on two streams); slot j = chain j of the workspace unless chain_map
    // time horizon (sample_skeleton(sampler, T, ...), sample.jl:323-439; pdmp_sample_skeleton_until_device): a chain stops after its first event
    // with t >= t_stop; resume: the workspace holds the state the previous chunk left (column 0 = its last event again); chain_map: slot -> chain
    // of a resumed chunk (the chains still running, compacted); count[slot] = columns recorded in this chunk
    double t_stop;
    const int64_t *chain_map;
    int64_t *count;
    int32_t resume, pad2;
};

// SGN: signed_bound (the constructor default).  There the slope of every coordinate's rate is the same double at both ends of a
// cell (gv_i v_i), so den == 0 exactly and the tangent intersection needs no division -- as in the group kernel's kLinear path.
#ifdef PDMP_DENSE_PLAIN_STORES     // A/B builds
#define PDMP_ROW_STORE(ptr, val) (*(ptr) = (val))
#else
#define PDMP_ROW_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif
// (round 5, same arrays, bench.py --ab-env: plain stores 7.80 ms against 7.98 with a run-time switch compiled in -- and 7.40 without the switch:
//  the launch is bound by these stores, scripts/probes/store_probe.hip, and the kernel's schedule around them is touchy)
// ---- Totals of sixteen per-lane values over the wave, all at once ("reduce-scatter"): instead of sixteen butterflies of six stages
// each (16 x 22 instructions, and -- walked slot by slot until the hit -- sixteen dependent chains through the LDS crossbar), every
// stage HALVES the number of values a lane carries while it doubles the lanes each value has been summed over:
//   xor 32 (v_permlane32_swap of slots s | s + 8), xor 16 (v_permlane16_swap of s | s + 4), xor 8 (row_ror:8 of s | s + 2),
//   then the quads are made uniform (xor 1, xor 2) and the last exchange (s | s + 1) goes through row_half_mirror.
// 63 instructions, log depth.  Lane L ends up with the total of slot  ((L >> 2) & 1) + 2 ((L >> 3) & 1) + 4 (L >> 4).
// The order of each slot's sum differs from wave_sum's (this path is held to a tolerance, not bit for bit: header of this file).
constexpr int DPP_ROW_ROR8 = 0x128, DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
__device__ __forceinline__ void swap32_pair(double x, double y, double &lo_half, double &hi_half) {   // -> [x.lo | y.lo], [x.hi | y.hi]
    const auto rl = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    lo_half = __hiloint2double(rh[0], rl[0]); hi_half = __hiloint2double(rh[1], rl[1]);
}
__device__ __forceinline__ void swap16_pair(double x, double y, double &even, double &odd) {           // -> [x.r0, y.r0, x.r2, y.r2], [x.r1, y.r1, x.r3, y.r3]
    const auto rl = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    even = __hiloint2double(rh[0], rl[0]); odd = __hiloint2double(rh[1], rl[1]);
}
__device__ __forceinline__ double slot_totals16(const double (&val)[16], int lane) {
    double z[8], w[4], u[2];
#pragma unroll
    for (int s = 0; s < 8; ++s) { double a, b; swap32_pair(val[s], val[s + 8], a, b); z[s] = a + b; }      // lanes < 32: slot s, lanes >= 32: slot s + 8
#pragma unroll
    for (int s = 0; s < 4; ++s) { double a, b; swap16_pair(z[s], z[s + 4], a, b); w[s] = a + b; }          // row r: slot s + 4 r
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {                                                                           // half h of a row: slot s + 2 h + 4 r
        const double keep = b3 ? w[s + 2] : w[s], send = b3 ? w[s] : w[s + 2];
        u[s] = keep + dpp_d<DPP_ROW_ROR8>(send);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) { u[s] = u[s] + dpp_d<DPP_QUAD_XOR1>(u[s]); u[s] = u[s] + dpp_d<DPP_QUAD_XOR2>(u[s]); }   // quads uniform
    const double keep = b2 ? u[1] : u[0], send = b2 ? u[0] : u[1];
    return keep + dpp_d<DPP_ROW_HALF_MIRROR>(send);                                                          // quad q of a half: slot q + 2 h + 4 r
}
__device__ __forceinline__ int slot_lane16(int s) { return ((s & 1) << 2) | (((s >> 1) & 1) << 3) | ((s >> 2) << 4); }   // a lane that holds slot s

// UNTIL: the time-horizon features (t_stop, resume, chain_map, count) -- separate instantiations, so that the fixed-count kernels keep the
// registers and instructions they had (config 4: 7.44 ms; with the features compiled in 7.85)
template <int KC, bool SGN, bool UNTIL = false>
__global__ __launch_bounds__(256, SGN ? 2 : 1) void di_segment_kernel(const DiParams p) {
    const int lane = lane_id();
    const int64_t j = p.c_lo + (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // slot in this launch's outputs
    if (j >= p.c_hi) return;
    const int64_t c = (UNTIL && p.chain_map) ? p.chain_map[j] : j;      // the chain: workspace columns, state record, random stream
    DiState cs;
    if (p.k_begin == 0 && !(UNTIL && p.resume)) { cs.t = 0.0; cs.horizon = p.tmax; cs.rng_n = 0; cs.dead = 0; cs.pad = 0; }
    else cs = p.st[c];
    if (cs.dead || (UNTIL && cs.pad && p.k_begin != 0)) return;      // (pad: the chain has passed t_stop -- in this chunk; a resumed chunk still records its column 0)
    const int d = p.d;
    const bool adaptive = p.adaptive != 0;
    constexpr bool sgn = SGN;
    // v, gx = P x, gv = P v in registers; x is only touched by flows and records: it lives in LDS (KC*512 B per wave).
    // (Carrying the signed products (P x) .* v, (P v) .* v instead -- 4 instead of 7 instructions per coordinate and cell, the right end of
    // a cell reused as the left end of the next -- was built twice and measured slower both times: round 3 8.31 -> 8.70 ms, round 4 (after
    // the wave sums had lost their LDS stages) 7.54 -> 8.03 ms: sixteen more live registers cost more than the instructions save.)
    __shared__ double x_lds[4][KC * WAVE];
    double *x = x_lds[threadIdx.x >> 6] + lane;          // slot s at x[s * WAVE]
    double v[KC], gx[KC], gv[KC];
    {
        const double *xin = p.In + (size_t)c * d, *vin = p.In + ((size_t)p.C + (size_t)c) * d;
        const double *gxin = p.Out + (size_t)c * d, *gvin = p.Out + ((size_t)p.C + (size_t)c) * d;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const int i = s * WAVE + lane;
            const bool ok = i < d;
            x[s * WAVE] = ok ? xin[i] : 0.0; v[s] = ok ? vin[i] : 0.0; gx[s] = ok ? gxin[i] : 0.0; gv[s] = ok ? gvin[i] : 0.0;
        }
    }
    ChainRng rng;
    rng.init(p.seed, (uint64_t)(p.chain0 + c));
    rng.restore(cs.rng_n);
    double t = cs.t, horizon = cs.horizon, ts = 0.0, ar = 0.0, E = 0.0, hB = horizon;
    double ev[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int errored = 0, rejected = 0, hits = 0, mode = DB_FRESH;
    int64_t k = p.k_begin, it = 0;
    bool dead = false;

    auto record = [&](int64_t kk) {
        const size_t e = (size_t)j * (size_t)p.n_sk + (size_t)kk;      // (j: the chain's slot in this launch's outputs)
        double *Xe = p.X + e * (size_t)d, *Ve = p.V + e * (size_t)d;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const int i = s * WAVE + lane;
            // whole 64-byte lines (a row is 8 d bytes, d = 1000: 125 lines), written once and not read again by this launch: streaming
            // stores keep them from displacing the columns of P in the L2
            if (i < d && !(p.pad & 1)) { PDMP_ROW_STORE(&Xe[i], x[s * WAVE]); PDMP_ROW_STORE(&Ve[i], v[s]); }      // (p.pad: PDMP_DENSE_DEBUG, measurements only)
        }
        if (lane == 0) {
            p.t[e] = t;
            if (p.H) p.H[e] = horizon;
            if (p.AR) p.AR[e] = ar;
            if (p.EB) p.EB[e] = errored;
            if (p.RJ) p.RJ[e] = rejected;
            if (p.HH) p.HH[e] = hits;
            if (p.EV) for (int q = 0; q < 5; ++q) p.EV[5 * e + q] = ev[q];
        }
    };
    auto flow = [&](double tau) {              // x <- x + v tau (ZigZagSamplers.jl:80) and P x with it
#pragma unroll
        for (int s = 0; s < KC; ++s) { x[s * WAVE] = fma(v[s], tau, x[s * WAVE]); gx[s] = fma(tau, gv[s], gx[s]); }
    };

    if (k == 0) { record(0); k = 1; }          // init_state + record of the initial state (sample.jl:271-272)
    while (k < p.k_end && (!UNTIL || t < p.t_stop)) {              // (`while state.t < T`, sample.jl:372; t_stop = +inf for the fixed-count path)
        if (++it > p.cap * 4) { dead = true; break; }
        // ---- bound over [0, harg]: one_step_of_thinning! (:65-85) / erroneous_acceptance_rate! (:131-151) / if_reject! (:188-203)
        const double harg = (mode == DB_ERR) ? horizon / 2 : ((mode == DB_REUSE) ? hB : horizon);
        hB = harg;
        const double h = harg / 9.0;
        const double e1 = rng.exponential();
        E = (mode == DB_REUSE) ? E + e1 : e1;
        double acc = 0.0, cprev = 0.0, tp = __builtin_huge_val(), lb = 0.0;
        for (int j = 0; j < 9; ++j) {          // cells built lazily in grid order (upper_bound_grid_vect, UpperBound.jl:208-246)
            const double t0 = (double)j * h, t1 = (j == 8) ? harg : (double)(j + 1) * h;
            double part = 0.0;
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                const double vi = v[s];
                double v0 = fma(t0, gv[s], gx[s]) * vi, v1 = fma(t1, gv[s], gx[s]) * vi;   // signed_rate_vect at the two grid times
                if constexpr (sgn) {
                    // "affine cell" (DESIGN.md 2): under the signed bound every coordinate's rate is linear in t with the same slope double
                    // gv_i v_i at both ends, the reference's tangent intersection divides by zero and clamps to h or 0, `inter` is v0 or
                    // v0 + g h = v1 up to rounding: the cell maximum is max(v0, v1, 0) (the slope may have either sign here)
                    part += fmax(fmax(v0, v1), 0.0);
                    continue;
                }
                const double g = gv[s] * vi;
                double g0 = g, g1 = g;
                if (!sgn) { g0 = (0.0 < v0) ? g : 0.0; g1 = (0.0 < v1) ? g : 0.0; v0 = fmax(0.0, v0); v1 = fmax(0.0, v1); }
                const double num = fma(-g0, t0, fma(g1, t1, v0 - v1));         // :229
                const double den = g1 - g0;
                const double q = (sgn || den == 0.0) ? ((num > 0.0) ? h : 0.0) : fmin(fmax(num / den, 0.0), h);   // :234-235
                const double inter = fma(g0, q, v0);                           // :237
                part += fmax(fmax(fmax(v0, v1), inter), 0.0);                  // :239-241
            }
            const double bj = wave_sum(part);                                  // :246
            acc += bj;
            const double cj = acc * h;
            lb = bj;
            if (!(cj < E)) { tp = t0 + (E - cprev) / (cj - cprev) * h; break; }   // next_event, :264-273
            cprev = cj;
        }
        bool propose;
        if (mode == DB_ERR) {                                                  // :131-151
            if (adaptive) horizon = hB;
            errored++;
            { const int slot = errored % 5;
#pragma unroll
              for (int q = 0; q < 5; ++q) if (q == slot) ev[q] = ar; }
            propose = tp < horizon;
        } else if (mode == DB_REUSE) {                                         // :188-203
            if (adaptive) horizon = horizon / 1.04;
            rejected++;
            if (tp > horizon) { flow(horizon); ts += horizon; hits++; }
            propose = tp < horizon;
        } else if (tp > horizon) {                                             // move_to_horizon!, :87-101
            flow(horizon);
            ts += horizon;
            hits++;
            if (adaptive) horizon = horizon * 1.01;
            propose = false;
        } else {
            propose = tp < horizon;
        }
        mode = DB_FRESH;
        if (!propose) continue;
        // ---- proposal at tp (ac_step!, :113-129): lambda = sum max(0, grad-U(x + v tp) .* v), ZigZagSamplers.jl:83-86
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < KC; ++s) part += fmax(0.0, fma(tp, gv[s], gx[s]) * v[s]);
        const double lam = wave_sum(part);
        ar = lam / lb;
        if (ar > 1.0) { mode = DB_ERR; continue; }
        const double u = rng.uniform();
        if (!(u < ar)) { mode = DB_REUSE; continue; }
        // ---- accepted (if_accept!, :170-186)
        flow(tp);                                                              // gx becomes exactly the gradient the rate used
        const double u2 = rng.uniform();
        if (lam > 0.0) {
            // velocity_jump (ZigZagSamplers.jl:101-107): Categorical over lambda_i / S as an un-normalised walk against u2*S in
            // coordinate order; level 1 = slot totals (64 coordinates each), level 2 = one scan inside the slot that was hit
            const double theta = u2 * lam;
            int sh = -1;
            double carry = 0.0, carry_at = 0.0;
            if constexpr (KC == 16) {
                // all sixteen slot totals at once (slot_totals16), gathered into lanes 0..15 in slot order, one 16-lane scan, one ballot
                double lv[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) lv[s] = fmax(0.0, gx[s] * v[s]);
                const double tot = slot_totals16(lv, lane);
                const double Ts = __shfl(tot, slot_lane16(lane & 15), WAVE);       // lane s (and s + 16, ...) <- the total of slot s
                double pre = Ts;                                                    // inclusive prefix over the slots, inside each row of 16 lanes
                { double q;
                  q = dpp_d<DPP_ROW_SHR1>(pre); if ((lane & 15) >= 1) pre = pre + q;
                  q = dpp_d<DPP_ROW_SHR2>(pre); if ((lane & 15) >= 2) pre = pre + q;
                  q = dpp_d<DPP_ROW_SHR4>(pre); if ((lane & 15) >= 4) pre = pre + q;
                  q = dpp_d<DPP_ROW_SHR8>(pre); if ((lane & 15) >= 8) pre = pre + q; }
                const unsigned hit16 = (unsigned)(__ballot(pre > theta) & 0xffffull);
                if (hit16 != 0u) {
                    sh = (int)__builtin_ctz(hit16);
                    carry_at = (sh == 0) ? 0.0 : readlane_d(pre, sh - 1);
                }
            } else {
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                if (sh < 0) {                    // wave-uniform: slots after the hit one are not summed
                    const double Ts = wave_sum(fmax(0.0, gx[s] * v[s]));
                    if (carry + Ts > theta) { sh = s; carry_at = carry; }
                    carry = carry + Ts;
                }
            }
            }
            int m = -1;
            if (sh >= 0) {
                double lsel = 0.0;
#pragma unroll
                for (int s = 0; s < KC; ++s) if (s == sh) lsel = fmax(0.0, gx[s] * v[s]);
                const double inc = wave_scan_incl(lsel, lane);
                const unsigned long long hit = __ballot((sh * WAVE + lane < d) && (carry_at + inc > theta));
                if (hit != 0ull) m = sh * WAVE + (int)__builtin_ctzll(hit);
            }
            if (m < 0) m = d - 1;
            m = __builtin_amdgcn_readfirstlane(m);
            // v[m] *= -1 and P v follows by the rank-one term
            const int ms = m >> 6, ml = m & (WAVE - 1);
            double vsel = 0.0;
#pragma unroll
            for (int s = 0; s < KC; ++s) if (s == ms) vsel = v[s];
            const double f = 2.0 * readlane_d(vsel, ml);
            const double *__restrict__ Pm = p.P + (size_t)m * (size_t)d;
            double pc[KC];                      // column m of P: issued here, consumed after the record's stores (latency overlapped)
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                const int i = s * WAVE + lane;
                pc[s] = (i < d && !(p.pad & 2)) ? Pm[i] : 0.0;
            }
#pragma unroll
            for (int s = 0; s < KC; ++s) if (s == ms && lane == ml) v[s] = -v[s];
            t = t + tp + ts;
            ts = 0.0;
            record(k);
#pragma unroll
            for (int s = 0; s < KC; ++s) gv[s] = gv[s] - f * pc[s];
        } else {
            if (lane == 0) atomicAdd(&p.status[1], 1);                         // Q11 guard
            t = t + tp + ts;
            ts = 0.0;
            record(k);
        }
        errored = 0; rejected = 0; hits = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) ev[q] = 0.0;
        it = 0;
        ++k;
    }
    // state for the next segment: x, v back into the GEMM's input, the scalars into st
    {
        double *xo = p.In + (size_t)c * d, *vo = p.In + ((size_t)p.C + (size_t)c) * d;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const int i = s * WAVE + lane;
            if (i < d) { xo[i] = x[s * WAVE]; vo[i] = v[s]; }
        }
    }
    if (lane == 0) {
        cs.t = t; cs.horizon = horizon; cs.rng_n = rng.n; cs.dead = dead ? 1 : 0;
        cs.pad = (UNTIL && !(t < p.t_stop)) ? 1 : 0;
        if (UNTIL && p.count && (cs.pad || k >= p.n_sk)) p.count[j] = k;      // columns recorded in this chunk (the group kernel's rule: engine_api.hip, UntilArgs)
        p.st[c] = cs;
        if (dead) atomicAdd(&p.status[0], 1);
    }
}


// ---------------------------------------------------------------- BPS on a dense precision: one GEMM per EVENT (round 3)
// BouncyParticleSamplers.jl:39-74 with grad-U = P x.  Between two events of a chain nothing needs a product with P: along the flow
//     <grad-U(x + v t), v> = <P x, v> + t <P v, v> = A + B t,
// so with gx = P x and gv = P v at hand the whole thinning iteration -- the grid bound of the scalar bounded function
// (upper_bound_grid, UpperBound.jl:92-137: values A + B t_j (+ refresh_rate), slope B, tangent intersections, cumulative sum,
// inversion), horizon moves, rejections, the proposal's rate -- is two dot products and scalar arithmetic, wave-uniform.  Only an
// accepted event changes v other than by a flow (reflection v - 2 <g, v> / <g, g> g with g = P x at the event, or a refresh by
// normals), and P v' is a product with all of P: the chain records the event, writes x and v' back and YIELDS; the host multiplies
// P [x | v] of ALL chains with one fp64 GEMM (rocBLAS, matrix cores) and launches the next event.  n_sk - 1 GEMMs of d x d x 2C per
// launch instead of ~12 mat-vecs per event and chain (the group kernel's dense target: 7e4 events/s at d = 1000).  P x is carried
// by the flows' recurrence (P x += tau P v) and re-based on an exact product every PDMP_DENSE_RESYNC events (default 64).  One wave per chain, layout and state as
// di_segment_kernel.  Against the checker (reference order: a mat-vec per grid time) to a tolerance, as the Zig-Zag forms.
template <int KC>
__device__ __forceinline__ void dr_bps_chain(const DiParams &p, const int64_t j, double *x_tile, const int lane) {
    // (the 32-slot kernel, d = 1025 .. 2048, has no register to spare: 256 + spills into AGPRs with the time horizon compiled in -- the host
    //  refuses the horizon there)
    constexpr bool kUntil = KC < 32;
    const int64_t c = (kUntil && p.chain_map) ? p.chain_map[j] : j;      // (j: output slot of this launch; c: the chain -- as di_segment_kernel)
    DiState cs;
    if (p.k_begin == 0 && !(kUntil && p.resume)) { cs.t = 0.0; cs.horizon = p.tmax; cs.rng_n = 0; cs.dead = 0; cs.pad = 0; }
    else cs = p.st[c];
    if (cs.dead || (kUntil && cs.pad && p.k_begin != 0)) return;      // (pad: the chain has passed t_stop -- in this chunk; a resumed chunk still records its column 0)
    const int d = p.d;
    const bool adaptive = p.adaptive != 0, sgn = p.signed_bound != 0;
    const double refresh = p.refresh, refresh_b = sgn ? p.refresh : 0.0;     // (the signed bound adds refresh_rate to every cell again, AbstractPDMP.jl:104-112)
    double *x = x_tile + lane;                           // slot s at x[s * WAVE]
    double v[KC], gx[KC], gv[KC];
    {
        const double *xin = p.In + (size_t)c * d, *vin = p.In + ((size_t)p.C + (size_t)c) * d;
        const double *gxin = p.Out + (size_t)c * d, *gvin = p.Out + ((size_t)p.C + (size_t)c) * d;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const int i = s * WAVE + lane;
            const bool ok = i < d;
            x[s * WAVE] = ok ? xin[i] : 0.0; v[s] = ok ? vin[i] : 0.0; gx[s] = ok ? gxin[i] : 0.0; gv[s] = ok ? gvin[i] : 0.0;
        }
    }
    ChainRng rng;
    rng.init(p.seed, (uint64_t)(p.chain0 + c));
    rng.restore(cs.rng_n);
    double t = cs.t, horizon = cs.horizon, ts = 0.0, ar = 0.0, E = 0.0, hB = horizon;
    double ev[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int errored = 0, rejected = 0, hits = 0, mode = DB_FRESH;
    int64_t k = p.k_begin, it = 0;
    bool dead = false;

    auto record = [&](int64_t kk) {
        const size_t e = (size_t)j * (size_t)p.n_sk + (size_t)kk;      // (j: the chain's slot in this launch's outputs)
        double *Xe = p.X + e * (size_t)d, *Ve = p.V + e * (size_t)d;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const int i = s * WAVE + lane;
            if (i < d) { PDMP_ROW_STORE(&Xe[i], x[s * WAVE]); PDMP_ROW_STORE(&Ve[i], v[s]); }
        }
        if (lane == 0) {
            p.t[e] = t;
            if (p.H) p.H[e] = horizon;
            if (p.AR) p.AR[e] = ar;
            if (p.EB) p.EB[e] = errored;
            if (p.RJ) p.RJ[e] = rejected;
            if (p.HH) p.HH[e] = hits;
            if (p.EV) for (int q = 0; q < 5; ++q) p.EV[5 * e + q] = ev[q];
        }
    };
    auto flow = [&](double tau) {              // x <- x + v tau (BouncyParticleSamplers.jl:32) and P x with it
#pragma unroll
        for (int s = 0; s < KC; ++s) { x[s * WAVE] = fma(v[s], tau, x[s * WAVE]); gx[s] = fma(tau, gv[s], gx[s]); }
    };

    if (k == 0) { record(0); k = 1; }          // init_state + record of the initial state (sample.jl:271-272)
    while (k < p.k_end && (!kUntil || t < p.t_stop)) {              // (`while state.t < T`, sample.jl:372; t_stop = +inf for the fixed-count path)
        if (++it > p.cap * 4) { dead = true; break; }
        // ---- the rate along the flow from the current state: A + B t
        double pa = 0.0, pb = 0.0;
#pragma unroll
        for (int s = 0; s < KC; ++s) { pa += gx[s] * v[s]; pb += gv[s] * v[s]; }
        const double A = wave_sum(pa), B = wave_sum(pb);
        // ---- bound over [0, harg]: one_step_of_thinning! (:65-85) / erroneous_acceptance_rate! (:131-151) / if_reject! (:188-203)
        const double harg = (mode == DB_ERR) ? horizon / 2 : ((mode == DB_REUSE) ? hB : horizon);
        hB = harg;
        const double h = harg / 9.0;
        const double e1 = rng.exponential();
        E = (mode == DB_REUSE) ? E + e1 : e1;
        double acc = 0.0, cprev = 0.0, bj = 0.0, tp = __builtin_huge_val(), lb = 0.0;
        bool found = false;
        auto point = [&](double tt, double &val, double &slp) {                // the bounded function and its t-derivative, :39-47 / UpperBound.jl:97-102
            const double dv = fma(tt, B, A);
            val = (sgn ? dv : fmax(0.0, dv)) + refresh;                        // includes refresh_rate (Q2)
            slp = sgn ? B : ((0.0 < dv) ? B : 0.0);
        };
        double f0, s0;
        point(0.0, f0, s0);
        for (int j = 0; j < 9 && !found; ++j) {                                // cells in grid order, lazily (upper_bound_grid, UpperBound.jl:92-137)
            double f1, s1;
            point((j == 8) ? harg : (double)(j + 1) * h, f1, s1);
            const double num = fma(s1, h, f0 - f1);                            // :123
            const double den = s1 - s0;
            const double q = (den == 0.0) ? ((num > 0.0) ? h : 0.0) : fmin(fmax(num / den, 0.0), h);   // :124-125
            const double inter = fma(s0, q, f0);                               // :127
            bj = fmax(fmax(fmax(f0, f1), inter), 0.0) + refresh_b;             // :128-131
            acc += bj;
            const double cj = acc * h;                                         // :133-134
            if (!(cj < E)) { found = true; tp = (double)j * h + (E - cprev) / (cj - cprev) * h; }   // next_event, :264-273
            cprev = cj;
            f0 = f1; s0 = s1;
        }
        lb = bj;
        bool propose;
        if (mode == DB_ERR) {                                                  // :131-151
            if (adaptive) horizon = hB;
            errored++;
            { const int slot = errored % 5;
#pragma unroll
              for (int q = 0; q < 5; ++q) if (q == slot) ev[q] = ar; }
            propose = tp < horizon;
        } else if (mode == DB_REUSE) {                                         // :188-203
            if (adaptive) horizon = horizon / 1.04;
            rejected++;
            if (tp > horizon) { flow(horizon); ts += horizon; hits++; }
            propose = tp < horizon;
        } else if (tp > horizon) {                                             // move_to_horizon!, :87-101
            flow(horizon);
            ts += horizon;
            hits++;
            if (adaptive) horizon = horizon * 1.01;
            propose = false;
        } else {
            propose = tp < horizon;
        }
        mode = DB_FRESH;
        if (!propose) continue;
        // ---- proposal at tp (ac_step!, :113-129): lambda = max(0, <grad-U(x + v tp), v>) + refresh_rate
        const double raw = fma(tp, B, A);
        const double lam = fmax(0.0, raw) + refresh;
        ar = lam / lb;
        if (ar > 1.0) { mode = DB_ERR; continue; }
        const double u = rng.uniform();
        if (!(u < ar)) { mode = DB_REUSE; continue; }
        // ---- accepted (if_accept!, :170-186): flow, then velocity_jump (BouncyParticleSamplers.jl:49-74)
        flow(tp);                                                              // gx = grad-U at the event
        const double bounce = fmax(0.0, raw);                                  // dot(grad-U(x), v) = the proposal's sum, :51-52
        const double prob = bounce / (bounce + refresh);                       // NaN if both 0 (Q9): then the refresh branch
        const double u2 = rng.uniform();
        if (u2 < prob) {
            double pg = 0.0;
#pragma unroll
            for (int s = 0; s < KC; ++s) pg += gx[s] * gx[s];
            const double gg = wave_sum(pg);
            if (gg != 0.0) {
                const double scale = 2.0 * raw / gg;                           // :60
#pragma unroll
                for (int s = 0; s < KC; ++s) v[s] = v[s] - scale * gx[s];
            }
        } else {                                                               // randn!, :65-71
            double pz = 0.0;
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                const int i = s * WAVE + lane;
                v[s] = (i < d) ? rng.normal_coord(i) : 0.0;
                pz += v[s] * v[s];
            }
            rng.normals_done(d);
            if (!p.gaussian_velocity) {                                        // normalize!, :71
                const double inv = 1.0 / __dsqrt_rn(wave_sum(pz));
#pragma unroll
                for (int s = 0; s < KC; ++s) v[s] = v[s] * inv;
            }
        }
        t = t + tp + ts;
        ts = 0.0;
        record(k);
        errored = 0; rejected = 0; hits = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) ev[q] = 0.0;
        it = 0;
        ++k;
    }
    {                                          // x, v back into the GEMM's input: P v' comes with the next launch; P x is carried along
        double *xo = p.In + (size_t)c * d, *vo = p.In + ((size_t)p.C + (size_t)c) * d;   // (gx = P x at the event, by the flows' recurrences) and
        double *go = p.Out + (size_t)c * d;                                              // re-based by a full product every PDMP_DENSE_RESYNC events
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const int i = s * WAVE + lane;
            if (i < d) { xo[i] = x[s * WAVE]; vo[i] = v[s]; go[i] = gx[s]; }
        }
    }
    if (lane == 0) {
        cs.t = t; cs.horizon = horizon; cs.rng_n = rng.n; cs.dead = dead ? 1 : 0;
        cs.pad = (kUntil && !(t < p.t_stop)) ? 1 : 0;
        if (kUntil && p.count && (cs.pad || k >= p.n_sk)) p.count[j] = k;      // columns recorded in this chunk (the group kernel's rule: engine_api.hip, UntilArgs)
        p.st[c] = cs;
        if (dead) atomicAdd(&p.status[0], 1);
    }
}
// one wave per chain; a launch of fewer workgroups than chains / 4 walks the chains with the grid's stride (PDMP_DENSE_STEP_BLOCKS: a step
// kernel that leaves room on the CUs for the other lane's product, dense BPS / Boomerang pipeline)
template <int KC>
__global__ __launch_bounds__(256, (KC >= 32 ? 1 : 2)) void dr_bps_kernel(const DiParams p) {
    __shared__ double x_lds[4][KC * WAVE];
    const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int64_t c = p.c_lo + (int64_t)blockIdx.x * 4 + wv; c < p.c_hi; c += (int64_t)gridDim.x * 4) dr_bps_chain<KC>(p, c, x_lds[wv], lane);
}

// ---- Boomerang on a dense precision, the same scheme: between events the flow is a rotation of (x, v) and of (P x, P v) by the same angle
// (16 slots: x, v, P x, P v and the rotation's temporaries need more than 256 registers: one wave per SIMD)
template <int KC>
__device__ __forceinline__ void dr_boom_chain(const DiParams &p, const int64_t j, double *x_tile, const int lane) {
    const int64_t c = p.chain_map ? p.chain_map[j] : j;
    DiState cs;
    if (p.k_begin == 0 && !p.resume) { cs.t = 0.0; cs.horizon = p.tmax; cs.rng_n = 0; cs.dead = 0; cs.pad = 0; }
    else cs = p.st[c];
    if (cs.dead || (cs.pad && p.k_begin != 0)) return;      // (pad: the chain has passed t_stop -- in this chunk; a resumed chunk still records its column 0)
    const int d = p.d;
    const bool adaptive = p.adaptive != 0, sgn = p.signed_bound != 0;
    const double refresh = p.refresh, refresh_b = sgn ? p.refresh : 0.0;     // (the signed bound adds refresh_rate to every cell again, AbstractPDMP.jl:104-112)
    double *x = x_tile + lane;                           // slot s at x[s * WAVE]
    double v[KC], gx[KC], gv[KC];
    {
        const double *xin = p.In + (size_t)c * d, *vin = p.In + ((size_t)p.C + (size_t)c) * d;
        const double *gxin = p.Out + (size_t)c * d, *gvin = p.Out + ((size_t)p.C + (size_t)c) * d;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const int i = s * WAVE + lane;
            const bool ok = i < d;
            x[s * WAVE] = ok ? xin[i] : 0.0; v[s] = ok ? vin[i] : 0.0; gx[s] = ok ? gxin[i] : 0.0; gv[s] = ok ? gvin[i] : 0.0;
        }
    }
    ChainRng rng;
    rng.init(p.seed, (uint64_t)(p.chain0 + c));
    rng.restore(cs.rng_n);
    double t = cs.t, horizon = cs.horizon, ts = 0.0, ar = 0.0, E = 0.0, hB = horizon;
    double ev[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int errored = 0, rejected = 0, hits = 0, mode = DB_FRESH;
    int64_t k = p.k_begin, it = 0;
    bool dead = false;

    auto record = [&](int64_t kk) {
        const size_t e = (size_t)j * (size_t)p.n_sk + (size_t)kk;      // (j: the chain's slot in this launch's outputs)
        double *Xe = p.X + e * (size_t)d, *Ve = p.V + e * (size_t)d;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const int i = s * WAVE + lane;
            if (i < d) { PDMP_ROW_STORE(&Xe[i], x[s * WAVE]); PDMP_ROW_STORE(&Ve[i], v[s]); }
        }
        if (lane == 0) {
            p.t[e] = t;
            if (p.H) p.H[e] = horizon;
            if (p.AR) p.AR[e] = ar;
            if (p.EB) p.EB[e] = errored;
            if (p.RJ) p.RJ[e] = rejected;
            if (p.HH) p.HH[e] = hits;
            if (p.EV) for (int q = 0; q < 5; ++q) p.EV[5 * e + q] = ev[q];
        }
    };
    auto flow = [&](double tau) {              // the rotation (x, v) <- (x cos + v sin, -x sin + v cos), BoomerangSamplers.jl:31, and (P x, P v) with it
        double sn, cn;
        det_sincos(tau, sn, cn);
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const double xo = x[s * WAVE], vo = v[s], go = gx[s], ho = gv[s];
            x[s * WAVE] = xo * cn + vo * sn; v[s] = (-xo) * sn + vo * cn;
            gx[s] = go * cn + ho * sn; gv[s] = (-go) * sn + ho * cn;
        }
    };

    if (k == 0) { record(0); k = 1; }          // init_state + record of the initial state (sample.jl:271-272)
    while (k < p.k_end && t < p.t_stop) {              // (`while state.t < T`, sample.jl:372; t_stop = +inf for the fixed-count path)
        if (++it > p.cap * 4) { dead = true; break; }
        // ---- the rate along the flow from the current state.  The reference's Boomerang RATE uses grad-U itself (BoomerangSamplers.jl:38-41;
        // only the jump uses grad-U - x: Q8 of SURVEY.md 3.7, reproduced), and grad-U = P x is linear, so with gx = P x, gv = P v:
        //   <grad-U(x_t), v_t> = <gx c + gv s, -x s + v c> = A1 c^2 - A2 s^2 + (A3 - A4) c s,   A1 = <gx, v>, A2 = <gv, x>, A3 = <gv, v>, A4 = <gx, x>
        double p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const double xo = x[s * WAVE];
            p1 += gx[s] * v[s]; p2 += gv[s] * xo; p3 += gv[s] * v[s]; p4 += gx[s] * xo;
        }
        const double A1 = wave_sum(p1), A2 = wave_sum(p2), Dq = wave_sum(p3) - wave_sum(p4), A12 = A1 + A2;
        auto rate_at = [&](double tt, double &dv, double &de) {
            double sn, cn;
            det_sincos(tt, sn, cn);
            const double cc = cn * cn, ss = sn * sn, cs2 = cn * sn;
            dv = A1 * cc - A2 * ss + Dq * cs2;
            de = Dq * (cc - ss) - 2.0 * cs2 * A12;                             // its t-derivative (the forward-mode slope)
        };
        // ---- bound over [0, harg]: one_step_of_thinning! (:65-85) / erroneous_acceptance_rate! (:131-151) / if_reject! (:188-203)
        const double harg = (mode == DB_ERR) ? horizon / 2 : ((mode == DB_REUSE) ? hB : horizon);
        hB = harg;
        const double h = harg / 9.0;
        const double e1 = rng.exponential();
        E = (mode == DB_REUSE) ? E + e1 : e1;
        double acc = 0.0, cprev = 0.0, bj = 0.0, tp = __builtin_huge_val(), lb = 0.0;
        bool found = false;
        auto point = [&](double tt, double &val, double &slp) {                // the bounded function and its t-derivative, BoomerangSamplers.jl:38-46 / UpperBound.jl:97-102
            double dv, de;
            rate_at(tt, dv, de);
            val = (sgn ? dv : fmax(0.0, dv)) + refresh;                        // includes refresh_rate (Q2)
            slp = sgn ? de : ((0.0 < dv) ? de : 0.0);
        };
        double f0, s0;
        point(0.0, f0, s0);
        for (int j = 0; j < 9 && !found; ++j) {                                // cells in grid order, lazily (upper_bound_grid, UpperBound.jl:92-137)
            double f1, s1;
            point((j == 8) ? harg : (double)(j + 1) * h, f1, s1);
            const double num = fma(s1, h, f0 - f1);                            // :123
            const double den = s1 - s0;
            const double q = (den == 0.0) ? ((num > 0.0) ? h : 0.0) : fmin(fmax(num / den, 0.0), h);   // :124-125
            const double inter = fma(s0, q, f0);                               // :127
            bj = fmax(fmax(fmax(f0, f1), inter), 0.0) + refresh_b;             // :128-131
            acc += bj;
            const double cj = acc * h;                                         // :133-134
            if (!(cj < E)) { found = true; tp = (double)j * h + (E - cprev) / (cj - cprev) * h; }   // next_event, :264-273
            cprev = cj;
            f0 = f1; s0 = s1;
        }
        lb = bj;
        bool propose;
        if (mode == DB_ERR) {                                                  // :131-151
            if (adaptive) horizon = hB;
            errored++;
            { const int slot = errored % 5;
#pragma unroll
              for (int q = 0; q < 5; ++q) if (q == slot) ev[q] = ar; }
            propose = tp < horizon;
        } else if (mode == DB_REUSE) {                                         // :188-203
            if (adaptive) horizon = horizon / 1.04;
            rejected++;
            if (tp > horizon) { flow(horizon); ts += horizon; hits++; }
            propose = tp < horizon;
        } else if (tp > horizon) {                                             // move_to_horizon!, :87-101
            flow(horizon);
            ts += horizon;
            hits++;
            if (adaptive) horizon = horizon * 1.01;
            propose = false;
        } else {
            propose = tp < horizon;
        }
        mode = DB_FRESH;
        if (!propose) continue;
        // ---- proposal at tp (ac_step!, :113-129): lambda = max(0, <grad-U(x_tp), v_tp>) + refresh_rate
        double raw, raw_e;
        rate_at(tp, raw, raw_e);
        const double lam = fmax(0.0, raw) + refresh;
        ar = lam / lb;
        if (ar > 1.0) { mode = DB_ERR; continue; }
        const double u = rng.uniform();
        if (!(u < ar)) { mode = DB_REUSE; continue; }
        // ---- accepted (if_accept!, :170-186): flow, then velocity_jump (BoomerangSamplers.jl:49-67)
        flow(tp);                                                              // gx - x = grad-U_effective at the event
        double pgv = 0.0;                                                      // dot(grad-U(x) - x, v): the JUMP's effective gradient, :51-52 (Q8)
#pragma unroll
        for (int s = 0; s < KC; ++s) pgv += (gx[s] - x[s * WAVE]) * v[s];
        const double bounce = fmax(0.0, wave_sum(pgv));
        const double prob = bounce / (bounce + refresh);                       // NaN if both 0 (Q9): then the refresh branch
        const double u2 = rng.uniform();
        if (u2 < prob) {                                                       // reflection on the unit normal e = grad-U_eff / |grad-U_eff|, :58-59
            double pg = 0.0;
#pragma unroll
            for (int s = 0; s < KC; ++s) { const double ge = gx[s] - x[s * WAVE]; pg += ge * ge; }
            const double nrm = __dsqrt_rn(wave_sum(pg));
            double pe = 0.0;
#pragma unroll
            for (int s = 0; s < KC; ++s) pe += v[s] * ((gx[s] - x[s * WAVE]) / nrm);
            const double ve = 2.0 * wave_sum(pe);
#pragma unroll
            for (int s = 0; s < KC; ++s) v[s] = v[s] - ve * ((gx[s] - x[s * WAVE]) / nrm);
        } else {                                                               // randn, :61-65
            double pz = 0.0;
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                const int i = s * WAVE + lane;
                v[s] = (i < d) ? rng.normal_coord(i) : 0.0;
                pz += v[s] * v[s];
            }
            rng.normals_done(d);
            (void)pz;
        }
        t = t + tp + ts;
        ts = 0.0;
        record(k);
        errored = 0; rejected = 0; hits = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) ev[q] = 0.0;
        it = 0;
        ++k;
    }
    {                                          // x, v back into the GEMM's input: P v' comes with the next launch; P x is carried along
        double *xo = p.In + (size_t)c * d, *vo = p.In + ((size_t)p.C + (size_t)c) * d;   // (gx = P x at the event, by the flows' recurrences) and
        double *go = p.Out + (size_t)c * d;                                              // re-based by a full product every PDMP_DENSE_RESYNC events
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const int i = s * WAVE + lane;
            if (i < d) { xo[i] = x[s * WAVE]; vo[i] = v[s]; go[i] = gx[s]; }
        }
    }
    if (lane == 0) {
        cs.t = t; cs.horizon = horizon; cs.rng_n = rng.n; cs.dead = dead ? 1 : 0;
        cs.pad = !(t < p.t_stop) ? 1 : 0;
        if (p.count && (cs.pad || k >= p.n_sk)) p.count[j] = k;      // columns recorded in this chunk (the group kernel's rule: engine_api.hip, UntilArgs)
        p.st[c] = cs;
        if (dead) atomicAdd(&p.status[0], 1);
    }
}
// one wave per chain; a launch of fewer workgroups than chains / 4 walks the chains with the grid's stride (PDMP_DENSE_STEP_BLOCKS: a step
// kernel that leaves room on the CUs for the other lane's product, dense BPS / Boomerang pipeline)
template <int KC>
__global__ __launch_bounds__(256, (KC >= 16 ? 1 : 2)) void dr_boom_kernel(const DiParams p) {
    __shared__ double x_lds[4][KC * WAVE];
    const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int64_t c = p.c_lo + (int64_t)blockIdx.x * 4 + wv; c < p.c_hi; c += (int64_t)gridDim.x * 4) dr_boom_chain<KC>(p, c, x_lds[wv], lane);
}

PDMP_NS_END

using namespace pdmp;

// Diagnostics of the last batched run of this process (pdmp_dense_batched_last_*): atomics, written by whichever handle ran last.
static std::atomic<long long> g_last_phases{0}, g_last_columns{0};
static std::atomic<int> g_last_mode{0};
// Everything a batched dense run needs beside its workspace -- the rocBLAS handles (one per pipeline lane; rocblas_set_stream on a handle
// shared between two engine handles would race), the second stream of a two-lane run and the events that fork / join it -- belongs to
// the ENGINE HANDLE that runs (round 5; rounds 1-4 kept them per device, process-wide): `pdmp_dense_bind_context(&handle->dense_ctx)` on
// the calling thread right before a run, created on first use, released by pdmp_dense_destroy_context when the handle is destroyed.
// "Distinct handles may be used from distinct host threads" (include/pdmp_hip.h) holds for the dense paths too.
struct DbSide { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
struct DenseCtx { rocblas_handle hb[2] = {nullptr, nullptr}; DbSide side; };
static thread_local DenseCtx **g_ctx_slot = nullptr;
extern "C" void pdmp_dense_bind_context(void **slot) { g_ctx_slot = (DenseCtx **)slot; }
extern "C" void pdmp_dense_destroy_context(void *ctx_v) {
    DenseCtx *c = (DenseCtx *)ctx_v;
    if (!c) return;
    for (int l = 0; l < 2; ++l) if (c->hb[l]) (void)rocblas_destroy_handle(c->hb[l]);
    if (c->side.fork) (void)hipEventDestroy(c->side.fork);
    if (c->side.join) (void)hipEventDestroy(c->side.join);
    if (c->side.stream) (void)hipStreamDestroy(c->side.stream);
    delete c;
}
static DenseCtx *db_ctx() {
    static DenseCtx *fallback = nullptr;               // a caller that binds nothing (none in this library): one process-wide context
    static std::mutex mu;
    if (!g_ctx_slot) { std::lock_guard<std::mutex> lock(mu); if (!fallback) fallback = new DenseCtx(); return fallback; }
    if (!*g_ctx_slot) *g_ctx_slot = new DenseCtx();
    return *g_ctx_slot;
}
static rocblas_handle db_handle(int device, int lane = 0) {        // the bound context's rocBLAS handle of a pipeline lane, created on first use
    (void)device;                                                  // (created with the handle's device current: the run functions set it)
    if (lane < 0 || lane > 1) return nullptr;
    DenseCtx *c = db_ctx();
    if (!c->hb[lane] && rocblas_create_handle(&c->hb[lane]) != rocblas_status_success) c->hb[lane] = nullptr;
    return c->hb[lane];
}
// the second stream of a two-lane run and the events that fork / join it (the bound context's, created on first use)
static DbSide *db_side(int device) {
    (void)device;
    DbSide &q = db_ctx()->side;
    if (!q.stream) {
        if (hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking) != hipSuccess) { q.stream = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&q.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&q.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    return &q;
}
extern "C" long long pdmp_dense_batched_last_phases(void) { return g_last_phases; }   // GEMM phases of the last run (diagnostics)
extern "C" long long pdmp_dense_batched_last_columns(void) { return g_last_columns; } // right-hand-side columns multiplied over those phases

// The product P * B of the batched forms: the hand-written fp64 MFMA kernel (dgemm_mfma.hip; gemm_impl = 1, even d) or rocBLAS (the A/B
// baseline; odd d).  Expands where DB_TRY / DB_BLAS, hb, one, zero, d, P_dev and stream are in scope.
#define DB_GEMM(ncols, Bp, Op) do {                                                                                              \
        if (gemm_impl == 1 && (d & 1) == 0) DB_TRY((hipError_t)pdmp_dgemm_mfma_polite(gemm_polite, d, (ncols), d, P_dev, d, (Bp), d, (Op), d, stream)); \
        else DB_BLAS(rocblas_dgemm(hb, rocblas_operation_none, rocblas_operation_none, d, (ncols), d, &one, P_dev, d, (Bp), d, &zero, (Op), d)); \
    } while (0)

// Host driver.  Returns hipSuccess-like 0, or a negative code: -3 runtime, -5 unsupported.
extern "C" int pdmp_dense_batched_run(int device, int d, int signed_bound, int adaptive, double tmax, int64_t cap,
                                      const double *P_dev, int64_t C, int64_t n_sk, int64_t chain0, uint64_t seed,
                                      const double *xinit, const double *vinit, double *X, double *V, double *t,
                                      double *H, double *AR, int32_t *EB, double *EV, int32_t *RJ, int32_t *HH,
                                      int32_t *status, void *stream_v, int full_pv, int gemm_impl, char *err, int errlen) {
    hipStream_t stream = (hipStream_t)stream_v;
    const int gemm_polite = 0;
#define DB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #expr, hipGetErrorString(e_)); goto fail; } } while (0)
#define DB_BLAS(expr) do { rocblas_status s_ = (expr); if (s_ != rocblas_status_success) { snprintf(err, errlen, "%s: rocblas status %d", #expr, (int)s_); goto fail; } } while (0)
    double *Xs = nullptr, *In = nullptr, *Out = nullptr, *Gprev = nullptr;
    DbState *st = nullptr;
    int32_t *remaining = nullptr;
    int rc = -3;
    {
        const size_t n = (size_t)C * (size_t)d;
        DB_TRY(hipMalloc(&Xs, n * 8));
        DB_TRY(hipMalloc(&In, 2 * n * 8));
        DB_TRY(hipMalloc(&Out, 2 * n * 8));
        DB_TRY(hipMalloc(&Gprev, n * 8));
        DB_TRY(hipMalloc(&st, (size_t)C * sizeof(DbState)));
        DB_TRY(hipMalloc(&remaining, sizeof(int32_t)));
        DB_TRY(hipMemsetAsync(remaining, 0, sizeof(int32_t), stream));
        DB_TRY(hipMemsetAsync(Out, 0, 2 * n * 8, stream));
        rocblas_handle hb = db_handle(device);
        if (!hb) { snprintf(err, errlen, "rocblas handle (device ordinal %d)", device); goto fail; }
        DB_BLAS(rocblas_set_stream(hb, stream));
        DB_BLAS(rocblas_set_pointer_mode(hb, rocblas_pointer_mode_host));

        DbParams p;
        p.d = d; p.signed_bound = signed_bound; p.adaptive = adaptive;
        p.C = C; p.n_sk = n_sk; p.chain0 = chain0; p.cap = cap; p.seed = seed; p.tmax = tmax;
        p.xinit = xinit; p.vinit = vinit; p.Xs = Xs; p.In = In; p.Out = Out; p.Gprev = Gprev; p.st = st; p.P = P_dev;
        p.X = X; p.V = V; p.t = t; p.H = H; p.AR = AR; p.EV = EV; p.EB = EB; p.RJ = RJ; p.HH = HH;
        p.status = status; p.remaining = remaining;
        const dim3 grid((unsigned)((C + 3) / 4)), block(256);
        hipLaunchKernelGGL(db_init_kernel, grid, block, 0, stream, p);
        DB_TRY(hipGetLastError());
        const double one = 1.0, zero = 0.0;
        // a chain needs ~1.4 cells + 1 proposal per event (+ horizon moves, rejections): budget 16 phases per event
        const int64_t max_phases = 16 * n_sk + 64;
        int32_t live = 1;
        g_last_phases = 0;
        g_last_columns = 0;
        g_last_mode = 1;
        for (int64_t ph = 0; ph < max_phases && live > 0; ++ph) {
            ++g_last_phases;
            // Out (d x C) = P (d x d) * requests (d x C): grad-U of every request.  The velocity half, P v of every chain, changes only
            // at flips, where the step kernel updates it by a rank-one term; a full d x 2C product re-syncs it every 32 phases.
            const bool full = (ph % 32) == 0 || full_pv;
            g_last_columns += full ? 2 * C : C;
            DB_GEMM((int)(full ? 2 * C : C), In, Out);
            hipLaunchKernelGGL(db_step_kernel, grid, block, 0, stream, p);
            if ((ph & 7) == 7) {
                DB_TRY(hipMemcpyAsync(&live, remaining, sizeof live, hipMemcpyDeviceToHost, stream));
                DB_TRY(hipStreamSynchronize(stream));
            }
        }
        DB_TRY(hipGetLastError());
        DB_TRY(hipMemcpyAsync(&live, remaining, sizeof live, hipMemcpyDeviceToHost, stream));
        DB_TRY(hipStreamSynchronize(stream));
        if (live > 0) { snprintf(err, errlen, "%d chain(s) did not finish within the phase budget", (int)live); rc = -4; goto fail; }
        rc = 0;
    }
fail:
    (void)hipStreamSynchronize(stream);
    (void)hipFree(Xs); (void)hipFree(In); (void)hipFree(Out); (void)hipFree(Gprev); (void)hipFree(st); (void)hipFree(remaining);
    return rc;
#undef DB_TRY
#undef DB_BLAS
}

extern "C" int pdmp_dense_batched_last_mode(void) { return g_last_mode; }   // 0 none, 1 GEMM per phase, 2 incremental (Zig-Zag), 3 one GEMM per event (BPS)


// Incremental driver: segments of `resync` events (PDMP_DENSE_RESYNC, default 256); before each segment one fp64 GEMM
// [P x | P v] = P [x | v] over all chains.  Asynchronous on `stream` (nothing is read back: a chain that hits the iteration cap
// is counted in status[0], which pdmp_check_status reports); the workspace (*ws, *ws_cap bytes) belongs to the caller's handle
// and is grown here on demand.  Returns 0, -3 (runtime error, message in err) or -5 when d > 1024 (the caller falls back).
extern "C" int pdmp_dense_incremental_run(int device, int d, int signed_bound, int adaptive, double tmax, int64_t cap,
                                          const double *P_dev, int64_t C, int64_t n_sk, int64_t chain0, uint64_t seed,
                                          const double *xinit, const double *vinit, double *X, double *V, double *t,
                                          double *H, double *AR, int32_t *EB, double *EV, int32_t *RJ, int32_t *HH,
                                          int32_t *status, void *stream_v, void **ws, size_t *ws_cap, int64_t resync_events, int gemm_impl,
                                          double t_stop, int resume, int64_t n_launch, const int64_t *chain_map, int64_t *count, char *err, int errlen) {
    // (time horizon: C = the chains of the workspace = the first chunk's; n_launch <= C output slots in this chunk, chain_map their chains)
    hipStream_t stream = (hipStream_t)stream_v;
    // the unsigned bound keeps the tangent intersection's division: its 16-slot instantiation does not fit the register file
    // without spilling (and no kernel with lane-dependent branches may spill, see GroupOcc in group_kernel.hpp)
    if (d > 16 * WAVE || (!signed_bound && d > 8 * WAVE)) return -5;
    const int gemm_polite = 0;
#define DB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #expr, hipGetErrorString(e_)); return -3; } } while (0)
#define DB_BLAS(expr) do { rocblas_status s_ = (expr); if (s_ != rocblas_status_success) { snprintf(err, errlen, "%s: rocblas status %d", #expr, (int)s_); return -3; } } while (0)
    const size_t n = (size_t)C * (size_t)d;
    const size_t need = 4 * n * 8 + (size_t)C * sizeof(DiState);
    if (resume && (!*ws || need > *ws_cap)) { snprintf(err, errlen, "resume: no workspace of a first chunk of %lld chains", (long long)C); return -3; }
    if (need > *ws_cap) {
        if (*ws) { DB_TRY(hipStreamSynchronize(stream)); DB_TRY(hipFree(*ws)); *ws = nullptr; *ws_cap = 0; }
        DB_TRY(hipMalloc(ws, need));
        *ws_cap = need;
    }
    double *In = (double *)*ws, *Out = In + 2 * n;
    DiState *st = (DiState *)(Out + 2 * n);
    const int64_t resync = resync_events >= 1 ? resync_events : 256;   // (policy: PDMP_DENSE_RESYNC)
    rocblas_handle hb = db_handle(device);
    if (!hb) { snprintf(err, errlen, "rocblas handle"); return -3; }
    if (!resume) {
        DB_TRY(hipMemcpyAsync(In, xinit, n * 8, hipMemcpyDeviceToDevice, stream));
        DB_TRY(hipMemcpyAsync(In + n, vinit, n * 8, hipMemcpyDeviceToDevice, stream));
    }
    DB_BLAS(rocblas_set_stream(hb, stream));
    DB_BLAS(rocblas_set_pointer_mode(hb, rocblas_pointer_mode_host));
    DiParams p;
    p.d = d; p.signed_bound = signed_bound; p.adaptive = adaptive;
    p.C = C; p.n_sk = n_sk; p.chain0 = chain0; p.cap = cap; p.seed = seed; p.tmax = tmax;
    p.In = In; p.Out = Out; p.P = P_dev; p.st = st;
    p.X = X; p.V = V; p.t = t; p.H = H; p.AR = AR; p.EV = EV; p.EB = EB; p.RJ = RJ; p.HH = HH;
    p.status = status;
    // gemm_impl, bits 8 - 9: PDMP_DENSE_DEBUG of the handle's launch policy (measurements only, the results are wrong): 1 = di_segment_kernel
    // without its row stores, 2 = without its loads of P's columns -- what the launch costs beside them
    const int debug_mask = (gemm_impl >> 8) & 3;
    gemm_impl &= 0xff;
    p.refresh = 0.0; p.gaussian_velocity = 0; p.pad = debug_mask;
    p.t_stop = t_stop; p.chain_map = chain_map; p.count = count; p.resume = resume; p.pad2 = 0;
    const dim3 grid((unsigned)((n_launch + 3) / 4)), block(256);
    const double one = 1.0, zero = 0.0;
    g_last_phases = 0; g_last_columns = 0; g_last_mode = 2;
    for (int64_t k0 = 0; k0 < n_sk;) {
        const int64_t from = (k0 == 0) ? 1 : k0, k1 = (from + resync < n_sk) ? from + resync : n_sk;
        DB_GEMM((int)(2 * C), In, Out);
        g_last_columns += 2 * C;
        p.k_begin = k0; p.k_end = k1;
        p.c_lo = 0; p.c_hi = n_launch;
        const bool until = count != nullptr || resume || chain_map != nullptr || t_stop < __builtin_huge_val();
        if (until) {
            if (signed_bound) {
                if (d <= 4 * WAVE) hipLaunchKernelGGL((di_segment_kernel<4, true, true>), grid, block, 0, stream, p);
                else if (d <= 8 * WAVE) hipLaunchKernelGGL((di_segment_kernel<8, true, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((di_segment_kernel<16, true, true>), grid, block, 0, stream, p);
            } else {
                if (d <= 4 * WAVE) hipLaunchKernelGGL((di_segment_kernel<4, false, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((di_segment_kernel<8, false, true>), grid, block, 0, stream, p);
            }
        } else if (signed_bound) {
            if (d <= 4 * WAVE) hipLaunchKernelGGL((di_segment_kernel<4, true>), grid, block, 0, stream, p);
            else if (d <= 8 * WAVE) hipLaunchKernelGGL((di_segment_kernel<8, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((di_segment_kernel<16, true>), grid, block, 0, stream, p);
        } else {
            if (d <= 4 * WAVE) hipLaunchKernelGGL((di_segment_kernel<4, false>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((di_segment_kernel<8, false>), grid, block, 0, stream, p);
        }
        DB_TRY(hipGetLastError());
        k0 = k1;
    }
    return 0;
#undef DB_TRY
#undef DB_BLAS
}

// BPS / Boomerang on a dense precision: one GEMM + one launch of dr_bps_kernel / dr_boom_kernel per skeleton column (the first launch records columns 0 and 1).
// Asynchronous on `stream`, workspace as pdmp_dense_incremental_run.  Returns 0, -3, or -5 when d > 2048.
extern "C" int pdmp_dense_bps_run(int device, int boomerang, int d, int signed_bound, int adaptive, int gaussian_velocity, double tmax, double refresh,
                                  int64_t cap, const double *P_dev, int64_t C, int64_t n_sk, int64_t chain0, uint64_t seed,
                                  const double *xinit, const double *vinit, double *X, double *V, double *t,
                                  double *H, double *AR, int32_t *EB, double *EV, int32_t *RJ, int32_t *HH,
                                  int32_t *status, void *stream_v, void **ws, size_t *ws_cap, int64_t resync_events, int gemm_impl, int lanes,
                                  double t_stop, int resume, int64_t n_launch, const int64_t *chain_map, int64_t *count, char *err, int errlen) {
    hipStream_t stream = (hipStream_t)stream_v;
    if (d > (boomerang ? 24 : 32) * WAVE) return -5;
#define DB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { snprintf(err, errlen, "%s: %s", #expr, hipGetErrorString(e_)); return -3; } } while (0)
#define DB_BLAS(expr) do { rocblas_status s_ = (expr); if (s_ != rocblas_status_success) { snprintf(err, errlen, "%s: rocblas status %d", #expr, (int)s_); return -3; } } while (0)
    const size_t n = (size_t)C * (size_t)d;
    const size_t need = 4 * n * 8 + (size_t)C * sizeof(DiState);
    if (resume && (!*ws || need > *ws_cap)) { snprintf(err, errlen, "resume: no workspace of a first chunk of %lld chains", (long long)C); return -3; }
    if (need > *ws_cap) {
        if (*ws) { DB_TRY(hipStreamSynchronize(stream)); DB_TRY(hipFree(*ws)); *ws = nullptr; *ws_cap = 0; }
        DB_TRY(hipMalloc(ws, need));
        *ws_cap = need;
    }
    double *In = (double *)*ws, *Out = In + 2 * n;
    DiState *st = (DiState *)(Out + 2 * n);
    rocblas_handle hb = db_handle(device);
    if (!hb) { snprintf(err, errlen, "rocblas handle"); return -3; }
    if (!resume) {
        DB_TRY(hipMemcpyAsync(In, xinit, n * 8, hipMemcpyDeviceToDevice, stream));
        DB_TRY(hipMemcpyAsync(In + n, vinit, n * 8, hipMemcpyDeviceToDevice, stream));
    }
    DB_BLAS(rocblas_set_stream(hb, stream));
    DB_BLAS(rocblas_set_pointer_mode(hb, rocblas_pointer_mode_host));
    DiParams p;
    p.d = d; p.signed_bound = signed_bound; p.adaptive = adaptive;
    p.C = C; p.n_sk = n_sk; p.chain0 = chain0; p.cap = cap; p.seed = seed; p.tmax = tmax;
    p.In = In; p.Out = Out; p.P = P_dev; p.st = st;
    p.X = X; p.V = V; p.t = t; p.H = H; p.AR = AR; p.EV = EV; p.EB = EB; p.RJ = RJ; p.HH = HH;
    p.status = status;
    p.refresh = refresh; p.gaussian_velocity = gaussian_velocity; p.pad = 0;
    p.t_stop = t_stop; p.chain_map = chain_map; p.count = count; p.resume = resume; p.pad2 = 0;
    const dim3 block(256);
    const double one = 1.0, zero = 0.0;
    g_last_phases = 0; g_last_columns = 0; g_last_mode = 3;
    const int64_t resync = resync_events >= 1 ? resync_events : 64;   // events between two full products P [x | v]; in between only P v' is multiplied (policy: PDMP_DENSE_RESYNC)
    // Chains are independent: the ensemble is cut into TWO half-ensembles that run the same event loop on two streams.  Measured (d = 1000,
    // 10^4 chains x 200): 108 -> 93 ms for BPS.  What it buys is mostly the library's better rate on the half-sized product (65 against 51
    // TFLOP/s) and the launch gaps it fills; the products and the step kernels of the two lanes do NOT end up side by side (see the note in the
    // loop below).  Same arithmetic per chain: bit-identical to one lane with the hand-written product (rocBLAS may pick another kernel for
    // another column count).
    const int nl = (lanes >= 2 && C >= 256 && !chain_map && n_launch == C) ? 2 : 1;   // (a compacted chunk: its slots are not a contiguous range of the workspace's columns)
    static const int polite_env = getenv("PDMP_DGEMM_POLITE") ? atoi(getenv("PDMP_DGEMM_POLITE")) : 2;   // (tuning: workgroups of the product per CU beside the other lane's step kernel)
    const int gemm_polite = (nl == 2) ? polite_env : 0;
    // workgroups of a step kernel (PDMP_DENSE_STEP_BLOCKS; 0 = one wave per chain).  The kernels that fit one workgroup per CU only (Boomerang
    // from 9 slots on, BPS with 32: 256 registers) run as ONE persistent workgroup per CU beside the other lane's product: measured at
    // d = 1000, 10^4 chains x 200, Boomerang: 115.2 -> 103.7 ms (192 / 320 workgroups: 107.6 / 111.5); BPS with 16 slots (two workgroups per
    // CU): 94 - 96 ms whatever the count
    static const int step_env = getenv("PDMP_DENSE_STEP_BLOCKS") ? atoi(getenv("PDMP_DENSE_STEP_BLOCKS")) : -1;
    int step_blocks = 0;
    if (nl == 2) {
        step_blocks = step_env >= 0 ? step_env : 0;
        if (step_env < 0 && (boomerang ? d > 8 * WAVE : d > 16 * WAVE)) {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) step_blocks = cus;
        }
    }
    DbSide *side = (nl == 2) ? db_side(device) : nullptr;
    if (nl == 2 && !side) { snprintf(err, errlen, "side stream"); return -3; }
    hipStream_t ls[2] = {stream, side ? side->stream : stream};
    rocblas_handle lh[2] = {hb, (nl == 2) ? db_handle(device, 1) : hb};
    if (!lh[1]) { snprintf(err, errlen, "rocblas handle"); return -3; }
    if (nl == 2) {
        DB_BLAS(rocblas_set_stream(lh[1], ls[1]));
        DB_BLAS(rocblas_set_pointer_mode(lh[1], rocblas_pointer_mode_host));
        DB_TRY(hipEventRecord(side->fork, stream));                       // the side lane starts after the initial states are in the workspace
        DB_TRY(hipStreamWaitEvent(ls[1], side->fork, 0));
    }
    const int64_t cut = (nl == 2) ? ((C / 2 + 3) / 4 * 4) : n_launch;    // (whole blocks of four chains per lane)
    for (int64_t k0 = 0, ph = 0; k0 < n_sk; ++ph) {
        const int64_t k1 = (k0 == 0) ? ((n_sk < 2) ? n_sk : 2) : k0 + 1;
        const bool full = (ph % resync) == 0;
        ++g_last_phases;
        g_last_columns += full ? 2 * C : C;
        p.k_begin = k0; p.k_end = k1;
        for (int l = 0; l < nl; ++l) {
            const int64_t lo = (l == 0) ? 0 : cut, hi = (l == 0) ? cut : n_launch;
            if (hi <= lo) continue;
            hipStream_t stream = ls[l];                                    // (DB_GEMM and the launches below use this lane's stream and handle)
            rocblas_handle hb = lh[l];
            const size_t off = (size_t)lo * (size_t)d;
            const int cols = (nl == 2) ? (int)(hi - lo) : (int)C;          // (one lane: every column of the workspace -- a compacted chunk's chains lie anywhere in it)
            // (what the two lanes buy is the library's better rate on the half-sized product, 65 against 51 TFLOP/s -- the kernel trace shows the
            //  lanes running IN phase, product beside product and step beside step, and the launch costs the sum of the two kinds of work;
            //  round 4 tried to change that and could not: products forced to take turns across the lanes 94.6 -> 112 ms, a step kernel with
            //  110 registers and no LDS that could share a SIMD with the product 95.4 -> 96.1 ms: DESIGN.md 4.3c)
            if (full) DB_GEMM(cols, In + off, Out + off);                  // P x of this lane's chains
            DB_GEMM(cols, In + n + off, Out + n + off);                    // P v'
            p.c_lo = lo; p.c_hi = hi;
            const unsigned full_grid = (unsigned)((hi - lo + 3) / 4);
            const dim3 grid((step_blocks > 0 && (unsigned)step_blocks < full_grid) ? (unsigned)step_blocks : full_grid);
            if (boomerang) {
                if (d <= 4 * WAVE) hipLaunchKernelGGL((dr_boom_kernel<4>), grid, block, 0, stream, p);
                else if (d <= 8 * WAVE) hipLaunchKernelGGL((dr_boom_kernel<8>), grid, block, 0, stream, p);
                else if (d <= 16 * WAVE) hipLaunchKernelGGL((dr_boom_kernel<16>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((dr_boom_kernel<24>), grid, block, 0, stream, p);      // d <= 1536 (32 slots of v, P x, P v spill)
            } else {
                if (d <= 4 * WAVE) hipLaunchKernelGGL((dr_bps_kernel<4>), grid, block, 0, stream, p);
                else if (d <= 8 * WAVE) hipLaunchKernelGGL((dr_bps_kernel<8>), grid, block, 0, stream, p);
                else if (d <= 16 * WAVE) hipLaunchKernelGGL((dr_bps_kernel<16>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((dr_bps_kernel<32>), grid, block, 0, stream, p);       // d <= 2048: 64 KB of LDS per block, one block per CU
            }
            DB_TRY(hipGetLastError());
        }
        k0 = k1;
    }
    if (nl == 2) {
        DB_TRY(hipEventRecord(side->join, ls[1]));                        // the caller's stream ends after both lanes
        DB_TRY(hipStreamWaitEvent(stream, side->join, 0));
    }
    return 0;
#undef DB_TRY
#undef DB_BLAS
}
