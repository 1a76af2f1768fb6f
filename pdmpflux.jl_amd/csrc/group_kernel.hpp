// group_kernel.hpp -- the fast path of sample_skeleton() (src/sample.jl:253-284) for grid_size = 10
// and forward-mode slopes: G lanes per chain (G = 1,2,4,...,64 chosen from d), 64/G chains per wave.
//
// Why this layout (profiles/r1): the wave-per-chain kernel (skeleton_kernel.hpp) is VALU-issue bound
// (94 % VALU busy, 1100 wave-instructions per event) and most of those instructions are per-chain
// overhead replicated over 64 lanes: wave reductions of the nine box_max sums, the scalar
// next_event / RNG / log work, 14 idle lanes at d = 50.  Here
//   * coordinate i of a chain lives in LDS slot i/G of lane (i mod G) of the chain's lane group
//     (conflict-free ds_read_b64: consecutive lanes, consecutive 8-byte words);
//   * each lane walks its d/G coordinates sequentially and accumulates one box_max cell of the
//     BoundBox (src/Composites.jl:15-20) at a time; cells are built lazily in grid order until the
//     exponential is covered (bit-identical to the full bound, see bound_invert_vect), so a cell costs
//     log2(G) cross-lane adds and the per-chain scalar work is shared by 64/G chains per instruction;
//   * with G = 1 every sum over coordinates runs in exactly the reference's left-to-right order.
// The thinning state machine (src/SamplingLoopInplace.jl:27-217) is flattened into two phases per
// lane -- BOUND (one_step_of_thinning! / erroneous_acceptance_rate!) and PROPOSE (ac_step!) -- so the
// chains of a wave stay in lock step: with the adaptive horizon a chain needs ~1.2 bounds and ~1.04
// proposals per event, so almost every iteration every lane does BOUND then PROPOSE -> accept.
//
// Arithmetic contract ("GROUP(G)" order of the checker): a sum over coordinates is, per lane, the
// left-to-right sum of the coordinates it owns (i = g, g+G, ...), then an ascending xor-butterfly
// over the G lanes; cum_sum is the running sum of those box_max totals times h (the reference sums
// per-dimension cumsums, UpperBound.jl:243-246: equal up to rounding); the Zig-Zag flip index walks
// slots of G coordinates with a Kogge-Stone scan inside a slot and a running carry between slots.
#pragma once
#include "skeleton_kernel.hpp"
#include "user_ad.hpp"

PDMP_NS_BEGIN

#ifdef PDMP_NO_DPP_BOUND        // A/B builds
constexpr bool defined_no_dpp_bound = true;
#else
constexpr bool defined_no_dpp_bound = false;
#endif
constexpr int GNG = 10;   // grid_size of the register / lane forms (the default of every reference constructor); the LDS form takes any 2 .. 64 at run time
constexpr int KREG = 13;  // slots per lane of the register form (x, v in VGPRs): d <= 13 G

struct GroupParams {
    SkelParams s;
    int32_t G;            // lanes per chain (power of two, 1..64)
    int32_t K;            // coordinate slots per lane = ceil(d / G)
    int32_t tpar;         // doubles of target parameters staged in LDS (0 or 2d)
    int32_t tiles;        // K*64-double LDS tiles per wave: 2 (x, v) + the target's scratch tiles
    int32_t cell_budget;  // cells of a lazily built Zig-Zag bound per pass of the state machine (1..9)
    // dynamic schedule: work unit u = (batch b = u % nb of 64/G chains, segment r = u / nb of seg_len events)
    int32_t nseg;         // segments per chain
    int64_t nb;           // batches = ceil(C / (64/G))
    int64_t seg_len;      // events per segment
    int32_t *ctr;         // unit counter (zeroed before the launch)
    int32_t *progress;    // progress[b] = segments of batch b completed (zeroed before the launch)
    double *state;        // per chain: x[d], v[d], t, horizon, ar, rng counter  (stride 2d+4 doubles)
    const int32_t *only_if;   // null, or a device flag: the register-form kernels return at once while it is 0 (lane_kernel.hpp ran instead)
    int32_t host_flags;       // read by the host-side launchers only (the handle's launch policy): 1 no lean instantiation, 2 print the launch geometry
};

// ---------------------------------------------------------------- group-of-lanes primitives (G uniform)
// Stages inside a row of 16 lanes use DPP (register-to-register, no LDS crossbar latency); only the
// 16- and 32-lane stages of a SUM use the permlane swaps of gfx950 (wave_math.hpp), of a SCAN ds_bpermute.  Same operands and order as
// __shfl_xor / __shfl_up.
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;

__device__ __forceinline__ double group_sum(double a, int G) {       // ascending xor-butterfly over G lanes
    if (G > 1) a = a + dpp_d<DPP_QUAD_XOR1>(a);
    if (G > 2) a = a + dpp_d<DPP_QUAD_XOR2>(a);
    if (G > 4) a = a + dpp_d<DPP_ROW_HALF_MIRROR>(a);   // quads are uniform after the first two stages: partner = xor 4
    if (G > 8) a = a + dpp_d<DPP_ROW_MIRROR>(a);        // likewise xor 8
    if (G > 16) a = xor16_sum(a);                       // (v_permlane16_swap / v_permlane32_swap: same operands as the shuffles, no LDS crossbar)
    if (G > 32) a = xor32_sum(a);
    return a;
}
__device__ __forceinline__ double group_scan_incl(double a, int g, int G) {   // Kogge-Stone inclusive scan over G lanes
    if (G > 16) {                      // groups spanning several DPP rows: row_shr cannot cross a row boundary
        for (int off = 1; off < G; off <<= 1) {
            const double t = __shfl_up(a, off, WAVE);
            if (g >= off) a = a + t;
        }
        return a;
    }
    if (G > 1) { const double t = dpp_d<DPP_ROW_SHR1>(a); if (g >= 1) a = a + t; }
    if (G > 2) { const double t = dpp_d<DPP_ROW_SHR2>(a); if (g >= 2) a = a + t; }
    if (G > 4) { const double t = dpp_d<DPP_ROW_SHR4>(a); if (g >= 4) a = a + t; }
    if (G > 8) { const double t = dpp_d<DPP_ROW_SHR8>(a); if (g >= 8) a = a + t; }
    return a;
}

// One 8-byte word of the hand-off state, stored write-through (global_store_dwordx2 ... sc1): visible to the other XCDs' L2s
// once the storing wave's s_waitcnt vmcnt(0) has passed, with no release fence (cdna_hip_programming.md Guideline 16, R1).
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ void publish(double *p, double x) {
    __hip_atomic_store((gu64 *)p, (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool B> struct BoolC { static constexpr bool value = B; };      // (a compile-time flag as a value: hiprtc brings no <type_traits>)
struct TimePt { double t, s, c, sp, spe; };   // grid time (+ sin/cos for the Boomerang rotation; SpeedUpZigZag: s, c = the flow's
                                              // scalar v1 X1(t) and its t-derivative, sp, spe = speed(x_t) and its t-derivative)

// flow of one coordinate: ZigZagSamplers.jl:80 / BouncyParticleSamplers.jl:32 (x + v t, v); BoomerangSamplers.jl:31 (rotation)
template <bool ROT> __device__ __forceinline__ void flow_coord(double x, double v, const TimePt &tp, Dual &xt, Dual &vt) {
    if constexpr (ROT) {
        xt = {x * tp.c + v * tp.s, x * (-tp.s) + v * tp.c};
        vt = {(-x) * tp.s + v * tp.c, (-x) * tp.c + v * (-tp.s)};
    } else {
        xt = {fma(v, tp.t, x), v};                           // fused: GROUP order of the checker
        vt = {v, 0.0};
    }
}
template <bool ROT> __device__ __forceinline__ void flow_coord(double x, double v, const TimePt &tp, double &xt, double &vt) {
    if constexpr (ROT) { xt = x * tp.c + v * tp.s; vt = (-x) * tp.s + v * tp.c; }
    else { xt = fma(v, tp.t, x); vt = v; }
}

// A chain's position at one time point as an indexable vector, for potentials given as source (user_jit.hip):
// w[j] = coordinate j of the flow at w.tp, read from the chain's LDS tiles (any lane of the group may read any slot).
template <class T> struct UserView {
    const double *xs, *vs;
    int base, logG, d;
    bool rot;                          // Boomerang rotation instead of the linear flow
    TimePt tp;
    __device__ __forceinline__ int size() const { return d; }
    __device__ __forceinline__ T operator[](int j) const {
        const int o = (j >> logG) * WAVE + base + (j & ((1 << logG) - 1));
        T xt, vt;
        if (rot) flow_coord<true>(xs[o], vs[o], tp, xt, vt); else flow_coord<false>(xs[o], vs[o], tp, xt, vt);
        return xt;
    }
};
// the same vector seeded for the partial with respect to coordinate i (one forward-mode pass of ForwardDiff.gradient)
template <class T> struct SeededView {
    const UserView<T> &w;
    int i;
    __device__ __forceinline__ int size() const { return w.d; }
    __device__ __forceinline__ Fwd<T> operator[](int j) const { return {w[j], lift<T>(j == i ? 1.0 : 0.0)}; }
};

// ---------------------------------------------------------------- targets in the group layout
// grad(i, x_i, shared): coordinate i of grad-U at a (dual or plain) position; `shared` carries what a
// non-separable potential needs from other coordinates of the same chain at the same time point.
struct NoShared {};
template <int TARGET> struct GTarget;

template <> struct GTarget<T_GAUSS_ISO> {
    static constexpr bool kShared = false, kTile = false, kUser = false;
    static constexpr bool kLinear = true;      // grad-U is affine in x: along a linear flow the rate's slope is the same at every grid time
    static constexpr int kAffineFrom = 0;      // coordinates i >= kAffineFrom have d grad_i / dx_j = const * delta_ij
    static constexpr bool kMonotone = true;    // ... with const >= 0: the Zig-Zag rate (x_i + v_i t) v_i of such a coordinate never decreases in t
    template <class T> using Sh = NoShared;
    __device__ __forceinline__ void init(const double *, int) {}
    template <class T> __device__ __forceinline__ NoShared shared(T, T) const { return {}; }
    template <class T> __device__ __forceinline__ T grad(int, T x, const NoShared &) const { return x; }
};
template <> struct GTarget<T_GAUSS_DIAG> {
    static constexpr bool kShared = false, kTile = false, kUser = false;
    static constexpr bool kLinear = true;
    static constexpr int kAffineFrom = 0;
    static constexpr bool kMonotone = false;   // a precision may be negative ...
    bool nonneg = false;                       // ... but the host has looked (SkelParams::tflags bit 0): with every precision >= 0 the Zig-Zag rate
                                               // (prec_i (x_i + v_i t - mu_i)) v_i never decreases in t -- rounding is monotone through the fma, the subtraction and
                                               // the two products -- so max(v0, v1, 0) IS max(v1, 0), bit for bit, and the left end of an affine cell need not be
                                               // evaluated (round 5; the checker keeps computing max(v0, v1, 0): the same double)
    template <class T> using Sh = NoShared;
    const double *mu, *prec;            // LDS copies
    __device__ __forceinline__ void init(const double *lds_params, int d) { mu = lds_params; prec = lds_params + d; }
    template <class T> __device__ __forceinline__ NoShared shared(T, T) const { return {}; }
    __device__ __forceinline__ double grad(int i, double x, const NoShared &) const { return prec[i] * (x - mu[i]); }
    __device__ __forceinline__ Dual grad(int i, Dual x, const NoShared &) const {
        const double p = prec[i];
        return {p * (x.v - mu[i]), p * x.e};
    }
};
template <class T> struct BananaSh { T g0, r; };
template <> struct GTarget<T_BANANA> {          // test/test_config.jl:33-36
    static constexpr bool kShared = true, kTile = false, kUser = false;
    static constexpr bool kLinear = false;
    static constexpr int kAffineFrom = 2;      // only x_1, x_2 are twisted: every other coordinate is a unit Gaussian
    static constexpr bool kMonotone = true;
    template <class T> using Sh = BananaSh<T>;
    int d;
    __device__ __forceinline__ void init(const double *, int dim) { d = dim; }
    template <class T> __device__ __forceinline__ BananaSh<T> shared(T x0, T x1) const {
        const T sq = x0 * x0;
        const T m = sq - 1.0;
        const T r = x1 - m;
        const T a = 2.0 * x0;
        const T pr = a * r;
        return {x0 - pr, r};
    }
    template <class T> __device__ __forceinline__ T grad(int i, T x, const BananaSh<T> &sh) const {
        if (d < 2) return x;
        return i == 0 ? sh.g0 : (i == 1 ? sh.r : x);
    }
};

#ifdef PDMP_USER_KIND
// Potential given as source (pdmp_create_from_source): PDMP_USER_KIND 0 = scalar potential pdmp_user::U(x, d), differentiated
// here by forward mode (ADBackend.jl:95-109); 1 = coordinate i of the gradient, pdmp_user::grad_i(i, x, d) (ADBackend.jl:55-66).
template <> struct GTarget<T_USER> {
    static constexpr bool kShared = true, kTile = false, kUser = true;
    static constexpr bool kLinear = false;
    static constexpr int kAffineFrom = 1 << 30;
    static constexpr bool kMonotone = false;
    template <class T> using Sh = UserView<T>;
    __device__ __forceinline__ void init(const double *, int) {}
    template <class T> __device__ __forceinline__ T grad(int i, T, const UserView<T> &w) const {
#if PDMP_USER_KIND == 0
        const SeededView<T> sx{w, i};
        const Fwd<T> u = pdmp_user::U(sx, w.d);
        return u.e;
#else
        return pdmp_user::grad_i(i, w, w.d);
#endif
    }
};
#endif

// KC = 0: K = ceil(d/G) slots per lane at run time, x and v in LDS tiles.  KC > 0 (register form): up to KC slots, x and v in
// REGISTERS, every slot loop fully unrolled with a scalar skip of the slots >= K -- no ds_read latency in the loops, no slot
// addressing.  Same operations in the same order: the two forms agree bit for bit.
#define PDMP_SLOTS(s) _Pragma("unroll kUnroll") for (int s = 0; s < nk(); ++s) if (!kReg || s < kmin || s < K)
// TAPE: the random source is a set of pre-recorded tapes instead of the Philox stream (pdmp_set_tape) -- separate
// instantiations (inst_tape_*.hip), so the kernels that serve the Philox stream are compiled from exactly the code they had before.
// LEAN: the throughput instantiation of a FULL 13-slot register kernel (round 3, as the lane kernels' DIAG = false): a launch that
// records no diagnostic array and uses none of the time-horizon features (t_stop, resume, chain map, column counts) runs a kernel that
// does not contain them -- their pointers, counters and the five-slot error ring cost scalar registers (spilled to VGPR lanes and
// read back with v_readlane in the event loop) and instructions in kernels that are bound by VALU issue.
// RTGRID (round 5, LDS form): grid_size is a run-time number (2 .. 64) instead of the constructors' default 10 -- a separate instantiation: with the
// two kernels differ in registers (GroupOcc: Zig-Zag on the banana with finite-difference slopes holds three waves per SIMD only with the grid a
// constant: launch_group_impl keeps a constant-grid kernel exactly there); measured otherwise equal (StickyZigZag 35.5 / 35.6 ms, SpeedUpZigZag 65.8 / 66.2).
template <int SAMPLER, int TARGET, bool FD, int KC = 0, int GC = 0, bool TAPE = false, bool LEAN = false, bool RTGRID = false>      // GC: lanes per chain when it is a compile-time constant (register form)
struct GroupChain {
    using Tgt = GTarget<TARGET>;
    static constexpr bool kReg = KC > 0;
    static constexpr bool kZz = SAMPLER == S_ZIGZAG || SAMPLER == S_SUZZ || SAMPLER == S_STICKY;   // Zig-Zag family: per-coordinate rates, flips
    static constexpr bool kSticky = SAMPLER == S_STICKY;
    static constexpr int kUnroll = KC > 0 ? KC : 2;   // slot loops: fully unrolled in the register form, by two in the LDS form (RHMC +10 %, the others +0-3 %)
    static_assert(!kReg || (!Tgt::kTile && !Tgt::kUser && KC >= 2), "register form: separable / banana targets only");
    const SkelParams &p;
    const int G, K, lane, g, base;       // g = lane within the group, base = first lane of the group
    // register form: the host picks the smallest G with ceil(d/G) <= KREG, so K >= (KREG+1)/2 whenever G >= 2 -- those slots
    // need no skip (G is a compile-time constant there) and their dependency chains interleave
    const int kmin;
    double *xs, *vs;                     // this wave's LDS tiles: slot s of lane l at [s*64 + l]
    mutable double xr[kReg ? KC : 1], vr[kReg ? KC : 1];   // register form: slot s of this lane
    double *fe_t[4] = {nullptr, nullptr, nullptr, nullptr};   // ForwardECMC: LDS tiles for n, v_orth, e1 / g1, e2 / g2 / the proposal
    __device__ __forceinline__ int nk() const { if constexpr (kReg) return KC; else return K; }
    // The register form serves the constructors' DEFAULT bound flags only -- grid_size = 10 and, for Zig-Zag, the vectorized signed
    // bound (the host routes every other combination to the LDS form): the unsigned / non-vectorized / Brent code is not compiled into
    // these kernels at all, which is what keeps them inside 2 waves per SIMD without spills.
    __device__ __forceinline__ bool brent() const { if constexpr (kReg) return false; else return p.n_grid == 0; }
    // grid_size of the bounds (UpperBound.jl:92,203: n_grid is a free parameter; the reference's tests use 5, 8, 10, 20): a run-time number
    // in the RTGRID instantiations of the LDS form (2 .. 64), the constructors' default -- a constant -- everywhere else
    __device__ __forceinline__ int ng() const { if constexpr (RTGRID && !kReg) return p.n_grid; else return GNG; }
    __device__ __forceinline__ bool vectorized() const { if constexpr (kReg) return true; else return p.vectorized != 0; }
    __device__ __forceinline__ bool zz_signed() const { if constexpr (kReg) return true; else return p.signed_bound != 0; }
    // coordinate i = s*G + g exists (i < d): only the last slot K-1 can be partly padded (register form: a scalar test for the others)
    __device__ __forceinline__ bool in_dim(int s, int i) const { if constexpr (kReg) return s < K - 1 || i < p.d; else return i < p.d; }
    Tgt tgt;
    ChainRng rng;
    int32_t guard = 0;
    // ---- random stream, prefetched per pass of the state machine.  A pass draws at most three variates of a chain -- the bound's
    // exponential, the acceptance uniform, the jump's uniform: stream positions n, n+1, n+2 = Philox blocks n>>1 and (n>>1)+1.
    // Drawn one by one (ChainRng::uniform) each draw runs the ten Philox rounds whenever ANY of the wave's chains starts a new block,
    // i.e. three times per pass although a chain consumes 1.5 blocks.  With G >= 2 lanes per chain the even lanes of a chain compute
    // block n>>1 and the odd lanes block (n>>1)+1 in ONE execution at the top of the pass; a draw then picks its half and fetches it
    // from the even or odd neighbour by quad_perm DPP.  Same variates at the same stream positions (the stream is counter-based).
    // Register form only (G is a compile-time constant there; the LDS-form and the runtime-compiled kernels draw one variate at a time:
    // a runtime-compiled Boomerang kernel -- 256 VGPRs + 32 AGPRs at one wave per SIMD -- misbehaved with the six more live registers).
#ifdef PDMP_NO_PREFETCH          // A/B builds (scripts/ab.sh)
    static constexpr bool kPrefetch = false;
#else
    static constexpr bool kPrefetch = kReg && !TAPE && (SAMPLER == S_ZIGZAG || SAMPLER == S_BPS || SAMPLER == S_BOOMERANG);
#endif
    // ---- tape source: one tape per kind of variate (Exp(1), U(0,1), N(0,1)), each in consumption order, chain-major by GLOBAL
    // chain id -- the format of julia/dump_tape.jl, of the wave-per-chain kernel (skeleton_kernel.hpp: r_exp / r_uni / r_normal) and
    // of the checker.  An exhausted tape yields 1.0 / 0.5 / 0.0, flags the launch (status[3]) and
    // stops the chain.  A taped launch runs in ONE segment (the hand-off state carries one stream position, a tape has three).
    const double *tp_e = nullptr, *tp_u = nullptr, *tp_n = nullptr;
    int32_t tp_ie = 0, tp_iu = 0, tp_in = 0;
    bool tp_over = false;
    __device__ __forceinline__ void tape_open(int64_t c) {
        if constexpr (TAPE) {
            const size_t gc = (size_t)(p.chain0 + c);
            tp_e = p.tape_exp + gc * (size_t)p.tape_n_exp;
            tp_u = p.tape_uni + gc * (size_t)p.tape_n_uni;
            tp_n = p.tape_nrm + gc * (size_t)p.tape_n_nrm;
            tp_ie = 0; tp_iu = 0; tp_in = 0; tp_over = false;
        }
    }
    // the chain's next uniform / exponential outside the prefetched stream positions (and every draw of a taped launch)
    __device__ __forceinline__ double r_uniform() {
        if constexpr (TAPE) { if (tp_iu < p.tape_n_uni) return tp_u[tp_iu++]; tp_over = true; return 0.5; }
        else return rng.uniform();
    }
    __device__ __forceinline__ double r_exponential() {
        if constexpr (TAPE) { if (tp_ie < p.tape_n_exp) return tp_e[tp_ie++]; tp_over = true; return 1.0; }
        else return rng.exponential();
    }
    // coordinate i of the chain's next vector of normals (call r_normals_done(count) once afterwards)
    __device__ __forceinline__ double r_normal_coord(int i) const {
        if constexpr (TAPE) return (tp_in + i < p.tape_n_nrm) ? tp_n[tp_in + i] : 0.0;
        else return rng.normal_coord(i);
    }
    __device__ __forceinline__ void r_normal_pair(int P, double &z0, double &z1) const {     // normals 2P, 2P + 1 of that vector
        if constexpr (TAPE) { z0 = r_normal_coord(2 * P); z1 = r_normal_coord(2 * P + 1); }
        else rng.normal_pair(P, z0, z1);
    }
    __device__ __forceinline__ void r_normals_done(int n) {
        if constexpr (TAPE) { if (tp_in + n > p.tape_n_nrm) tp_over = true; tp_in += n; }
        else rng.normals_done(n);
    }
    // The normals of ANOTHER chain of this wave (the one whose first lane is L), for the wave-cooperative refreshes: its stream
    // counters (Philox) or its tape position, fetched from lane L.
    struct NormalSrc { ChainRng r; const double *tn; int left; };
    template <bool UNIFORM_L> __device__ __forceinline__ NormalSrc normal_src(int L) const {
        NormalSrc o;
        o.r = rng; o.tn = nullptr; o.left = 0;                                     // seed words are the same in every lane
        if constexpr (TAPE) {
            const unsigned long long a = (unsigned long long)(uintptr_t)(tp_n + tp_in);
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)a, L, WAVE), hi = (uint32_t)__shfl((int)(uint32_t)(a >> 32), L, WAVE);
            o.tn = (const double *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
            o.left = __shfl((int)(p.tape_n_nrm - tp_in), L, WAVE);
        } else if constexpr (UNIFORM_L) {                                          // L wave-uniform: the counters come as scalars
            o.r.c2 = (uint32_t)__builtin_amdgcn_readlane((int)rng.c2, L);
            o.r.c3 = (uint32_t)__builtin_amdgcn_readlane((int)rng.c3, L);
            o.r.n = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rng.n >> 32), L) << 32) |
                    (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rng.n, L);
        } else {
            o.r.c2 = (uint32_t)__shfl((int)rng.c2, L, WAVE);
            o.r.c3 = (uint32_t)__shfl((int)rng.c3, L, WAVE);
            o.r.n = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(rng.n >> 32), L, WAVE) << 32) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)rng.n, L, WAVE);
        }
        return o;
    }
    __device__ __forceinline__ static void src_pair(const NormalSrc &o, int P, double &z0, double &z1) {
        if constexpr (TAPE) { z0 = (2 * P < o.left) ? o.tn[2 * P] : 0.0; z1 = (2 * P + 1 < o.left) ? o.tn[2 * P + 1] : 0.0; }
        else o.r.normal_pair(P, z0, z1);
    }
    uint32_t pf_w0 = 0, pf_w1 = 0, pf_w2 = 0, pf_w3 = 0;
    uint64_t pf_n = 0;
    __device__ __forceinline__ void rng_prefetch() {
        if constexpr (kPrefetch) {
            if (G >= 2) {
                pf_n = rng.n;
                uint32_t w[4];
                rng.block_at((pf_n >> 1) + (uint64_t)(g & 1), w);
                pf_w0 = w[0]; pf_w1 = w[1]; pf_w2 = w[2]; pf_w3 = w[3];
            }
        }
    }
    __device__ __forceinline__ double draw_uniform() {
        if constexpr (TAPE) return r_uniform();
        if constexpr (kPrefetch) {
            if (G >= 2) {
                const uint64_t idx = rng.n;
                const uint32_t sel = (uint32_t)((idx >> 1) - (pf_n >> 1));           // 0: the even lanes' block, 1: the odd lanes'
                const bool half = (idx & 1ull) != 0ull;
                const int lo_own = (int)(half ? pf_w2 : pf_w0), hi_own = (int)(half ? pf_w3 : pf_w1);
                const int lo_e = __builtin_amdgcn_update_dpp(0, lo_own, 0xA0, 0xF, 0xF, true);   // quad_perm [0,0,2,2]: the pair's even lane
                const int hi_e = __builtin_amdgcn_update_dpp(0, hi_own, 0xA0, 0xF, 0xF, true);
                const int lo_o = __builtin_amdgcn_update_dpp(0, lo_own, 0xF5, 0xF, 0xF, true);   // quad_perm [1,1,3,3]: the odd lane
                const int hi_o = __builtin_amdgcn_update_dpp(0, hi_own, 0xF5, 0xF, 0xF, true);
                uint32_t lo = (uint32_t)(sel ? lo_o : lo_e), hi = (uint32_t)(sel ? hi_o : hi_e);
                if (__ballot(sel > 1u) != 0ull) {                                    // a fourth draw in one pass (no sampler of this family
                    uint32_t w[4];                                                   // makes one): its block directly
                    rng.block_at(idx >> 1, w);
                    if (sel > 1u) { lo = half ? w[2] : w[0]; hi = half ? w[3] : w[1]; }
                }
                ++rng.n;
                return bits_to_u(lo, hi);
            }
        }
        return rng.uniform();
    }
    __device__ __forceinline__ double draw_exponential() {
        if constexpr (TAPE) return r_exponential();
        else return -det_log(draw_uniform());
    }

    const int cell_budget;               // cells of a lazily built bound per pass of the state machine (bound_advance_vect)
    __device__ __forceinline__ GroupChain(const GroupParams &gp, int l, double *x_tile, double *v_tile)
        : p(gp.s), G(gp.G), K(gp.K), lane(l), g(l & (gp.G - 1)), base(l & ~(gp.G - 1)), kmin(gp.G >= 2 ? (KC + 1) / 2 : 1), xs(x_tile), vs(v_tile),
          cell_budget(gp.cell_budget) {}

    __device__ __forceinline__ double &X(int s) const { if constexpr (kReg) return xr[s]; else return xs[s * WAVE + lane]; }
    __device__ __forceinline__ double &V(int s) const { if constexpr (kReg) return vr[s]; else return vs[s * WAVE + lane]; }
    // ---- StickyZigZag (src/StickySamplingLoop.jl): bit s of amask = the coordinate of slot s is active (not stuck at 0);
    // flows, rates and bounds use v .* is_active (PDMPState.v_active, SamplingLoopInplace.jl:13-25), the velocity jump the full v
    uint32_t amask = 0xffffffffu;
    const double *kap = nullptr;         // kappa[d] in LDS (thawing rates)
    double tt = 0.0, thaw_rate = 0.0;    // time to thaw and the rate it was drawn with (StickySamplingLoop.jl:40-46)
    __device__ __forceinline__ bool active(int s) const { if constexpr (kSticky) return (amask >> s) & 1u; else return true; }
    __device__ __forceinline__ double VU(int s) const { if constexpr (kSticky) return active(s) ? V(s) : 0.0; else return V(s); }
    __device__ __forceinline__ void coord(int i, double &x, double &v) const {   // any coordinate of my chain (i = 0, 1 in the register form)
        const int s = (G == WAVE) ? (i >> 6) : i / G, l = base + (i - s * G);
        if constexpr (kSticky) {
            const uint32_t m = (uint32_t)__shfl((int)amask, l, WAVE);
            double xv, vv;
            if constexpr (kReg) { xv = __shfl((s == 0) ? xr[0] : xr[1], l, WAVE); vv = __shfl((s == 0) ? vr[0] : vr[1], l, WAVE); }   // (round 4: the register form has no x / v tiles)
            else { xv = xs[s * WAVE + l]; vv = vs[s * WAVE + l]; }
            x = xv;
            v = ((m >> s) & 1u) ? vv : 0.0;
            return;
        }
        if constexpr (kReg) {
            if (i < 2 && (G == 4 || G == 2)) {          // lane i of every group by quad_perm: register to register, no LDS crossbar
                if (G == 4) {
                    if (i == 0) { x = dpp_d<0x00>(xr[0]); v = dpp_d<0x00>(vr[0]); }      // quad_perm [0,0,0,0]
                    else { x = dpp_d<0x55>(xr[0]); v = dpp_d<0x55>(vr[0]); }             // quad_perm [1,1,1,1]
                } else {
                    if (i == 0) { x = dpp_d<0xA0>(xr[0]); v = dpp_d<0xA0>(vr[0]); }      // quad_perm [0,0,2,2]
                    else { x = dpp_d<0xF5>(xr[0]); v = dpp_d<0xF5>(vr[0]); }             // quad_perm [1,1,3,3]
                }
                return;
            }
            x = __shfl((s == 0) ? xr[0] : xr[1], l, WAVE);
            v = __shfl((s == 0) ? vr[0] : vr[1], l, WAVE);
        } else {
            x = xs[s * WAVE + l];
            v = vs[s * WAVE + l];
        }
    }
    // ---- SpeedUpZigZag (SpeedUpZigZagSamplers.jl:71-83).  The closed-form flow is x_t = y + (v1 X1(t)) v with y = x - (v1 x1) v:
    // per (x, v) a context of chain-level scalars (two dot products), per time point the scalar k(t) = v1 X1(t) as a dual in t
    // and, for grad-U_effective = speed grad-U - x / speed, speed(x_t) = sqrt(1 + x_t.x_t) as a dual (one more dot product).
    // Every operation in the order written in the reference; derivatives by the forward-mode rules, no fma.
    mutable double sz_a = 0.0, sz_A = 0.0, sz_cd = 0.0, sz_base = 0.0, sz_sd = 0.0, sz_v1 = 0.0;
    mutable bool sz_ok = false;
    __device__ __forceinline__ void suzz_ctx() const {
        if (__ballot(!sz_ok) == 0ull) return;
        double x1, v1;
        coord(0, x1, v1);
        const double av = v1 * x1;
        double yv = 0.0, yy = 0.0;
        PDMP_SLOTS(s) { const double y = X(s) - av * V(s); yv += y * V(s); yy += y * y; }
        yv = group_sum(yv, G);
        yy = group_sum(yy, G);
        const double dd = (double)p.d;
        const double c = v1 * yv;
        sz_a = av;
        sz_A = (1.0 + yy) / dd - (c * c) / (dd * dd);
        sz_cd = c / dd;
        const double Y0 = x1 + sz_cd;
        sz_base = Y0 + __dsqrt_rn(Y0 * Y0 + sz_A);
        sz_sd = __dsqrt_rn(dd) * v1;
        sz_v1 = v1;
        sz_ok = true;
    }
    __device__ __forceinline__ void suzz_speed(TimePt &tp, bool at_state) const {     // speed(x_t) (at_state: of the stored x itself)
        double sv = 0.0, se = 0.0;
        PDMP_SLOTS(s) {
            Dual xt, vt;
            if (at_state) xt = {X(s), 0.0}; else flow(X(s), V(s), tp, xt, vt);
            sv += xt.v * xt.v;
            se += xt.e * xt.v + xt.v * xt.e;
        }
        const double s2v = 1.0 + group_sum(sv, G), s2e = group_sum(se, G);
        tp.sp = __dsqrt_rn(s2v);
        tp.spe = s2e / (2.0 * tp.sp);
    }
    __device__ __forceinline__ void time_pt(double t, TimePt &tp) const {
        tp.t = t;
        tp.sp = 1.0; tp.spe = 0.0;
        if constexpr (SAMPLER == S_BOOMERANG) det_sincos(t, tp.s, tp.c);
        else if constexpr (SAMPLER == S_SUZZ) {
            suzz_ctx();
            const double ev = det_exp(sz_sd * t), ee = sz_sd * ev;
            const double bv = sz_base * ev, be = sz_base * ee;
            const double b2v = bv * bv, b2e = be * bv + bv * be;
            const double nv = b2v - sz_A, dv = 2.0 * bv, de = 2.0 * be;
            const double qv = nv / dv, qe = (b2e - qv * de) / dv;
            tp.s = sz_v1 * (qv - sz_cd);
            tp.c = sz_v1 * qe;
            suzz_speed(tp, false);
        } else { tp.s = 0.0; tp.c = 1.0; }
    }
    // flow of one coordinate: ZigZagSamplers.jl:80 / BouncyParticleSamplers.jl:32 / BoomerangSamplers.jl:31 / SpeedUpZigZagSamplers.jl:71-79
    __device__ __forceinline__ void flow(double x, double v, const TimePt &tp, Dual &xt, Dual &vt) const {
        if constexpr (SAMPLER == S_SUZZ) { const double y = x - sz_a * v; xt = {y + tp.s * v, tp.c * v}; vt = {v, 0.0}; }
        else flow_coord<SAMPLER == S_BOOMERANG>(x, v, tp, xt, vt);
    }
    __device__ __forceinline__ void flow(double x, double v, const TimePt &tp, double &xt, double &vt) const {
        if constexpr (SAMPLER == S_SUZZ) { const double y = x - sz_a * v; xt = y + tp.s * v; vt = v; }
        else flow_coord<SAMPLER == S_BOOMERANG>(x, v, tp, xt, vt);
    }
    // grad-U_effective(x_t) = speed grad-U - x_t / speed (SpeedUpZigZagSamplers.jl:81-83); identity for the other samplers
    __device__ __forceinline__ Dual eff(Dual g, Dual xt, const TimePt &tp) const {
        if constexpr (SAMPLER == S_SUZZ) {
            const double mv = tp.sp * g.v, me = tp.spe * g.v + tp.sp * g.e;
            const double qv = xt.v / tp.sp, qe = (xt.e - qv * tp.spe) / tp.sp;
            return {mv - qv, me - qe};
        } else return g;
    }
    __device__ __forceinline__ double eff(double g, double xt, const TimePt &tp) const {
        if constexpr (SAMPLER == S_SUZZ) return tp.sp * g - xt / tp.sp; else return g;
    }
    // out[i] = sum_j P[i,j] in[j], j left to right per row (row i owned by lane i % 64, slot i / 64).
    // Eight rows per lane are accumulated together so that eight independent 512-byte column pieces are in
    // flight per j (a single running sum per lane would be bound by the load latency); padded rows read row 0
    // and are zeroed at the end.
    __device__ __forceinline__ void matvec(const double *P, const double *in, double *out) const {
        constexpr int R = 8;
        const int d = p.d;
        for (int s0 = 0; s0 < K; s0 += R) {
            double acc[R];
            const double *row[R];
            bool ok[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int i = (s0 + q) * WAVE + lane;
                ok[q] = (s0 + q < K) && (i < d);
                row[q] = P + (ok[q] ? i : 0);
                acc[q] = 0.0;
            }
#pragma unroll 4
            for (int j = 0; j < d; ++j) {
                const double xj = in[j];
                const size_t off = (size_t)j * d;
#pragma unroll
                for (int q = 0; q < R; ++q) acc[q] += row[q][off] * xj;
            }
#pragma unroll
            for (int q = 0; q < R; ++q)
                if (s0 + q < K) out[(s0 + q) * WAVE + lane] = ok[q] ? acc[q] : 0.0;
        }
    }
    template <class T>
    __device__ __forceinline__ typename Tgt::template Sh<T> shared_at(const TimePt &tp, int which = 0) {
        if constexpr (Tgt::kTile) {
            PDMP_SLOTS(s) {
                double xt, vt;
                flow(X(s), VU(s), tp, xt, vt);
                tgt.tx[s * WAVE + lane] = xt;
            }
            matvec(tgt.P, tgt.tx, tgt.tg[which]);
            if constexpr (SAMPLER == S_BOOMERANG) {                 // the Boomerang's flow rotates the velocity: d/dt (P x_t) = P v_t, a product per time point
                PDMP_SLOTS(s) {                                     // (round 4: with the kept P v the slopes were those of t = 0 at every grid time, and the
                    double xt, vt;                                  //  chains whose bound takes a tangent intersection left the checker by ~1e-4)
                    flow(X(s), VU(s), tp, xt, vt);
                    tgt.tx[s * WAVE + lane] = vt;
                }
                matvec(tgt.P, tgt.tx, tgt.tb);
                tgt.b_valid = false;
            } else
            if (!tgt.b_valid) { matvec(tgt.P, vs, tgt.tb); tgt.b_valid = true; }   // G = 64: the v tile is the vector v
            return {tgt.tg[which], tgt.tb};
        } else if constexpr (Tgt::kUser) {
            return UserView<T>{xs, vs, base, __builtin_ctz(G), p.d, SAMPLER == S_BOOMERANG, tp};
        } else if constexpr (Tgt::kShared) {
            if (p.d >= 2) {
                double x0, v0, x1, v1;
                coord(0, x0, v0);
                coord(1, x1, v1);
                T a, b, va, vb;
                flow(x0, v0, tp, a, va);
                flow(x1, v1, tp, b, vb);
                return tgt.template shared<T>(a, b);
            }
            return typename Tgt::template Sh<T>{};
        } else {
            return {};
        }
    }
    __device__ __forceinline__ double sdot(const Dual &gd, const Dual &vt) const {   // d/dt of g*v
        if constexpr (SAMPLER == S_BOOMERANG) return gd.e * vt.v + gd.v * vt.e;
        else return gd.e * vt.v;
    }

    // ---- "Affine tail" (arithmetic contract, GROUP order; mirrored by the checker's lr_* functions): BPS with forward-mode slopes on a
    // potential whose gradient components i >= aff are affine in their own coordinate (Gaussians: aff = 0; banana: aff = 2).  Along the
    // linear flow the bounded function g(x + v t).v (BouncyParticleSamplers.jl:39-47) is  head(t) + A + B t  with
    //   A = sum_{i >= aff} g_i(x) v_i,  B = sum_{i >= aff} (dg_i/dx_i v_i) v_i  -- ONE pass over the coordinates per state (x, v) --
    //   head(t) = sum_{i < aff} g_i(x_t) v_i on forward-mode numbers (the banana's two twisted coordinates, evaluated by every lane of
    //   the chain's group from broadcast copies: no cross-lane sum per grid time).
    // Every grid time of every bound and the proposal's rate then cost O(1) instead of a pass over d coordinates and two butterflies:
    // BPS on the banana went from 404 to ~100 VALU instructions per event.
    static constexpr bool kLowRank = (SAMPLER == S_BPS || SAMPLER == S_FECMC) && !FD && !Tgt::kTile && !Tgt::kUser;   // (ForwardECMC bounds BPS's rate, ForwardEventChainMonteCarlo.jl:19-43)
    double lr_A = 0.0, lr_B = 0.0, lr_x0 = 0.0, lr_v0 = 0.0, lr_x1 = 0.0, lr_v1 = 0.0;
    mutable bool lr_ok = false;                  // (lr_A, lr_B, head copies) belong to the current (x, v)
    __device__ __forceinline__ int lr_aff() const { return (Tgt::kAffineFrom > 0 && Tgt::kAffineFrom < (1 << 30) && p.d >= 2) ? Tgt::kAffineFrom : 0; }
    __device__ __forceinline__ void lr_prepare() {
        if (__ballot(!lr_ok) == 0ull) return;
        const int aff = lr_aff();
        double a = 0.0, b = 0.0;
        const typename Tgt::template Sh<Dual> nosh{};                      // (an affine component reads no other coordinate)
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            const bool tail = in_dim(s, i) && i >= aff;
            const Dual gd = tgt.grad(in_dim(s, i) ? i : 0, Dual{X(s), V(s)}, nosh);
            const double ta = gd.v * V(s), tb = gd.e * V(s);
            a += tail ? ta : 0.0;
            b += tail ? tb : 0.0;
        }
        a = group_sum(a, G);
        b = group_sum(b, G);
        double x0 = 0.0, v0 = 0.0, x1 = 0.0, v1 = 0.0;
        if (aff > 0) { coord(0, x0, v0); coord(1, x1, v1); }
        if (!lr_ok) { lr_A = a; lr_B = b; lr_x0 = x0; lr_v0 = v0; lr_x1 = x1; lr_v1 = v1; lr_ok = true; }
    }
    // value (and t-derivative) of g(x_t).v at time t from the prepared coefficients
    __device__ __forceinline__ void lr_eval(double t, double &fv, double &fe) const {
        fv = fma(lr_B, t, lr_A);
        fe = lr_B;
        if constexpr (Tgt::kShared && !Tgt::kTile && !Tgt::kUser) {        // a twisted head: the banana's coordinates 0 and 1
            if (lr_aff() > 0) {
                const Dual a{fma(lr_v0, t, lr_x0), lr_v0}, b{fma(lr_v1, t, lr_x1), lr_v1};
                const auto sh = tgt.template shared<Dual>(a, b);           // {g_0, g_1}
                const double hv = (0.0 + sh.g0.v * lr_v0) + sh.r.v * lr_v1, he = (0.0 + sh.g0.e * lr_v0) + sh.r.e * lr_v1;
                fv = hv + fv;
                fe = he + fe;
            }
        }
    }

    // ---- "Quadratic form" (arithmetic contract, GROUP order; mirrored by the checker's lrb_* functions): Boomerang on the isotropic
    // Gaussian.  Along the rotation (BoomerangSamplers.jl:31) the bounded function x_t . v_t of :38-46 is
    //   XV (c^2 - s^2) + (VV - XX) s c,   d/dt = XV (-4 s c) + (VV - XX) (c^2 - s^2),   XX = x.x, XV = x.v, VV = v.v of the state:
    // one pass over the coordinates per state instead of one per grid time and finite-difference point (30 per bound with the
    // constructor's default AD_backend = "FiniteDiff"); what remains per point is its sin / cos.
    static constexpr bool kQuad = SAMPLER == S_BOOMERANG && TARGET == T_GAUSS_ISO;
    double q_xx = 0.0, q_xv = 0.0, q_vv = 0.0;
    __device__ __forceinline__ void q_prepare() {
        if (__ballot(!lr_ok) == 0ull) return;
        double a = 0.0, b = 0.0, c = 0.0;
        PDMP_SLOTS(s) { const double x = X(s), v = V(s); a += x * x; b += x * v; c += v * v; }    // (padded coordinates hold zeros)
        a = group_sum(a, G); b = group_sum(b, G); c = group_sum(c, G);
        if (!lr_ok) { q_xx = a; q_xv = b; q_vv = c; lr_ok = true; }
    }
    __device__ __forceinline__ double q_raw(const TimePt &tp, double *de) const {
        const double cc = tp.c * tp.c, ss = tp.s * tp.s, sc = tp.s * tp.c, dq = q_vv - q_xx;
        if (de) *de = q_xv * (-4.0 * sc) + dq * (cc - ss);
        return q_xv * (cc - ss) + dq * sc;
    }
    __device__ __forceinline__ double q_bounded(const TimePt &tp, double *slp) const {     // signed_rate / rate incl. refresh_rate (Q2)
        double de;
        const double dv = q_raw(tp, &de);
        const bool sgn = p.signed_bound != 0;
        if (slp) *slp = sgn ? de : ((0.0 < dv) ? de : 0.0);
        return (sgn ? dv : fmax(0.0, dv)) + p.refresh;
    }

    // ---- tangent-intersection cell, UpperBound.jl:229-241 (absolute-time form, Q1) / :123-130
    // den == 0 (every Gaussian coordinate) gives num/0 = +-Inf or NaN, which the reference's
    // NaN->0 + clamp turns into h or 0: same bits as the division path, no v_div sequence.
    __device__ __forceinline__ static double cell_q(double num, double den, double h, bool fast) {
        if (fast) return (num > 0.0) ? h : 0.0;
        return fmin(fmax(num / den, 0.0), h);       // fmax drops NaN: the reference's replace(NaN => 0)
    }

    // ---- Bound + inversion, fused and LAZY.
    // next_event (UpperBound.jl:264-273) only reads cum_sum up to the first cell with cum_sum >= E and that
    // cell's box_max, so the cells of upper_bound_grid_vect / upper_bound_grid (:203-247 / :92-137) are built
    // in grid order and the loop stops as soon as every chain of the wave has found its cell.  Each cell is
    // computed with exactly the arithmetic of the full bound, so (tp, lambda_bar) are bit-identical to
    // building all n-1 cells; with the adaptive horizon several times the typical event time, two or
    // three cells out of nine are built on average.  E is drawn before the bound (the bound consumes
    // no random numbers, so the stream position is the reference's).
    // Padded coordinates (i >= d) hold x = v = 0: val = slope = 0, cell = max(0,0,0,0) = 0, no mask needed;
    // their parameter index is clamped so grad stays finite.
    template <bool SIGNED>
    __device__ __forceinline__ void eval_vect(int ip, double x, double v, const TimePt &tp,
                                              const typename Tgt::template Sh<Dual> &sh, double &val, double &slp) const {
        Dual xt, vt;                                                  // :208-210
        flow(x, v, tp, xt, vt);
        const Dual gd = eff(tgt.grad(ip, xt, sh), xt, tp);
        const double sv = gd.v * vt.v, sd = sdot(gd, vt);
        val = SIGNED ? sv : fmax(0.0, sv);
        slp = SIGNED ? sd : ((0.0 < sv) ? sd : 0.0);
    }
    // finite_difference_derivative (UpperBound.jl:50-76) sample points of one grid time, start = 0
    struct FdPt {
        TimePt c, m, pl;            // t, t-h (clamped to 0), t+h (clamped to the horizon)
        double den;                 // denominator of the difference quotient
        int kind;                   // 0 central, 1 forward (t-h == t), 2 backward (t+h == t), 3 degenerate
    };
    __device__ __forceinline__ void fd_point(double t, double horizon, FdPt &P) const {
        const double hh = 1.4901161193847656e-08 * fmax(1.0, fabs(t));               // sqrt(eps) * max(1, |t|), :58
        const double tm = fmax(0.0, t - hh), tq = (horizon < t + hh) ? horizon : t + hh;   // :60-61
        if constexpr (SAMPLER == S_BOOMERANG && TARGET == T_GAUSS_ISO) {
            // quadratic-form order: sin / cos of the two neighbours t + delta (|delta| <= 1.5e-8 max(1, t)) by the angle-addition
            // formulas from sin / cos of t -- sin(delta) = delta and cos(delta) = 1 - delta^2 / 2 to the last bit at that size; delta is
            // the exact difference of the rounded points (Sterbenz).  One range reduction per grid time instead of three.
            time_pt(t, P.c);
            auto shifted = [&](double tt, TimePt &o) {
                const double dl = tt - t, cd = 1.0 - 0.5 * (dl * dl);
                o = P.c;
                o.t = tt;
                o.s = P.c.s * cd + P.c.c * dl;
                o.c = P.c.c * cd - P.c.s * dl;
            };
            shifted(tm, P.m);
            shifted(tq, P.pl);
        } else { time_pt(t, P.c); time_pt(tm, P.m); time_pt(tq, P.pl); }
        P.kind = (tq == tm) ? 3 : ((tm == t) ? 1 : ((tq == t) ? 2 : 0));              // :64-75
        P.den = (P.kind == 1) ? tq - t : ((P.kind == 2) ? t - tm : tq - tm);
    }
    template <bool SIGNED>
    __device__ __forceinline__ double fval_vect(int ip, double x, double v, const TimePt &tp,
                                                const typename Tgt::template Sh<double> &sh) const {
        double xt, vt;
        flow(x, v, tp, xt, vt);
        const double sv = eff(tgt.grad(ip, xt, sh), xt, tp) * vt;
        return SIGNED ? sv : fmax(0.0, sv);
    }
    __device__ __forceinline__ static double fd_slope(const FdPt &P, double fc, double fm, double fp) {
        const double num = (P.kind == 1) ? fp - fc : ((P.kind == 2) ? fc - fm : fp - fm);
        return (P.kind == 3) ? fc - fc : num / P.den;
    }
    // (Round 5, measured and NOT shipped, on the vectorised Zig-Zag bound's 26 slopes per cell and lane at d = 50: (a) one refined reciprocal of
    // P.den per grid time and the three closing steps of the division per coordinate; (b) the two TIMES of each quotient selected once per grid
    // time instead of three values per coordinate.  Both are bit-identical to fd_slope and both made the LDS form SLOWER, 59.2 -> 68.7 ms at
    // config-2 size: the compiler already hoists the denominator's half of its division expansion and the selects' conditions out of the slot loop.)

    // Progress of this lane's chain through its lazily built bound: cells [0, b_next) are summed.  The cells are built at most
    // `gp.cell_budget` per pass of the state machine: a chain that needs more (the ~2 % that run to the horizon need all nine)
    // resumes in the next pass beside the fresh bounds of the other chains of the wave, instead of holding all of them for
    // six more cells.  Same cells, same order, same sums per chain: nothing changes in any chain's arithmetic.
    int b_next = 0;
    double b_acc = 0.0, b_cprev = 0.0, b_last = 0.0, b_cbefore = 0.0;    // b_cbefore: cum_sum before the last built cell
    bool b_recheck = false;
    typename Tgt::template Sh<Dual> b_shf{};
    __device__ __forceinline__ void bound_begin(double &tpo) {
        b_next = 0; b_acc = 0.0; b_cprev = 0.0; b_last = 0.0; b_cbefore = 0.0; b_recheck = false;
        tpo = __builtin_huge_val();
    }
    // After a rejection the reference inverts the SAME bound again with a larger exponential (if_reject!, :188-203; Q4): x, v
    // and the bound's horizon are unchanged, the cumulative sums are monotone and the new exponential is larger, so the new
    // cell is the one found last time or a later one -- the built cells are kept and the inversion resumes at the last one.
    __device__ __forceinline__ void bound_resume(double &tpo) { b_recheck = true; tpo = __builtin_huge_val(); }

    // returns true when this lane's bound is finished: its cell was found (tpo, lbo set) or all cells are built (lbo = box_max[end])
    template <bool SIGNED, bool FDM>
    __device__ __forceinline__ bool bound_advance_vect(double harg, double E, double &tpo, double &lbo, int budget) {
        const double h = harg / (double)(ng() - 1);
        bool found = false, found_now = false;
#ifdef PDMP_STAMP
        if (lane == 0 && p.stamps) atomicAdd(&p.stamps[7], 1ull);
#endif
        if (b_recheck) {                                                                // resumed after a rejection: the cell found last time first
            b_recheck = false;
            if (b_next > 0 && !(b_cprev < E)) {                                         // b_cprev = cum_sum including cell b_next - 1
                found = true;
                tpo = (double)(b_next - 1) * h + (E - b_cbefore) / (b_cprev - b_cbefore) * h;
                lbo = b_last;
            }
        }
        for (int pass = 0; pass < budget; ++pass) {
            const bool act = !found && b_next < ng() - 1;                                // the lanes of a chain agree
            if (__ballot(act) == 0ull) break;
#ifdef PDMP_STAMP
            if (lane == 0 && p.stamps) atomicAdd(&p.stamps[6], 1ull);
#endif
            if (act) {
                const int j = b_next;
                const double t0 = (double)j * h, t1 = (j == ng() - 2) ? harg : (double)(j + 1) * h;
                double bsum = 0.0;
                if constexpr (!FDM) {
                    TimePt p0, p1;
                    time_pt(t0, p0);
                    time_pt(t1, p1);
                    if constexpr (Tgt::kShared) {                                      // left endpoint: carried over from the previous cell
                        if (__ballot(j == 0) != 0ull) { const auto f = shared_at<Dual>(p0, 0); if (j == 0) b_shf = f; }
                    }
                    const auto sh0 = b_shf, sh1 = shared_at<Dual>(p1, (j + 1) & 1);
                    // "Affine cell" (arithmetic contract, GROUP order; mirrored by the checker): forward-mode Zig-Zag under the signed
                    // bound, slot of coordinates whose gradient component is affine in its own coordinate -- the rate is linear in t,
                    // its slope the same double at both ends of the cell, so the reference's tangent intersection (UpperBound.jl:229-237)
                    // divides by zero, clamps to h or 0 and yields v0 or v0 + g h = v1 up to rounding: the cell maximum (:239-241) is
                    // max(v0, v1, 0).  No slopes, no intersection; with a non-negative slope v1 >= v0 exactly (rounding is monotone),
                    // so only the right end is evaluated.  4 instead of 15 fp64 operations per coordinate and cell on the Gaussians.
                    // MONO: compile-time for the isotropic Gaussian / the banana; the diagonal Gaussian takes one of two copies of the slot
                    // loop by a wave-uniform branch on GTarget::nonneg (inside the loop the compiler turned the branch into a select and
                    // evaluated the left end regardless: measured, no gain).
                    auto slots = [&](auto mono_c) {
                        constexpr bool MONO = decltype(mono_c)::value;
                        PDMP_SLOTS(s) {
                            const int i = s * G + g;
                            const int ip = (in_dim(s, i)) ? i : 0;
                            const double x = X(s), v = VU(s);
                            if (SIGNED && SAMPLER == S_ZIGZAG && s * G >= Tgt::kAffineFrom) {      // wave-uniform; a constant in the register form
                                const typename Tgt::template Sh<double> nosh{};                        // (an affine component reads no other coordinate)
                                const double v1 = fval_vect<true>(ip, x, v, p1, nosh);
                                if constexpr (MONO) bsum += fmax(v1, 0.0);
                                else bsum += fmax(fmax(fval_vect<true>(ip, x, v, p0, nosh), v1), 0.0);
                            } else {
                                double v0, g0, v1, g1;
                                eval_vect<SIGNED>(ip, x, v, p0, sh0, v0, g0);
                                eval_vect<SIGNED>(ip, x, v, p1, sh1, v1, g1);
                                bsum += cell_vect(v0, g0, v1, g1, t0, t1, h);
                            }
                        }
                    };
                    if constexpr (TARGET == T_GAUSS_DIAG && SIGNED && SAMPLER == S_ZIGZAG) {         // (only where the affine cell exists: the other samplers keep ONE copy of the loop)
                        if (tgt.nonneg) slots(BoolC<true>{}); else slots(BoolC<false>{});
                    } else slots(BoolC<Tgt::kMonotone>{});
                    b_shf = sh1;
                } else {                                                               // AD_backend = "FiniteDiff", :223-224
                    FdPt P0, P1;
                    fd_point(t0, harg, P0);
                    fd_point(t1, harg, P1);
                    const auto a0 = shared_at<double>(P0.c), a1 = shared_at<double>(P0.m), a2 = shared_at<double>(P0.pl);
                    const auto b0 = shared_at<double>(P1.c), b1 = shared_at<double>(P1.m), b2 = shared_at<double>(P1.pl);
                    PDMP_SLOTS(s) {
                        const int i = s * G + g;
                        const int ip = (in_dim(s, i)) ? i : 0;
                        const double x = X(s), v = VU(s);
                        const double v0 = fval_vect<SIGNED>(ip, x, v, P0.c, a0);
                        const double g0 = fd_slope(P0, v0, fval_vect<SIGNED>(ip, x, v, P0.m, a1), fval_vect<SIGNED>(ip, x, v, P0.pl, a2));
                        const double v1 = fval_vect<SIGNED>(ip, x, v, P1.c, b0);
                        const double g1 = fd_slope(P1, v1, fval_vect<SIGNED>(ip, x, v, P1.m, b1), fval_vect<SIGNED>(ip, x, v, P1.pl, b2));
                        bsum += cell_vect(v0, g0, v1, g1, t0, t1, h);
                    }
                }
                const double bj = group_sum(bsum, G);
                b_acc += bj;                                                           // cumsum, :243-244 (engine order)
                const double cj = b_acc * h;
                if (!(cj < E)) {                                                       // searchsortedfirst, :265
                    found = true;
                    found_now = true;                                                  // the division of :269 follows the loop: once per pass
                    lbo = bj;                                                          // :270
                }
                b_cbefore = b_cprev;
                b_cprev = cj;
                b_last = bj;
                b_next = j + 1;
            }
        }
        // next_event, :269, for the chains that found their cell in this pass: cell b_next - 1, cum_sum b_cbefore before and b_cprev after it
        if (found_now) tpo = (double)(b_next - 1) * h + (E - b_cbefore) / (b_cprev - b_cbefore) * h;
        const bool fin = found || b_next >= ng() - 1;
        if (fin && !found) lbo = b_last;                                               // box_max[end], :270
        return fin;
    }
    // one tangent-intersection cell of one coordinate, UpperBound.jl:229-241 (absolute-time form, Q1)
    __device__ __forceinline__ static double cell_vect(double v0, double g0, double v1, double g1, double t0, double t1, double h) {
        const double num = fma(-g0, t0, fma(g1, t1, v0 - v1));                         // :229 ((v0-v1) + g1 t1) - g0 t0, fused
        const double den = g1 - g0;
        const bool fast = __ballot(den != 0.0) == 0ull;                                // wave-uniform: no lane divides (cell_q)
        const double q = cell_q(num, den, h, fast);                                    // :234-235
        const double inter = fma(g0, q, v0);                                           // :237
        return fmax(fmax(fmax(v0, v1), inter), 0.0);                                   // :239-241
    }

    // scalar bounded function at one grid time: BouncyParticleSamplers.jl:39-47, BoomerangSamplers.jl:38-46;
    // non-vectorized Zig-Zag bounds rate = sum(max.(0,.)), ZigZagSamplers.jl:83-86
    __device__ __forceinline__ void eval_scalar(const TimePt &tp, double &val, double &slp) {
        const auto sh = shared_at<Dual>(tp);
        double fv = 0.0, fe = 0.0;
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            const bool valid = in_dim(s, i);
            Dual xt, vt;
            flow(X(s), VU(s), tp, xt, vt);
            const Dual gd = eff(tgt.grad(i, xt, sh), xt, tp);
            double a = gd.v * vt.v, b = sdot(gd, vt);
            if constexpr (kZz) {
                b = (0.0 < a) ? b : 0.0;
                a = fmax(0.0, a);
            }
            fv += valid ? a : 0.0;
            fe += valid ? b : 0.0;
        }
        const double dv = group_sum(fv, G), de = group_sum(fe, G);
        if constexpr (kZz) { val = dv; slp = de; return; }
        const bool sgn = p.signed_bound != 0;
        val = (sgn ? dv : fmax(0.0, dv)) + p.refresh;                 // includes refresh_rate (Q2)
        slp = sgn ? de : ((0.0 < dv) ? de : 0.0);
    }
    // plain value of the scalar bounded function at up to three time points in one pass over the coordinates
    template <int NPT>
    __device__ __forceinline__ void fval_scalar(const TimePt (&tp)[NPT], double (&val)[NPT]) {
        typename Tgt::template Sh<double> sh[NPT];
        double f[NPT];
#pragma unroll
        for (int q = 0; q < NPT; ++q) { sh[q] = shared_at<double>(tp[q]); f[q] = 0.0; }
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            const bool valid = in_dim(s, i);
            const double x = X(s), v = VU(s);
#pragma unroll
            for (int q = 0; q < NPT; ++q) {
                double xt, vt;
                flow(x, v, tp[q], xt, vt);
                double a = eff(tgt.grad(i, xt, sh[q]), xt, tp[q]) * vt;
                if constexpr (kZz) a = fmax(0.0, a);
                f[q] += valid ? a : 0.0;
            }
        }
        const bool sgn = p.signed_bound != 0;
#pragma unroll
        for (int q = 0; q < NPT; ++q) {
            const double dv = group_sum(f[q], G);
            if constexpr (kZz) val[q] = dv;
            else val[q] = (sgn ? dv : fmax(0.0, dv)) + p.refresh;
        }
    }
    // NP grid times in ONE pass over the coordinates (x, v of a slot are read once and the NP dependency chains
    // interleave); per grid time the arithmetic and the summation order are those of eval_scalar.
    template <int NP>
    __device__ __forceinline__ void eval_scalar_multi(const TimePt (&tp)[NP], double (&val)[NP], double (&slp)[NP]) {
        typename Tgt::template Sh<Dual> sh[NP];
        double fv[NP], fe[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) { sh[q] = shared_at<Dual>(tp[q]); fv[q] = 0.0; fe[q] = 0.0; }
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            const bool valid = in_dim(s, i);
            const double x = X(s), v = VU(s);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                Dual xt, vt;
                flow(x, v, tp[q], xt, vt);
                const Dual gd = eff(tgt.grad(i, xt, sh[q]), xt, tp[q]);
                double a = gd.v * vt.v, b = sdot(gd, vt);
                if constexpr (kZz) {
                    b = (0.0 < a) ? b : 0.0;
                    a = fmax(0.0, a);
                }
                fv[q] += valid ? a : 0.0;
                fe[q] += valid ? b : 0.0;
            }
        }
        const bool sgn = p.signed_bound != 0;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double dv = group_sum(fv[q], G), de = group_sum(fe[q], G);
            if constexpr (kZz) { val[q] = dv; slp[q] = de; }
            else {
                val[q] = (sgn ? dv : fmax(0.0, dv)) + p.refresh;              // includes refresh_rate (Q2)
                slp[q] = sgn ? de : ((0.0 < dv) ? de : 0.0);
            }
        }
    }
    template <bool FDM>
    __device__ __forceinline__ void point_scalar(double t, double harg, double &val, double &slp) {
        if constexpr (!FDM) {
            TimePt pt;
            time_pt(t, pt);
            eval_scalar(pt, val, slp);                                                 // :97, :102
        } else {                                                                       // :117-118
            FdPt P;
            fd_point(t, harg, P);
            const TimePt tp3[3] = {P.c, P.m, P.pl};
            double f3[3];
            fval_scalar<3>(tp3, f3);
            val = f3[0];
            slp = fd_slope(P, f3[0], f3[1], f3[2]);
        }
    }
    template <bool FDM>
    __device__ __forceinline__ void bound_invert_scalar(double harg, double E, double &tpo, double &lbo) {
        const double h = harg / (double)(ng() - 1);
        const double refresh = (p.signed_bound != 0) ? p.refresh : 0.0;               // AbstractPDMP.jl:104-112
        double acc = 0.0, cprev = 0.0, bj = 0.0, f0, s0;
        bool found = false;
        tpo = __builtin_huge_val();
        int inv_j = 0;
        double inv_c0 = 0.0, inv_c1 = 1.0;
        auto cell = [&](int j, double f1, double s1) {                                 // cell j from grid times j (f0, s0) and j+1
            const double num = fma(s1, h, f0 - f1);                                    // :123 relative form
            // equal slopes at both ends in every lane (affine gradients under the signed bound: the slope is the same sum at
            // every grid time) -> num / 0 = +-Inf or NaN clamps to h or 0: same bits without the division sequence
            // (BPS on an affine gradient only: elsewhere the slopes differ and the test costs more than it saves)
            const double den = s1 - s0;
            bool fast = false;
            if constexpr (SAMPLER == S_BPS && Tgt::kLinear) fast = (__ballot(den != 0.0) == 0ull);
            const double q = cell_q(num, den, h, fast);                                // :124-125
            const double inter = fma(s0, q, f0);                                       // :127
            bj = fmax(fmax(fmax(f0, f1), inter), 0.0);                                 // :128-130
            bj += refresh;                                                             // :131
            acc += bj;
            const double cj = acc * h;                                                 // :133-134
            if (!found && !(cj < E)) {                                                 // searchsortedfirst, :265 -- the inversion's division
                found = true;                                                          // (:269) is done ONCE after the loop (inv_*), not in
                inv_j = j; inv_c0 = cprev; inv_c1 = cj;                                // every cell some chain of the wave finishes in
                lbo = bj;
            }
            cprev = cj;
            f0 = f1;
            s0 = s1;
        };

        // ---- Grid times across the lanes of the chain (register form, G >= 2, bounded functions that cost O(1) per time point: the
        // affine tail and the quadratic form).  The ten grid times are independent, so lane g of the chain's G evaluates times g, g+G, ...
        // and cells g, g+G, ... (its right neighbour's values come by lane shuffle); the nine cell maxima are then gathered to every
        // lane, which runs the reference's cumulative sum and search in grid order.  Same arithmetic per point and per cell, same order
        // of the cumulative sum: bit-identical to the sequential loop below -- but ceil(10/G) steps instead of up to nine, which is what
        // the wave paid whenever ONE of its chains ran to the horizon (0.9 of BPS's 2.1 bounds per event do).
        if constexpr (kReg && GC >= 2 && GC <= 16 && (kQuad || (kLowRank && !FDM))) {
            constexpr int M = (GNG + GC - 1) / GC;                                     // time points per lane
            if constexpr (kQuad) q_prepare(); else lr_prepare();
            const bool sgn = p.signed_bound != 0;
            double pf[M + 1], ps[M + 1], bj[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int j = g + GC * m;
                const double tj = (j >= GNG - 1) ? harg : (double)j * h;               // (lanes beyond the grid evaluate the horizon again: unused)
                if constexpr (kQuad) {
                    if constexpr (!FDM) { TimePt tq; time_pt(tj, tq); pf[m] = q_bounded(tq, &ps[m]); }
                    else {
                        FdPt P;
                        fd_point(tj, harg, P);
                        pf[m] = q_bounded(P.c, nullptr);
                        ps[m] = fd_slope(P, pf[m], q_bounded(P.m, nullptr), q_bounded(P.pl, nullptr));
                    }
                } else {
                    double dv, de;
                    lr_eval(tj, dv, de);
                    pf[m] = (sgn ? dv : fmax(0.0, dv)) + p.refresh;                    // includes refresh_rate (Q2)
                    ps[m] = sgn ? de : ((0.0 < dv) ? de : 0.0);
                }
            }
            pf[M] = 0.0; ps[M] = 0.0;
            const int nxt = base + ((g + 1) & (GC - 1));                               // the lane that holds the next grid time
#pragma unroll
            for (int m = 0; m < M; ++m) {                                              // cell j = g + G m from times j and j + 1
                double a_f, a_s, b_f, b_s;
                if constexpr (GC == 4 && !defined_no_dpp_bound) {                      // the next lane of the quad by DPP (quad_perm [1,2,3,0]): no LDS round trip
                    a_f = dpp_d<0x39>(pf[m]); a_s = dpp_d<0x39>(ps[m]); b_f = dpp_d<0x39>(pf[m + 1]); b_s = dpp_d<0x39>(ps[m + 1]);
                } else {                                                               // (G = 8: row_shl:1 / row_shr:7 + a select measured no faster than the shuffle)
                    a_f = __shfl(pf[m], nxt, WAVE); a_s = __shfl(ps[m], nxt, WAVE);
                    b_f = __shfl(pf[m + 1], nxt, WAVE); b_s = __shfl(ps[m + 1], nxt, WAVE);
                }
                const bool wrap = g == GC - 1;                                         // time j + 1 sits in lane 0's next slot
                const double f1 = wrap ? b_f : a_f, s1 = wrap ? b_s : a_s, fz = pf[m], sz = ps[m];
                const double num = fma(s1, h, fz - f1);                                // :123 relative form
                const double den = s1 - sz;
                bool fast = false;
                if constexpr (SAMPLER == S_BPS && Tgt::kLinear) fast = (__ballot(den != 0.0) == 0ull);
                const double q = cell_q(num, den, h, fast);                            // :124-125
                const double inter = fma(sz, q, fz);                                   // :127
                bj[m] = fmax(fmax(fmax(fz, f1), inter), 0.0) + refresh;                // :128-131
            }
            int inv_jj = 0;
            double c0 = 0.0, c1 = 1.0, last = 0.0;
#pragma unroll
            for (int j = 0; j < GNG - 1; ++j) {                                        // cum_sum and searchsortedfirst in grid order, :133-134, :265
                double b;
                if constexpr (GC == 4 && !defined_no_dpp_bound) {                      // lane j % 4 of the quad, broadcast by DPP
                    const double src = bj[j / GC];
                    b = (j % 4 == 0) ? dpp_d<0x00>(src) : (j % 4 == 1) ? dpp_d<0x55>(src) : (j % 4 == 2) ? dpp_d<0xAA>(src) : dpp_d<0xFF>(src);
                } else b = __shfl(bj[j / GC], base + (j % GC), WAVE);
                acc += b;
                const double cj = acc * h;
                if (!found && !(cj < E)) { found = true; inv_jj = j; c0 = cprev; c1 = cj; lbo = b; }
                cprev = cj;
                last = b;
            }
            if (found) tpo = (double)inv_jj * h + (E - c0) / (c1 - c0) * h;            // next_event, :269
            else lbo = last;                                                           // box_max[end], :270
            return;
        }
        if constexpr (kQuad) {
            q_prepare();
            auto point = [&](double t, double &val, double &slp) {
                if constexpr (!FDM) {
                    TimePt tq;
                    time_pt(t, tq);
                    val = q_bounded(tq, &slp);                                         // :97, :102
                } else {                                                               // :117-118, finite differences :50-76
                    FdPt P;
                    fd_point(t, harg, P);
                    val = q_bounded(P.c, nullptr);
                    slp = fd_slope(P, val, q_bounded(P.m, nullptr), q_bounded(P.pl, nullptr));
                }
            };
            point(0.0, f0, s0);
            for (int j = 0; j < ng() - 1; ++j) {
                double f1, s1;
                point((j == ng() - 2) ? harg : (double)(j + 1) * h, f1, s1);
                cell(j, f1, s1);
                if (__ballot(!found) == 0ull) break;
            }
        } else
        if constexpr (kLowRank && !FDM) {
            lr_prepare();
            const bool sgn = p.signed_bound != 0;
            auto point = [&](double t, double &val, double &slp) {
                double dv, de;
                lr_eval(t, dv, de);
                val = (sgn ? dv : fmax(0.0, dv)) + p.refresh;                          // includes refresh_rate (Q2)
                slp = sgn ? de : ((0.0 < dv) ? de : 0.0);
            };
            point(0.0, f0, s0);
            for (int j = 0; j < ng() - 1; ++j) {                                        // lazy, cell by cell: a grid time is O(1) now
                double f1, s1;
                point((j == ng() - 2) ? harg : (double)(j + 1) * h, f1, s1);
                cell(j, f1, s1);
                if (__ballot(!found) == 0ull) break;
            }
        } else
        if constexpr (!FDM && !Tgt::kTile) {
            // lazy in batches of three cells: grid times {0,1,2,3}, {4,5,6}, {7,8,9}, ... one coordinate pass per batch (grid time j sits at
            // j h, the last one at the horizon itself; a batch that reaches past the grid evaluates the horizon again and skips those cells)
            const int n = ng();
            {
                TimePt tp[4];
                double f[4], sl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) time_pt((q >= n - 1) ? harg : (double)q * h, tp[q]);
                eval_scalar_multi<4>(tp, f, sl);                                       // :97, :102
                f0 = f[0]; s0 = sl[0];
#pragma unroll
                for (int q = 1; q < 4; ++q) if (q <= n - 1) cell(q - 1, f[q], sl[q]);
            }
            for (int jb = 3; jb < n - 1 && __ballot(!found) != 0ull; jb += 3) {
                TimePt tp[3];
                double f[3], sl[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) time_pt((jb + q >= n - 2) ? harg : (double)(jb + q + 1) * h, tp[q]);
                eval_scalar_multi<3>(tp, f, sl);
#pragma unroll
                for (int q = 0; q < 3; ++q) if (jb + q < n - 1) cell(jb + q, f[q], sl[q]);
            }
        } else {
            double f1, s1;
            point_scalar<FDM>(0.0, harg, f0, s0);
            for (int j = 0; j < ng() - 1; ++j) {
                point_scalar<FDM>((j == ng() - 2) ? harg : (double)(j + 1) * h, harg, f1, s1);
                cell(j, f1, s1);
                if (__ballot(!found) == 0ull) break;
            }
        }
        if (found) tpo = (double)inv_j * h + (E - inv_c0) / (inv_c1 - inv_c0) * h;     // next_event, :269
        else lbo = bj;
    }
    // ---- upper_bound_constant (UpperBound.jl:18-36) + next_event for grid_size == 0: the UNSIGNED rate
    // (AbstractPDMP.jl:121-125) is maximised over [0, harg] by Brent's method -- Optim.jl's univariate Brent()
    // with its default rel_tol = sqrt(eps), abs_tol = eps, iterations = 1000 (third-party, restated from the
    // published algorithm; same statement order as the checker) -- and bounds the rate by that one constant.
    __device__ __forceinline__ void bound_constant(double harg, double E, double &tpo, double &lbo) {
        const double rel_tol = 1.4901161193847656e-08, abs_tol = 2.220446049250313e-16;
        const double golden = 0.5 * (3.0 - 2.23606797749979);
        double x_lower = 0.0, x_upper = harg, raw;
        double new_x = x_lower + golden * (x_upper - x_lower);
        double new_f = -true_rate(new_x, raw);
        double step = 0.0, old_step = 0.0;
        double old_x = new_x, old_old_x = new_x, old_f = new_f, old_old_f = new_f;
        int iteration = 0;
        bool active = true;
        while (__ballot(active) != 0ull) {
            if (active) {
                double pp = 0.0, q = 0.0;
                const double x_tol = rel_tol * fabs(new_x) + abs_tol;
                const double x_mid = (x_upper + x_lower) / 2;
                if (iteration >= 1000 || fabs(new_x - x_mid) <= 2 * x_tol - (x_upper - x_lower) / 2) {
                    active = false;
                } else {
                    ++iteration;
                    if (fabs(old_step) > x_tol) {
                        const double r = (new_x - old_x) * (new_f - old_old_f);
                        q = (new_x - old_old_x) * (new_f - old_f);
                        pp = (new_x - old_old_x) * q - (new_x - old_x) * r;
                        q = 2 * (q - r);
                        if (q > 0) pp = -pp; else q = -q;
                    }
                    if (fabs(pp) < fabs(q * old_step / 2) && pp < q * (x_upper - new_x) && pp < q * (new_x - x_lower)) {
                        old_step = step;
                        step = pp / q;
                        const double x_temp = new_x + step;
                        if ((x_temp - x_lower) < 2 * x_tol || (x_upper - x_temp) < 2 * x_tol)
                            step = (new_x < x_mid) ? x_tol : -x_tol;
                    } else {
                        old_step = (new_x < x_mid) ? x_upper - new_x : x_lower - new_x;
                        step = golden * old_step;
                    }
                    const double xt = (fabs(step) >= x_tol) ? new_x + step : new_x + ((step > 0) ? x_tol : -x_tol);
                    const double ft = -true_rate(xt, raw);
                    if (ft < new_f) {
                        if (xt < new_x) x_upper = new_x; else x_lower = new_x;
                        old_old_x = old_x; old_old_f = old_f;
                        old_x = new_x; old_f = new_f;
                        new_x = xt; new_f = ft;
                    } else {
                        if (xt < new_x) x_lower = xt; else x_upper = xt;
                        if (ft <= old_f || old_x == new_x) {
                            old_old_x = old_x; old_old_f = old_f;
                            old_x = xt; old_f = ft;
                        } else if (ft <= old_old_f || old_old_x == new_x || old_old_x == old_x) {
                            old_old_x = xt; old_old_f = ft;
                        }
                    }
                }
            }
        }
        const double m = -new_f;                               // box_max = [-minimum], :28
        const double c1 = m * (harg - 0.0);                    // cum_sum[2], :32-33; step_size = harg - 0, :35
        lbo = m;
        tpo = (c1 < E) ? __builtin_huge_val() : 0.0 + (E - 0.0) / (c1 - 0.0) * (harg - 0.0);   // next_event, :264-273
    }

    // one pass over the bound of every lane that is in the BOUND phase; true = finished (only the vectorized grid bound is
    // built in instalments, the others finish in one call)
    __device__ __forceinline__ bool bound_advance(double harg, double E, double &tp, double &lb, int budget) {
        if constexpr (SAMPLER == S_RHMC) {      // the 2-point BoundBox of the constant refresh clock (RandomizedHamiltonianMonteCarlo.jl:208-218) + next_event
            const double m = p.refresh;
            const double c1 = m * (harg - 0.0);
            lb = m;
            tp = (c1 < E) ? __builtin_huge_val() : 0.0 + (E - 0.0) / (c1 - 0.0) * (harg - 0.0);
            return true;
        }
        if (brent()) { bound_constant(harg, E, tp, lb); return true; }
        if constexpr (kZz) {
            if (vectorized()) {
                if constexpr (kReg) return bound_advance_vect<true, FD>(harg, E, tp, lb, budget);
                else return zz_signed() ? bound_advance_vect<true, FD>(harg, E, tp, lb, budget) : bound_advance_vect<false, FD>(harg, E, tp, lb, budget);
            }
        }
        if constexpr (!(kReg && kZz)) bound_invert_scalar<FD>(harg, E, tp, lb);
        return true;
    }

    // ---- unsigned total rate of the accept test (SamplingLoopInplace.jl:116, Q3)
    // `raw` = the summed term before max/refresh: at an accepted proposal the state flows to exactly the
    // point evaluated here (same expression x + v*tp, same stored doubles), so the velocity jump's
    // sum(lambda) (Zig-Zag) / dot(grad-U, v) (BPS) is this very number and is not recomputed.
    __device__ __forceinline__ double true_rate(double t, double &raw) {
        if constexpr (SAMPLER == S_RHMC) { raw = 0.0; return p.refresh; }      // rate = (_x0, _v0, _t) -> refresh_rate, :133
        if constexpr (kQuad) {
            q_prepare();
            TimePt tq;
            time_pt(t, tq);
            raw = q_raw(tq, nullptr);
            return fmax(0.0, raw) + p.refresh;                                  // Boomerang :38-41
        }
        if constexpr (kLowRank) {
            lr_prepare();                                                       // (already prepared by the bound on the grid path)
            double fe;
            lr_eval(t, raw, fe);
            return fmax(0.0, raw) + p.refresh;                                  // BPS :39-42
        }
        TimePt tp;
        time_pt(t, tp);
        const auto sh = shared_at<double>(tp);
        double acc = 0.0;
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            double xt, vt;
            flow(X(s), VU(s), tp, xt, vt);
            const double gv = eff(tgt.grad(i, xt, sh), xt, tp) * vt;
            const double term = (kZz) ? fmax(0.0, gv) : gv;     // ZigZagSamplers.jl:83-86
            acc += (in_dim(s, i)) ? term : 0.0;
        }
        const double tot = group_sum(acc, G);
        raw = tot;
        if constexpr (kZz) return tot;
        else return fmax(0.0, tot) + p.refresh;                                 // BPS :39-42, Boomerang :38-41
    }

    // grad-U's shared part at the stored state itself (no flow)
    __device__ __forceinline__ typename Tgt::template Sh<double> shared_state() {
        typename Tgt::template Sh<double> sh{};
        if constexpr (Tgt::kUser) {
            TimePt t0; t0.t = 0.0; t0.s = 0.0; t0.c = 1.0; t0.sp = 1.0; t0.spe = 0.0;
            sh = shared_at<double>(t0);
        } else if constexpr (Tgt::kShared && !Tgt::kTile) {
            if (p.d >= 2) { double x0, v0, x1, v1; coord(0, x0, v0); coord(1, x1, v1); sh = tgt.template shared<double>(x0, x1); }
        }
        return sh;
    }
    // ---- RHMC: velocity-Verlet flow (RandomizedHamiltonianMonteCarlo.jl:97-130) of (x, v) in place: n = floor(t/h) steps of
    // size h and one of size r = t - n h; chains of a wave take their own numbers of steps (per-lane predicate).
    __device__ __forceinline__ void rhmc_step(double hh) {
        auto sh = shared_state();
        PDMP_SLOTS(s) {                                   // kick with grad-U of the whole old position first (a potential given as
            const int i = s * G + g;                      // source reads every coordinate), then the drift
            if (in_dim(s, i)) V(s) = V(s) - (0.5 * hh) * tgt.grad(i, X(s), sh);
        }
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            if (in_dim(s, i)) X(s) = X(s) + hh * V(s);
        }
        sh = shared_state();
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            if (in_dim(s, i)) V(s) = V(s) - (0.5 * hh) * tgt.grad(i, X(s), sh);
        }
    }
    __device__ __forceinline__ void rhmc_flow(double t) {
        const double h = p.rh_step_size;
        const long long n = (t == 0.0) ? 0 : (long long)floor(t / h);
        const double r = t - (double)n * h;
        const long long steps = (t == 0.0) ? 0 : n + ((r > 0.0) ? 1 : 0);
        for (long long k = 0; __ballot(k < steps) != 0ull; ++k) {
            if (k < steps) rhmc_step((k < n) ? h : r);
        }
    }

    __device__ __forceinline__ void flow_state(double t) {
        if constexpr (SAMPLER == S_RHMC) { rhmc_flow(t); return; }
        flow_state_closed(t);
    }
    __device__ __forceinline__ void flow_state_closed(double t) const {
        // register form: an opaque copy of t keeps the compiler from holding the KC positions true_rate just computed
        // (same expression, same bits) in registers across the accept test
        if constexpr (kReg) asm volatile("" : "+v"(t));
        TimePt tp;
        time_pt(t, tp);
        PDMP_SLOTS(s) {
            double xt, vt;
            flow(X(s), VU(s), tp, xt, vt);
            X(s) = xt;
            if constexpr (SAMPLER == S_BOOMERANG) V(s) = vt;
        }
        sz_ok = false;
        lr_ok = false;
    }

    // ---- velocity_jump: ZigZagSamplers.jl:101-107, BouncyParticleSamplers.jl:50-74, BoomerangSamplers.jl:49-67
    // ---- _velocity_jump_event_chain (ForwardEventChainMonteCarlo.jl:127-175; default variant: normal = false, speed_factor = 1)
    // with _orthogonal_switch (:60-91) / _full_refresh (:108-117).  The d-vectors n, v_orth, e1, e2 live in four LDS tiles;
    // every dot product is a per-lane left-to-right sum over the lane's coordinates and a butterfly over the group (padded
    // coordinates hold exact zeros).  u^(2/(dim-1)) = exp(log(u) * (2/(dim-1))) with the engine's log / exp, as in the checker;
    // randn(key, 2, dim) is column-major: g1[j], g2[j] = the two normals of Box-Muller pair j of the chain's stream.
    __device__ __forceinline__ double tdot(const double *a, const double *b) const {
        double acc = 0.0;
        PDMP_SLOTS(s) acc += a[s * WAVE + lane] * b[s * WAVE + lane];
        return group_sum(acc, G);
    }
    __device__ __forceinline__ bool fecmc_jump(const typename Tgt::template Sh<double> &sh) {
        const int d = p.d;
        double *tn = fe_t[0], *tvo = fe_t[1];
        const double u = r_uniform();                                               // :134
        const double pw = det_exp(det_log(u) * (2.0 / (double)(d - 1)));
        const double sf = p.fe_speed;                                               // speed_factor (:177-218; 1.0: the default jump, same bits)
        const double rho = sf * (-__dsqrt_rn(1.0 - pw));                            // :135 / :185
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            tn[s * WAVE + lane] = in_dim(s, i) ? tgt.grad(i, X(s), sh) : 0.0;       // n = copy(grad-U(x)), :143
            if (!in_dim(s, i)) V(s) = 0.0;
        }
        const double ng = __dsqrt_rn(tdot(tn, tn));
        PDMP_SLOTS(s) tn[s * WAVE + lane] = (ng == 0.0) ? 0.0 : tn[s * WAVE + lane] / ng;   // :144-149
        double acc = 0.0;
        PDMP_SLOTS(s) acc += V(s) * tn[s * WAVE + lane];
        const double vn = group_sum(acc, G);                                        // v . n, :152
        PDMP_SLOTS(s) { const double vp = vn * tn[s * WAVE + lane]; tvo[s * WAVE + lane] = V(s) - vp; }   // :152-153
        double nvo = __dsqrt_rn(tdot(tvo, tvo));
        if (nvo < 1e-10) {                                                          // :156-159
            PDMP_SLOTS(s) { const int i = s * G + g; tvo[s * WAVE + lane] = in_dim(s, i) ? r_normal_coord(i) : 0.0; }
            r_normals_done(d);
            const double dn = tdot(tvo, tn);
            PDMP_SLOTS(s) tvo[s * WAVE + lane] = tvo[s * WAVE + lane] - dn * tn[s * WAVE + lane];
            nvo = __dsqrt_rn(tdot(tvo, tvo));
        }
        const double u2 = r_uniform();                                              // :161
        const double sq = __dsqrt_rn(sf * sf - rho * rho);                          // :164 / :207
        if (u2 >= p.fe_mix_p) {                                                     // :162-166
            PDMP_SLOTS(s) { const int i = s * G + g; if (in_dim(s, i)) V(s) = tvo[s * WAVE + lane] / nvo * sq + rho * tn[s * WAVE + lane]; }
            return false;
        }
        fe_rho = rho; fe_sq = sq;
        return true;                                                                // refresh of the orthogonal part: fecmc_refresh()
    }

    // Second half of the jump for the chains whose lanes pass `mine`, called by the whole wave: the normals of
    // randn(key, 2, dim) (orthogonal switch) or randn(key, dim) (full refresh) are drawn one Box-Muller pair per LANE of the
    // wave into the chain's tiles (one pass for dim <= 64 resp. 128) instead of one pair per coordinate on the chain's own lanes.
    double fe_rho = 0.0, fe_sq = 0.0;
    __device__ __forceinline__ void fecmc_refresh(bool mine) {
        const int d = p.d, logG = __builtin_ctz(G);
        double *tn = fe_t[0], *tvo = fe_t[1], *t1 = fe_t[2], *t2 = fe_t[3];
        unsigned long long todo = __ballot(mine && g == 0);
        while (todo != 0ull) {
            const int L = (int)__builtin_ctzll(todo);                               // first lane of the chain
            todo &= todo - 1ull;
            const NormalSrc r = normal_src<true>(L);
            if (p.fe_switch) {                                                      // g1[P], g2[P] = pair P
                for (int P = lane; P < d; P += WAVE) {
                    double z0, z1;
                    src_pair(r, P, z0, z1);
                    const int o = (P >> logG) * WAVE + L + (P & (G - 1));
                    t1[o] = z0;
                    t2[o] = z1;
                }
            } else {                                                                // z[2P], z[2P+1] = pair P
                for (int P = lane; 2 * P < d; P += WAVE) {
                    double z0, z1;
                    src_pair(r, P, z0, z1);
                    const int i0 = 2 * P, i1 = 2 * P + 1;
                    t2[(i0 >> logG) * WAVE + L + (i0 & (G - 1))] = z0;
                    if (i1 < d) t2[(i1 >> logG) * WAVE + L + (i1 & (G - 1))] = z1;
                }
            }
        }
        if (!mine) return;
        const double rho = fe_rho, sq = fe_sq;
        PDMP_SLOTS(s) { const int i = s * G + g; if (!in_dim(s, i)) { t1[s * WAVE + lane] = 0.0; t2[s * WAVE + lane] = 0.0; } }
        if (p.fe_switch) {                                                          // _orthogonal_switch, :60-91
            r_normals_done(2 * d);
            const double d1 = tdot(t1, tn);
            PDMP_SLOTS(s) t1[s * WAVE + lane] = t1[s * WAVE + lane] - d1 * tn[s * WAVE + lane];        // :67
            const double d2 = tdot(t2, tn);
            PDMP_SLOTS(s) t2[s * WAVE + lane] = t2[s * WAVE + lane] - d2 * tn[s * WAVE + lane];        // :68
            const double n1 = __dsqrt_rn(tdot(t1, t1));
            PDMP_SLOTS(s) t1[s * WAVE + lane] = t1[s * WAVE + lane] / n1;                              // :71
            const double d21 = tdot(t2, t1);
            PDMP_SLOTS(s) t2[s * WAVE + lane] = t2[s * WAVE + lane] - d21 * t1[s * WAVE + lane];       // :72
            const double n2 = __dsqrt_rn(tdot(t2, t2));
            PDMP_SLOTS(s) t2[s * WAVE + lane] = t2[s * WAVE + lane] / n2;                              // :73
            const double a = tdot(tvo, t1), b = tdot(tvo, t2);
            double sn = 0.0, cs = 0.0;
            if (p.fe_ran_p) { const double ut = r_uniform(); det_sincos_2pi(ut, sn, cs); }             // theta = rand * 2pi, :79
            double acc = 0.0;
            PDMP_SLOTS(s) {
                const double vo = tvo[s * WAVE + lane], e1 = t1[s * WAVE + lane], e2 = t2[s * WAVE + lane];
                const double vr = (vo - a * e1) - b * e2;                                              // :76
                const double z = p.fe_ran_p ? (vr + (cs * e1 + sn * e2) * a) + (sn * e1 - cs * e2) * b  // :80
                                            : (vr + e2 * a) + e1 * b;                                  // :77
                acc += vo * z;
                t2[s * WAVE + lane] = z;
            }
            if (p.fe_positive) {                                                                       // :83-85
                const double sdot = group_sum(acc, G);
                const double sg = (sdot > 0.0) ? 1.0 : ((sdot < 0.0) ? -1.0 : sdot);
                PDMP_SLOTS(s) t2[s * WAVE + lane] = t2[s * WAVE + lane] * sg;
            }
        } else {                                                                    // _full_refresh, :108-117
            r_normals_done(d);
            const double nz = __dsqrt_rn(tdot(t2, t2));
            PDMP_SLOTS(s) t2[s * WAVE + lane] = t2[s * WAVE + lane] / nz;
            const double dn = tdot(t2, tn);
            PDMP_SLOTS(s) t2[s * WAVE + lane] = t2[s * WAVE + lane] - dn * tn[s * WAVE + lane];
        }
        const double np = __dsqrt_rn(tdot(t2, t2));                                 // :170-172
        PDMP_SLOTS(s) { const int i = s * G + g; if (in_dim(s, i)) V(s) = t2[s * WAVE + lane] / np * sq + rho * tn[s * WAVE + lane]; }
    }

    // returns true when the chain asks for a velocity refresh: the normals are drawn by refresh_velocities(), outside the
    // divergent accept branch, with all 64 lanes of the wave
    __device__ __forceinline__ bool velocity_jump(double raw) {
        lr_ok = false;
        TimePt t0;
        t0.t = 0.0; t0.s = 0.0; t0.c = 1.0; t0.sp = 1.0; t0.spe = 0.0;
        if constexpr (SAMPLER == S_SUZZ) suzz_speed(t0, true);          // speed(x) of the state itself, SpeedUpZigZagSamplers.jl:104
        typename Tgt::template Sh<double> sh;
        if constexpr (Tgt::kTile) {
            sh = {tgt.tg[0], tgt.tb};            // grad-U at the accepted point: tile 0 was filled by true_rate(tp)
            tgt.b_valid = false;                 // v is about to change
        } else if constexpr (Tgt::kUser) {
            sh = shared_at<double>(t0);          // the state already sits at the accepted point: flow by t = 0 is exact
        } else if constexpr (Tgt::kShared) {
            if (p.d >= 2) { double x0, v0, x1, v1; coord(0, x0, v0); coord(1, x1, v1); sh = tgt.template shared<double>(x0, x1); }
        }
        const int d = p.d;
        if constexpr (kZz) {
            double S = raw;                                              // sum(max.(0, grad-U .* v)), ZigZagSamplers.jl:102-103
            if constexpr (kSticky) {                                     // the accepted rate used v .* is_active; the jump uses the full v
                double a = 0.0;
                PDMP_SLOTS(s) { const int i = s * G + g; a += (in_dim(s, i)) ? fmax(0.0, eff(tgt.grad(i, X(s), sh), X(s), t0) * V(s)) : 0.0; }
                S = group_sum(a, G);
            }
            const double u = draw_uniform();                             // Categorical draw
            if (!(S > 0.0)) { guard++; return false; }                         // Q11
            // Categorical(lambda / S) (ZigZagSamplers.jl:102-104): the reference walks cp += lambda_i / S until
            // cp > u; the engine walks the un-normalised cumulative rate against theta = u * S (same index up
            // to rounding at a cell boundary, no fp64 division per coordinate) -- GROUP order of the checker.
            const double theta = u * S;
            // Level 1: slot totals T_s (the butterfly value equals the last lane of the Kogge-Stone scan bit
            // for bit), first slot whose running total exceeds theta.  Level 2: one scan inside that slot.
            double carry = 0.0, carry_at = 0.0, lam_hit = 0.0;           // lam_hit: this lane's rate in the slot that was hit
            int ms_hit = -1;
            PDMP_SLOTS(s) {
                const int i = s * G + g;
                const double lam = (in_dim(s, i)) ? fmax(0.0, eff(tgt.grad(i, X(s), sh), X(s), t0) * V(s)) : 0.0;
                const double T = group_sum(lam, G);
                if (ms_hit < 0 && (carry + T > theta)) { ms_hit = s; carry_at = carry; lam_hit = lam; }
                carry = carry + T;
                // LDS form: stop once every chain of the wave has its slot.  Register form: straight-line code -- with 64/G chains per
                // wave the last hit slot is nearly always one of the last, and without the ballot + branch the K slot totals interleave
                if constexpr (!kReg) { if (__ballot(ms_hit < 0) == 0ull) break; }
            }
            int m = -1;
            {
                const int sq = (ms_hit < 0) ? 0 : ms_hit;
                const int i = sq * G + g;
                const bool valid = (ms_hit >= 0) && in_dim(sq, i);
                const double lam = lam_hit;                              // 0 where the coordinate is padding or no slot was hit
                const double inc = group_scan_incl(lam, g, G);
                const unsigned long long ball = __ballot(valid && (carry_at + inc > theta));
                const unsigned long long grp = (G == WAVE) ? ball : ((ball >> base) & ((1ull << G) - 1ull));
                if (grp != 0ull) m = sq * G + (int)__builtin_ctzll(grp);
            }
            if (m < 0) m = d - 1;
            const int ms = m / G;
            if constexpr (kReg) { PDMP_SLOTS(s) if (s == ms && g == m - ms * G) V(s) = -V(s); }
            else if (g == m - ms * G) V(ms) = -V(ms);                    // v[m] *= -1
            sz_ok = false;
            return false;
        } else if constexpr (SAMPLER == S_RHMC) {
            // Horowitz refreshment v <- cos(phi) v + sin(phi) xi (:144-149): every accepted event of every chain needs d normals, so
            // the pairs are shared inside the chain's own lane group (lane g: pairs g, g+G, ...), each updating its two coordinates
            double sphi, cphi;
            det_sincos(p.rh_phi, sphi, cphi);
            const int logG = __builtin_ctz(G);
            for (int P = g; 2 * P < d; P += G) {
                double z0, z1;
                r_normal_pair(P, z0, z1);
                const int i0 = 2 * P, i1 = 2 * P + 1;
                const int o0 = (i0 >> logG) * WAVE + base + (i0 & (G - 1)), o1 = (i1 >> logG) * WAVE + base + (i1 & (G - 1));
                if constexpr (kReg) {            // register form: v lives in registers; the normals cross the lanes through the exchange tile
                    vs[o0] = z0;
                    if (i1 < d) vs[o1] = z1;
                } else {
                    vs[o0] = cphi * vs[o0] + sphi * z0;
                    if (i1 < d) vs[o1] = cphi * vs[o1] + sphi * z1;
                }
            }
            r_normals_done(d);
            if constexpr (kReg) { PDMP_SLOTS(s) { const int i = s * G + g; if (in_dim(s, i)) V(s) = cphi * V(s) + sphi * vs[s * WAVE + lane]; } }   // (the same expression per coordinate)
            return false;
        } else if constexpr (SAMPLER == S_FECMC) {
            return fecmc_jump(sh);
        } else {
            double acc = 0.0;
            double gv;
            if constexpr (SAMPLER == S_BPS) {
                gv = raw;                                                // dot(grad-U(x), v), BouncyParticleSamplers.jl:51-52
            } else {
                PDMP_SLOTS(s) {
                    const int i = s * G + g;
                    const double gi = tgt.grad(i, X(s), sh) - X(s);      // grad-U_effective, BoomerangSamplers.jl:51-52 (Q8)
                    acc += (in_dim(s, i)) ? gi * V(s) : 0.0;
                }
                gv = group_sum(acc, G);
            }
            const double bounce = fmax(0.0, gv);
            const double prob = bounce / (bounce + p.refresh);           // NaN if both 0 (Q9)
            const double u = draw_uniform();
            if (u < prob) {
                acc = 0.0;
                PDMP_SLOTS(s) {
                    const int i = s * G + g;
                    double gi = tgt.grad(i, X(s), sh);
                    if constexpr (SAMPLER == S_BOOMERANG) gi = gi - X(s);
                    acc += (in_dim(s, i)) ? gi * gi : 0.0;
                }
                const double gg = group_sum(acc, G);
                if constexpr (SAMPLER == S_BPS) {
                    if (gg == 0.0) return false;
                    const double scale = 2.0 * gv / gg;                  // :60
                    PDMP_SLOTS(s) {
                        const int i = s * G + g;
                        const double gi = tgt.grad(i, X(s), sh);
                        if (in_dim(s, i)) V(s) = V(s) - scale * gi;
                    }
                } else {
                    const double nrm = __dsqrt_rn(gg);                   // :58
                    acc = 0.0;
                    PDMP_SLOTS(s) {
                        const int i = s * G + g;
                        const double e = (tgt.grad(i, X(s), sh) - X(s)) / nrm;
                        acc += (in_dim(s, i)) ? V(s) * e : 0.0;
                    }
                    const double ve = 2.0 * group_sum(acc, G);
                    PDMP_SLOTS(s) {
                        const int i = s * G + g;
                        const double e = (tgt.grad(i, X(s), sh) - X(s)) / nrm;
                        if (in_dim(s, i)) V(s) = V(s) - ve * e;                 // :59
                    }
                }
            } else {
                return true;                                               // randn!, :67 / randn(dim), :65 -> refresh_velocities
            }
        }
        return false;
    }

    // Velocity refresh of the chains whose lanes pass `mine` (BouncyParticleSamplers.jl:65-71, BoomerangSamplers.jl:61-65).
    // Called by the whole wave: for each refreshing chain the Box-Muller pairs (one Philox block, log, sqrt, sincos per
    // PAIR; pair P = normals 2P, 2P+1 of the chain's stream) are spread over all 64 lanes -- one pass for d <= 128 --
    // instead of ceil(d / 2G) passes on the chain's own G lanes while the rest of the wave idles.  Same stream positions,
    // same arithmetic: the normals are those of the chain-local loop.
    __device__ __forceinline__ void refresh_velocities(bool mine) {
        const int d = p.d, logG = __builtin_ctz(G);
        const int P = (d + 1) >> 1;                                       // Box-Muller pairs per chain
        const unsigned long long todo = __ballot(mine && g == 0);        // first lanes of the refreshing chains
        const int R = (int)__builtin_popcountll(todo);
        // The pairs of ALL refreshing chains of the wave, back to back over the 64 lanes: lane l of pass q0 takes flat pair q0 + l =
        // (refreshing chain r, pair q - r P) -- d = 50: two chains per pass, d = 100 with eight chains: 6.25 passes instead of 8.
        if constexpr (SAMPLER == S_BOOMERANG) {                           // Boomerang (d = 100, config 5: 50 pairs per chain): measured slower when
            unsigned long long rest = todo;                               // packed -- one chain at a time, its stream counters in SGPRs
            while (rest != 0ull) {
                const int L = (int)__builtin_ctzll(rest);
                rest &= rest - 1ull;
                const NormalSrc r = normal_src<true>(L);
                for (int pr = lane; pr < P; pr += WAVE) {
                    double z0, z1;
                    src_pair(r, pr, z0, z1);
                    const int i0 = 2 * pr, i1 = 2 * pr + 1;
                    vs[(i0 >> logG) * WAVE + L + (i0 & (G - 1))] = z0;
                    if (i1 < d) vs[(i1 >> logG) * WAVE + L + (i1 & (G - 1))] = z1;
                }
            }
        } else
        for (int q0 = 0; q0 < R * P; q0 += WAVE) {
            const int q = q0 + lane;
            const bool act = q < R * P;
            int L = 0, pr = 0;                                            // first lane of my pair's chain, pair index inside it
            unsigned long long rest = todo;
            for (int r = 0; r < R; ++r) {                                 // R <= 64 / G iterations of two compares
                const int Lr = (int)__builtin_ctzll(rest);
                rest &= rest - 1ull;
                if (q >= r * P && q < (r + 1) * P) { L = Lr; pr = q - r * P; }
            }
            const NormalSrc r2 = normal_src<false>(L);                    // counters (or the tape position) of lane L
            if (act) {
                double z0, z1;
                src_pair(r2, pr, z0, z1);
                const int i0 = 2 * pr, i1 = 2 * pr + 1;
                vs[(i0 >> logG) * WAVE + L + (i0 & (G - 1))] = z0;
                if (i1 < d) vs[(i1 >> logG) * WAVE + L + (i1 & (G - 1))] = z1;
            }
        }
        if (mine) {
            lr_ok = false;
            r_normals_done(d);
            if constexpr (kReg) { PDMP_SLOTS(s) vr[s] = in_dim(s, s * G + g) ? vs[s * WAVE + lane] : 0.0; }   // register form: vs is an exchange tile only
            double acc = 0.0;
            PDMP_SLOTS(s) {
                const int i = s * G + g;
                const double z = V(s);
                acc += (in_dim(s, i)) ? z * z : 0.0;
            }
            if (SAMPLER == S_BPS && !p.gaussian_velocity) {              // normalize!, :71
                const double inv = 1.0 / __dsqrt_rn(group_sum(acc, G));
                PDMP_SLOTS(s) V(s) = V(s) * inv;
            }
        }
    }

    // ---- StickyZigZag events (src/StickySamplingLoop.jl)
    __device__ __forceinline__ bool group_any(bool b) const {
        const unsigned long long ball = __ballot(b);
        return ((G == WAVE) ? ball : ((ball >> base) & ((1ull << G) - 1ull))) != 0ull;
    }
    __device__ __forceinline__ double sticky_thaw_rate() const {           // sum of kappa over the frozen coordinates, :40-45
        double a = 0.0;
        PDMP_SLOTS(s) { const int i = s * G + g; a += (in_dim(s, i) && !active(s)) ? kap[i] : 0.0; }
        return group_sum(a, G);
    }
    __device__ __forceinline__ bool sticky_crosses(double et) const {      // any(x .* flow(x, v_active, et) .< 0), :51-58
        bool cr = false;
        PDMP_SLOTS(s) { const double xe = fma(VU(s), et, X(s)); cr = cr || (X(s) * xe < 0.0); }
        return group_any(cr);
    }
    // move_to_axes_and_stick, :73-107: the first axis the active coordinates reach (ties: lowest index); returns the time
    __device__ __forceinline__ double sticky_stick() {
        double tmin = __builtin_huge_val();
        int imin = 0x7fffffff;
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            if (in_dim(s, i) && active(s)) {
                const double dj = X(s) * V(s);
                if (dj < 0.0) { const double tj = -dj; if (tj < tmin) { tmin = tj; imin = i; } }
            }
        }
        for (int off = 1; off < G; off <<= 1) {                              // (time, index) minimum over the group
            const double to = __shfl_xor(tmin, off, WAVE);
            const int io = __shfl_xor(imin, off, WAVE);
            if (to < tmin || (to == tmin && io < imin)) { tmin = to; imin = io; }
        }
        flow_state_closed(tmin);
        PDMP_SLOTS(s) { if (s * G + g == imin) amask &= ~(1u << s); }
        sz_ok = false;
        return tmin;
    }
    // thaw_one_coordinate, :138-164: the frozen coordinates walked in index order against u = rand * total (`acc >= u`), slots of G
    // with an inclusive scan inside the slot (GROUP order of the checker)
    __device__ __forceinline__ void sticky_thaw() {
        flow_state_closed(tt);
        const double u = r_uniform() * thaw_rate;
        double carry = 0.0;
        int pick = -1, last = -1;
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            const bool fr = in_dim(s, i) && !active(s);
            const double inc = group_scan_incl(fr ? kap[i] : 0.0, g, G);
            const unsigned long long ball = __ballot(pick < 0 && fr && (carry + inc >= u));
            const unsigned long long grp = (G == WAVE) ? ball : ((ball >> base) & ((1ull << G) - 1ull));
            const unsigned long long fb = __ballot(fr);
            const unsigned long long fg = (G == WAVE) ? fb : ((fb >> base) & ((1ull << G) - 1ull));
            if (fg != 0ull) last = s * G + 63 - (int)__builtin_clzll(fg);
            if (pick < 0 && grp != 0ull) pick = s * G + (int)__builtin_ctzll(grp);
            carry = carry + __shfl(inc, base + G - 1, WAVE);                 // the slot's total = the scan's last lane
        }
        if (pick < 0) pick = last;
        PDMP_SLOTS(s) { if (s * G + g == pick) amask |= (1u << s); }
        sz_ok = false;
    }

    // ---- record!, Composites.jl:239-260
    __device__ __forceinline__ void record(int64_t c, int64_t k, double t, double horizon, double ar,
                                           int32_t errored, const double (&ev)[5], int32_t rejected,
                                           int32_t hits) const {
        const size_t e = (size_t)c * (size_t)p.n_sk + (size_t)k;
        double *Xe = p.X + e * (size_t)p.d, *Ve = p.V + e * (size_t)p.d;
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            if (in_dim(s, i)) { Xe[i] = X(s); Ve[i] = V(s); }
            if constexpr (kSticky) { if (p.IA && in_dim(s, i)) p.IA[e * (size_t)p.d + i] = (uint8_t)active(s); }   // Composites.jl:256
        }
        if (g == 0) {
            p.t[e] = t;
            if constexpr (!LEAN) {
                if (p.H) p.H[e] = horizon;
                if (p.AR) p.AR[e] = ar;
                if (p.EB) p.EB[e] = errored;
                if (p.RJ) p.RJ[e] = rejected;
                if (p.HH) p.HH[e] = hits;
                if (p.EV) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) p.EV[5 * e + q] = ev[q];
                }
            }
        }
    }

    // ---- sample_skeleton (sample.jl:253-284) for the chains of this wave, events of segment r.
    // Segments end right after an accepted event, where the thinning state is just
    // (x, v, t, horizon, ar, rng counter): every other field of PDMPState is reset or rebuilt
    // (SamplingLoopInplace.jl:28-31,69-76,181-183), so the hand-off between segments is 2d+4 doubles.
    __device__ __forceinline__ void run_segment(int64_t c, bool chain_ok, int r, int64_t seg_len, double *state, int nseg = 1) {
        const int d = p.d;
        const int64_t k_begin = 1 + (int64_t)r * seg_len;
        int64_t k_end = k_begin + seg_len;
        if (k_end > p.n_sk) k_end = p.n_sk;
        // cs: the chain's index in the ORIGINAL ensemble (state slot, initial state, random stream); c: its index in this launch (outputs)
        int64_t cs = chain_ok ? c : 0;
        if constexpr (!LEAN) { if (p.chain_map && chain_ok) cs = p.chain_map[c]; }
        // hand-off state of a chain: x[d], v[d], t, horizon, ar, stream position -- and, for StickyZigZag (round 3), the active masks of the
        // chain's G lanes (one word each: bit s = the coordinate of slot s moves), so that sticky chains too run in segments and resume
        const size_t stride = (size_t)(2 * d + 4) + (kSticky ? (size_t)G : 0);
        double *st = state + (size_t)cs * stride;
        double t = 0.0, horizon = p.tmax, ar = 0.0;
        uint64_t n0 = 0;
        bool from_state = r > 0;                                          // (wave-uniform)
        if constexpr (!LEAN) from_state = from_state || p.resume != 0;
        if (from_state && chain_ok) {
            t = st[2 * d]; horizon = st[2 * d + 1]; ar = st[2 * d + 2];
            n0 = (uint64_t)__double_as_longlong(st[2 * d + 3]);
        }
        PDMP_SLOTS(s) {
            const int i = s * G + g;
            const bool ok = chain_ok && in_dim(s, i);
            if (!from_state) {
                X(s) = ok ? p.xinit[(size_t)cs * d + i] : 0.0;           // copies: inputs are never mutated (Q7)
                V(s) = ok ? p.vinit[(size_t)cs * d + i] : 0.0;
            } else {
                X(s) = ok ? st[i] : 0.0;
                V(s) = ok ? st[d + i] : 0.0;
            }
        }
        rng.init_lane(p.seed, (uint64_t)(p.chain0 + cs));
        rng.restore(n0);
        if (chain_ok) tape_open(cs);
        amask = 0xffffffffu;                                              // is_active = trues(d), Composites.jl:95
        if constexpr (kSticky) { if (from_state && chain_ok) amask = (uint32_t)__double_as_longlong(st[2 * d + 4 + g]); }
        lr_ok = false;
        if constexpr (Tgt::kTile) tgt.b_valid = false;
        double ts = 0.0, tp = 0.0, E = 0.0, lb = 0.0;
        int32_t errored = 0, rejected = 0, hits = 0;
        double ev[5] = {0, 0, 0, 0, 0};
        const bool adaptive = p.adaptive != 0;
        bool done = !chain_ok;
        if (!done && r == 0) record(c, 0, t, horizon, ar, 0, ev, 0, 0);   // sample.jl:272 (a resumed chunk: the previous chunk's last event again)
        int32_t k = (int32_t)k_begin, it = 0;                             // host guarantees n_sk, cap < 2^31
        const int32_t cap = (int32_t)p.cap, kend = (int32_t)k_end;
        // time-horizon variant: `while state.t < T` (sample.jl:372) -- a chain that has reached t_stop (in an earlier segment, or with
        // T = 0 at its initial state) records nothing more; t_stop = +Inf otherwise
        bool fin_before = false, stopped = false;
        if constexpr (!LEAN) { fin_before = !(t < p.t_stop); if (fin_before) done = true; }
        if (k >= kend) done = true;
        // Per-lane phase: BOUND (mode FRESH / ERR / REUSE) or PROPOSE.  The BoundBox is never kept across
        // loop iterations: after a rejection the reference re-inverts the SAME bound with a larger
        // exponential (if_reject!, :188-203; Q4); x, v and the horizon the bound was built for (hB) are
        // unchanged since it was built, so rebuilding it gives the same bits and frees its 38 registers
        // (3 waves per SIMD instead of 2 for ~4 % more bound evaluations).
        enum : int { FRESH = 0, ERR = 1, REUSE = 2 };
        bool propose = false, brun = false;                              // brun: this lane's bound is under construction (bound_advance)
        int mode = FRESH;
        double hB = horizon;
        const int budget = (G == WAVE) ? WAVE : cell_budget;               // one chain per wave: nobody to wait for
        const bool resumable = kZz && vectorized() && !brent() && !Tgt::kTile;   // bounds built by bound_advance_vect
#ifdef PDMP_STAMP
        unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
#define STAMP(k) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; } while (0)
        unsigned long long st_last = __builtin_amdgcn_s_memtime();
#else
#define STAMP(k)
#endif
        while (__ballot(!done) != 0ull) {
            STAMP(5);
            if constexpr (TAPE) { if (tp_over) done = true; }             // tape exhausted: stop this chain (reported below)
            rng_prefetch();
            bool accepted = false, refresh = false, proposed = false, event_out = false;   // event_out: a StickyZigZag stick / thaw event
            // The flow of a pass -- to the horizon (move_to_horizon!, move_to_horizon2!) or to the accepted proposal (if_accept!) -- happens
            // in ONE place, behind the propose phase: a lane does at most one of the three per pass, and one call site instead of three
            // keeps the state registers free of copies at the joins of the divergent regions (round 3: 189 v_mov_b64 in the event loop).
            bool moves = false;
            double tflow = 0.0, raw = 0.0;
            if (!done && !propose) {
                // one_step_of_thinning! (:65-85), erroneous_acceptance_rate! (:131-151) or if_reject! (:188-203)
                if (!brun) {
                    const double harg = (mode == ERR) ? horizon / 2 : ((mode == REUSE) ? hB : horizon);
                    hB = harg;
                    const double e1 = draw_exponential();
                    E = (mode == REUSE) ? E + e1 : e1;                    // cumulative only after a rejection (:190)
                    if (mode == REUSE && resumable) bound_resume(tp); else bound_begin(tp);
                    brun = true;
                }
                if (bound_advance(hB, E, tp, lb, budget)) {
                    const double harg = hB;
                    brun = false;
                    if constexpr (kSticky) {
                        if (mode == FRESH) {                              // one_step_of_thinning_or_sticking_or_thawing, StickySamplingLoop.jl:40-46
                            thaw_rate = sticky_thaw_rate();
                            tt = __builtin_huge_val();
                            if (thaw_rate != 0.0) tt = r_exponential() / thaw_rate;
                        }
                    }
                    const double tnext = kSticky ? ((tp < tt) ? tp : tt) : tp;   // min(tp, tt); tt = Inf for the other samplers (Composites.jl:100)
                    if (mode == ERR) {
                        if (adaptive) horizon = harg;
                        errored++;
                        const int slot = errored % 5;                     // :148 (1-based errored%5+1)
#pragma unroll
                        for (int q = 0; q < 5; ++q) if (q == slot) ev[q] = ar;
                        propose = tnext < horizon;                        // moves_until_horizon! loop test, :106
                    } else if (mode == REUSE) {
                        if (adaptive) horizon = horizon / 1.04;           // :194
                        rejected++;
                        if (tnext > horizon) {                            // move_to_horizon2!, :205-217 (SamplingLoopInplace.jl:163)
                            moves = true; tflow = horizon;
                            ts += horizon;
                            hits++;
                        }
                        propose = tnext < horizon;
                    } else if (kSticky && sticky_crosses((tnext < horizon) ? tnext : horizon)) {   // StickySamplingLoop.jl:50-60
                        const double t_togo = sticky_stick();             // move_to_axes_and_stick, :73-107
                        t = t + (t_togo + ts);
                        ts = 0.0;
                        event_out = true;
                    } else if (tnext > horizon) {                         // move_to_horizon!, :87-101
                        moves = true; tflow = horizon;
                        ts += horizon;
                        hits++;
                        if (adaptive) horizon = horizon * 1.01;
                    } else {
                        propose = tnext < horizon;                        // tp == horizon: retry (Q6)
                    }
                    mode = FRESH;
                    if (++it > cap) { done = true; if (g == 0) atomicAdd(&p.status[0], 1); }
                }
            }
            STAMP(0);
            if (!done && propose && kSticky && !(tp < tt)) {              // thaw_one_coordinate, StickySamplingLoop.jl:125-129,138-164
                sticky_thaw();
                t = t + tt;                                               // (the reference drops ts here, :160-161)
                ts = 0.0;
                propose = false;
                event_out = true;
                proposed = true;
            }
            if (!done && propose) {
                const double lam = true_rate(tp, raw);                    // ac_step!, :113-129
                ar = lam / lb;
                STAMP(1);
                propose = false;
                if (ar > 1.0) {
                    mode = ERR;
                } else {                                                  // ac_step_with_proxy!, :153-168
                    const double u = draw_uniform();
                    if (u < ar) {                                         // if_accept!, :170-186
                        moves = true; tflow = tp;
                        accepted = true;
                    } else {
                        mode = REUSE;                                     // if_reject! continues in the BOUND phase
                    }
                }
                proposed = true;
            }
            if (moves) flow_state(tflow);
            STAMP(2);
            if (accepted) refresh = velocity_jump(raw);
            if constexpr (SAMPLER == S_BPS || SAMPLER == S_BOOMERANG) {   // the whole wave draws the normals of its refreshing chains
                if (__ballot(refresh) != 0ull) refresh_velocities(refresh);
            } else if constexpr (SAMPLER == S_FECMC) {
                if (__ballot(refresh) != 0ull) fecmc_refresh(refresh);
            }
            STAMP(3);
            if (accepted || event_out) {
                if (accepted) {
                    t = t + tp + ts;
                    ts = 0.0;
                    tp = 0.0;
                }
                record(c, k, t, horizon, ar, errored, ev, rejected, hits);
                errored = 0; rejected = 0; hits = 0;                      // get_event_state!, :28-31
#pragma unroll
                for (int q = 0; q < 5; ++q) ev[q] = 0.0;
                it = 0;
                if (++k >= kend) done = true;
                if constexpr (!LEAN) { if (!(t < p.t_stop)) { done = true; stopped = true; } }   // the event that passed T is recorded, then the loop ends (:372-384)
            }
            STAMP(4);
            if (proposed && ++it > cap && !done) { done = true; if (g == 0) atomicAdd(&p.status[0], 1); }
        }
#ifdef PDMP_STAMP
        if (lane == 0 && p.stamps) for (int q = 0; q < 6; ++q) atomicAdd(&p.stamps[q], st_acc[q]);
#endif
        if (chain_ok) {                                                   // hand-off / final state: write-through (sc1) stores, see group_kernel_body
            PDMP_SLOTS(s) {
                const int i = s * G + g;
                if (in_dim(s, i)) { publish(st + i, X(s)); publish(st + d + i, V(s)); }
            }
            if (g == 0) {
                publish(st + 2 * d, t); publish(st + 2 * d + 1, horizon); publish(st + 2 * d + 2, ar);
                publish(st + 2 * d + 3, __longlong_as_double((long long)rng.n));
            }
            if constexpr (kSticky) publish(st + 2 * d + 4 + g, __longlong_as_double((long long)amask));
        }
        if (guard && g == 0) { atomicAdd(&p.status[1], guard); guard = 0; }
        if constexpr (TAPE) { if (tp_over && g == 0) atomicAdd(&p.status[3], 1); }
        // columns this chain recorded in this launch: written by the segment it stopped in, else by its last segment
        if constexpr (!LEAN) {
            if (p.count && chain_ok && g == 0 && !fin_before && (stopped || r == nseg - 1)) p.count[c] = (int64_t)k;
            if (p.count && chain_ok && g == 0 && fin_before && r == 0) p.count[c] = 1;  // T = 0: the initial point alone (:350-353)
        }
    }
};

// Persistent waves pull work units from one device-wide counter (price: one returning atomic per
// unit, MI355X_MICROARCH.md row "dequeue").  Unit (b, r) continues unit (b, r-1), pulled nb units
// earlier by a wave that is resident and waits only on still older units, so the wait below is
// satisfied on arrival in practice and can never deadlock; it is bounded all the same.
// Hand-off protocol (cdna_hip_programming.md Guideline 16, form R1): write-through (sc1) stores of the 2d+4 state words
// -> the storing wave's s_waitcnt vmcnt(0) -> relaxed agent flag store by the same wave;  relaxed poll -> agent-scope
// acquire (invalidates this CU's L1 only) -> plain loads.
// U = x'Px/2, grad-U = P x with a dense precision matrix (BASELINE config 4).  One chain per wave (G = 64);
// the gradient of a whole time point is a mat-vec: x_t is written to an LDS scratch tile, row i of P x_t is
// accumulated left to right over j (the order of the checker) by the lane that owns coordinate i, P streams
// column-major from L2 / Infinity Cache (coalesced 512-byte pieces).  d/dt of grad-U along a linear flow is
// P v, the same for every time point of a bound and until the next velocity change, so it is cached.
// First correct version: per-chain mat-vecs; the lock-step fp64-MFMA GEMM over chains is the planned upgrade.
template <class T> struct DenseSh { const double *g, *b; };
template <> struct GTarget<T_GAUSS_DENSE> {
    static constexpr bool kShared = true, kTile = true, kUser = false;
    static constexpr bool kLinear = false;     // (it is affine, but its slope lives in an LDS tile: keep the generic path)
    static constexpr int kAffineFrom = 1 << 30;
    static constexpr bool kMonotone = false;
    static constexpr int kExtraTiles = 4;          // two gradient tiles, P v, scratch x_t
    template <class T> using Sh = DenseSh<T>;
    const double *P;
    double *tg[2], *tb, *tx;
    bool b_valid;
    __device__ __forceinline__ void init(const double *, int) {}
    __device__ __forceinline__ double grad(int i, double, const DenseSh<double> &sh) const { return sh.g[i]; }
    __device__ __forceinline__ Dual grad(int i, Dual, const DenseSh<Dual> &sh) const { return {sh.g[i], sh.b[i]}; }
};

// Bounded wait for progress[b] >= r (relaxed agent-scope poll + s_sleep), then the agent-scope acquire.
__device__ __forceinline__ bool wait_progress(const int32_t *flag, int r) {
    bool ok = false;
    for (int spins = 0; spins < (1 << 22) && !ok; ++spins) {
        ok = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= r;
        if (!ok) __builtin_amdgcn_s_sleep(64);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return ok;
}

// Occupancy target (waves per SIMD) of the LDS-form group kernel = its register budget (512 / waves VGPRs).  NO instantiation may
// spill a VGPR: lanes of different chains run different branches, and a register spilled inside a divergent region and reloaded
// after the join comes back wrong in the lanes that were inactive there (seen twice on the box: an early register-form build
// faulted at address 0, and StickyZigZag with 2x-unrolled slot loops returned wrong positions until its 12-25 spills were
// gone).  The table below is set from the compiler's resource-usage remarks and the Makefile fails the build on any spill in
// any inst_*.hip, so a change that adds register pressure has to lower a number here instead of shipping a spill.
template <int SAMPLER, int TARGET, bool FD, bool RTGRID = false> struct GroupOcc {
    static constexpr int kWaves =
        (SAMPLER == S_SUZZ) ? ((TARGET == T_BANANA && !FD) ? 1 : 2)     // the flow's chain-level scalars and the speed duals on top of the Zig-Zag state
        : (TARGET == T_GAUSS_DENSE || SAMPLER == S_STICKY) ? 2          // the mat-vec's row accumulators; the active mask and the thaw clock
        : (SAMPLER == S_ZIGZAG && TARGET == T_BANANA && FD && !RTGRID) ? 3   // (round 5: it fits 168 registers exactly when told to -- left at 2 the allocator took 169 and the
                                                                        //  launch ran at two waves per SIMD, 96.6 -> 132 ms at config-2 size; round 4 had 167 by luck; the run-time-grid instantiation would spill 2)
        : (TARGET == T_BANANA || TARGET == T_USER) ? 2                  // banana + finite differences keeps six shared records
        : (SAMPLER == S_RHMC) ? 4
        : 3;                                                            // Zig-Zag, Boomerang, ForwardECMC on the separable Gaussians
};

template <int SAMPLER, int TARGET, bool FD, int KC = 0, int GC = 0, bool TAPE = false, bool LEAN = false, bool RTGRID = false>
__device__ __forceinline__ void group_kernel_body(const GroupParams &gp) {
    extern __shared__ double smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    // target parameters (mu, prec) once per block
    for (int q = threadIdx.x; q < gp.tpar; q += blockDim.x) smem[q] = gp.s.tparams[q];
    __syncthreads();
    const size_t tsz = (size_t)gp.K * WAVE;
    double *tile = smem + gp.tpar + (size_t)wave * gp.tiles * tsz;
    const int cpw = WAVE / gp.G;
    // register form: no x tile; one K*64-double tile per wave only where refreshed velocities are exchanged between lanes
    GroupChain<SAMPLER, TARGET, FD, KC, GC, TAPE, LEAN, RTGRID> chain(gp, lane, tile, KC > 0 ? tile : tile + tsz);
    chain.tgt.init(smem, gp.s.d);
    if constexpr (TARGET == T_GAUSS_DIAG) chain.tgt.nonneg = (gp.s.tflags & 1) != 0;
    if constexpr (SAMPLER == S_FECMC) {                              // (register form: the exchange tile first, then the four jump vectors)
        for (int q = 0; q < 4; ++q) chain.fe_t[q] = tile + ((KC > 0 ? 1 : 2) + q) * tsz;
    }
    if constexpr (SAMPLER == S_STICKY) chain.kap = smem + (gp.tpar - gp.s.d);   // kappa[d]: the last d staged parameters
    if constexpr (GTarget<TARGET>::kTile) {
        chain.tgt.P = gp.s.tparams;
        chain.tgt.tg[0] = tile + 2 * tsz; chain.tgt.tg[1] = tile + 3 * tsz;
        chain.tgt.tb = tile + 4 * tsz; chain.tgt.tx = tile + 5 * tsz;
        chain.tgt.b_valid = false;
    }
    const int nb = (int)gp.nb, n_units = nb * gp.nseg;            // host guarantees nb * nseg < 2^31
    bool lost = false;
    for (int pulled = 0; pulled <= n_units && !lost; ++pulled) {  // a wave can never pull more than n_units units
        int u = 0;
        if (lane == 0) u = atomicAdd(gp.ctr, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= n_units) break;
        const int r = u / nb, b = u - r * nb;
        if (r > 0) lost = !wait_progress(gp.progress + b, r);
        if (!lost) {
            const int64_t c = (int64_t)b * cpw + lane / gp.G;
            chain.run_segment(c, c < gp.s.C, r, gp.seg_len, gp.state, gp.nseg);
            // no agent-scope release here: it is a buffer_wbl2, a write-back of the XCD's whole L2, which the event stores of
            // every wave keep full of dirty lines (measured: 35 -> 52 ms when the segments were made 3x shorter); the
            // hand-off state went out write-through instead, and the drain below covers it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&gp.progress[b], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lost && lane == 0) atomicAdd(&gp.s.status[2], 1);         // predecessor lost: report, never hang
}

// A launch that another kernel form serves (the lane kernel was enqueued and its unit-speed flag stayed 0): EVERY entry point of
// the group family returns at once, so that two forms can never both write the outputs and the shared scheduler / hand-off state.
__device__ __forceinline__ bool served_elsewhere(const GroupParams &gp) {
    return gp.only_if && __hip_atomic_load(gp.only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

template <int SAMPLER, int TARGET, bool FD, bool RTGRID = false>
__global__ __launch_bounds__(256, (GroupOcc<SAMPLER, TARGET, FD, RTGRID>::kWaves)) void group_kernel(const GroupParams gp) {
    if (served_elsewhere(gp)) return;
    group_kernel_body<SAMPLER, TARGET, FD, 0, 0, false, false, RTGRID>(gp);
}
// tape replay (inst_tape_*.hip): the same code with the tapes as random source; one wave per SIMD (512 registers: the tape
// pointers and positions can never cause a spill, and a replay is a parity run, not a throughput run)
template <int SAMPLER, int TARGET, bool FD, bool RTGRID = false>
__global__ __launch_bounds__(256, 1) void group_kernel_tape(const GroupParams gp) {
    if (served_elsewhere(gp)) return;
    group_kernel_body<SAMPLER, TARGET, FD, 0, 0, true, false, RTGRID>(gp);
}
template <int SAMPLER, int TARGET, bool FD, int KC, int GC, bool FULL>
__global__ __launch_bounds__(256, 1) void group_kernel_reg_tape(const GroupParams gp_in) {
    if (served_elsewhere(gp_in)) return;
    GroupParams gp = gp_in;
    gp.G = GC;
    if (FULL) gp.K = KC;
    group_kernel_body<SAMPLER, TARGET, FD, KC, GC, true>(gp);
}
// hand-off words per chain in the launch's state buffer: 2 d + 4 (x, v, t, horizon, ar, stream position); the lane kernel's pair form
// keeps d + 6 (lane_kernel.hpp: two sign words) -- more than that at d = 1
__host__ __device__ constexpr int lane_state_words(int d) { return 2 * d + 4 < 8 ? 8 : 2 * d + 4; }

// Which register-form kernels exist.  kAny: the form is used at all; kPartial: also for K < KC (with the per-slot skips);
// full<GC>(): a separate instantiation without the skips for K == KC.  Set from the compiler's resource-usage remarks so that no
// instantiation spills a VGPR (see below), and from measurements (Zig-Zag with finite differences: the LDS form is faster).
template <int SAMPLER, int TARGET, bool FD> struct RegForm {
    // Zig-Zag with finite-difference slopes (the default AD_backend of the manual-gradient constructor, ZigZagSamplers.jl:58-60): rounds 1-4
    // kept it in the LDS form (13-slot register kernels spill 2 .. 59 registers with the six evaluations per coordinate and cell; seven slots at
    // G = 8 were measured slower).  Round 5: the LEAN FULL 13-slot
    // kernels on the Gaussians -- no diagnostic arrays, no time horizon -- are free of spills: exactly those are built (kLeanOnly); a launch
    // that records diagnostics, the banana and every other slot count keep the LDS form.
    static constexpr bool kZzFd = SAMPLER == S_ZIGZAG && FD;
    static constexpr bool kLeanOnly = kZzFd;
    static constexpr bool kSlowZz = kZzFd && TARGET == T_BANANA;
    // RHMC (round 4): the FULL 13-slot kernels are free of spills on the isotropic Gaussian and the banana, the ones with the per-slot
    // skips on the diagonal Gaussian (FULL there spills 2 - 4 registers with the diagnostic arrays compiled in); the slope mode plays no
    // part in RHMC (constant refresh clock): only FD = false is instantiated.  SpeedUpZigZag spills 110 - 700 registers in every variant
    // at two waves per SIMD (its flow and effective gradient on forward-mode numbers) and still 30 -- into AGPRs -- at one wave per SIMD
    // (tried in round 4; a spill is not allowed here, see below) -- with THIRTEEN slots; exact-K kernels fit for K <= 7 (isotropic Gaussian),
    // 6 (diagonal), 4 (banana) and ship since the end of round 4 (inst_reg_suzz_k*.hip; forward-mode slopes).  ForwardECMC (round 4): x, v in registers, the jump's four
    // d-vectors stay in LDS tiles (five tiles per wave with the exchange tile), exact-K kernels for K <= 7 only (inst_reg_fecmc_k*.hip: 160 - 253
    // VGPRs, no spill, two waves per SIMD; forward-mode slopes) -- d = 50: 50.4 -> 27.5 ms (iso), 72.3 -> 28.7 ms (banana) against the LDS form's best layout.
    static constexpr bool kRhmc = SAMPLER == S_RHMC;
    static constexpr bool kPartial = kRhmc ? (TARGET == T_GAUSS_DIAG)
                                           : (!kZzFd && !(SAMPLER == S_BOOMERANG && TARGET != T_GAUSS_DIAG));   // would spill: Boomerang iso / banana with the skips
    template <int GC> static constexpr bool full() {        // would spill: Zig-Zag and (since the time-horizon stop of round 3) Boomerang on the diagonal Gaussian at one lane per chain
        if constexpr (kRhmc) return !(TARGET == T_GAUSS_DIAG && GC >= 2);
        return !kSlowZz && !((SAMPLER == S_ZIGZAG || SAMPLER == S_BOOMERANG) && TARGET == T_GAUSS_DIAG && GC == 1);
    }
    static constexpr bool kFecmc = SAMPLER == S_FECMC;
    static constexpr bool kAny = !kSlowZz && !(kFecmc && FD) && !(SAMPLER == S_SUZZ && FD) && !(SAMPLER == S_STICKY && FD) && !(kRhmc && FD);   // (StickyZigZag, round 4: exact-K kernels, K <= 7 -- 13 slots spill; inst_reg_sticky_k*.hip)
};

// register form (x, v in VGPRs, up to KC slots per lane, G fixed at compile time): no LDS tiles; 2 waves per SIMD -- the
// instantiations must stay free of VGPR spills (a spill under a partial exec mask loses the inactive lanes' values, and
// lanes of different chains run different branches here): `make check-spills`
template <int SAMPLER, int TARGET, bool FD, int KC, int GC, bool FULL, bool LEAN = false>
#ifndef PDMP_REG_WAVES
#define PDMP_REG_WAVES 2
#endif
__global__ __launch_bounds__(256, PDMP_REG_WAVES) void group_kernel_reg(const GroupParams gp_in) {
    if (gp_in.only_if && __hip_atomic_load(gp_in.only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    GroupParams gp = gp_in;
    gp.G = GC;                 // the host launches this instantiation only for G == GC, K <= KC: folds every use of G
    if (FULL) gp.K = KC;       // ... and the FULL one only for K == KC: the per-slot skips fold away too
    group_kernel_body<SAMPLER, TARGET, FD, KC, GC, false, LEAN>(gp);
}

#ifndef __HIPCC_RTC__   // host side (not part of a runtime-compiled user kernel)
// waves per block: as many (up to 4) as fit the CU's 160 KiB of LDS next to the staged parameters
inline int group_waves_per_block(int K, int tpar, int tiles = 2) {
    const size_t tile = (size_t)tiles * K * WAVE * sizeof(double), avail = 160 * 1024 - (size_t)tpar * sizeof(double);
    const size_t w = avail / tile;
    return w >= 4 ? 4 : (int)w;
}

// Segments per chain: enough units for ~32 rounds over the resident wave slots (tail <= ~3 %), at least
// 16 events per segment (hand-off = 2d+4 doubles per chain), one segment when the machine is not full.
inline void group_schedule(int64_t nb, int64_t slots, int64_t n_sk, int force, int32_t &nseg, int64_t &seg_len) {
    const int64_t events = n_sk > 1 ? n_sk - 1 : 1;
    int64_t want = nb > slots ? (32 * slots + nb - 1) / nb : 1;
    if (force > 0) {                       // test hook (PDMP_FORCE_SEGMENTS): exercise the hand-off on small ensembles
        want = force < events ? force : events;
        seg_len = (events + want - 1) / want;
        nseg = (int32_t)((events + seg_len - 1) / seg_len);
        return;
    }
    const int64_t max_seg = events / 16 > 0 ? events / 16 : 1;
    if (want > max_seg) want = max_seg;
    if (want < 1) want = 1;
    seg_len = (events + want - 1) / want;
    nseg = (int32_t)((events + seg_len - 1) / seg_len);
}

template <int SAMPLER, int TARGET>
hipError_t launch_group(const GroupParams &gp, hipStream_t stream);

template <int SAMPLER, int TARGET>
hipError_t launch_group_tape(const GroupParams &gp, hipStream_t stream);       // LDS form with the tapes as random source (inst_tape_lds*.hip)

// EXACT: only the FULL variant (K == KC) of this KC is instantiated (inst_reg_*_k.hip: one kernel per K = 7..12)
template <int SAMPLER, int TARGET, bool FD, int KC = 0, int GC = 0, bool EXACT = false, bool TAPE = false>
hipError_t launch_group_impl(const GroupParams &gp_in, hipStream_t stream) {
    GroupParams gp = gp_in;
    if (KC > 0) gp.tiles = (SAMPLER == S_ZIGZAG) ? 0 : (SAMPLER == S_FECMC ? 5 : 1);      // (ForwardECMC: the exchange tile + n, v_orth, e1, e2)
    const int cpw = WAVE / gp.G, wpb = KC > 0 ? 4 : group_waves_per_block(gp.K, gp.tpar, gp.tiles);
    if (wpb < 1) return hipErrorInvalidValue;
    const size_t lds = ((size_t)gp.tpar + (size_t)wpb * gp.tiles * gp.K * WAVE) * sizeof(double);
    void (*kernel)(const GroupParams) = nullptr;
    if constexpr (KC > 0) {
        using RF = RegForm<SAMPLER, TARGET, FD>;
        if constexpr (TAPE) {                       // (one wave per SIMD: no instantiation can spill, so every one exists)
            if constexpr (!EXACT) kernel = group_kernel_reg_tape<SAMPLER, TARGET, FD, KC, GC, false>;
            if (gp.K == KC) kernel = group_kernel_reg_tape<SAMPLER, TARGET, FD, KC, GC, true>;
        } else {
            if constexpr (RF::kPartial && !EXACT) kernel = group_kernel_reg<SAMPLER, TARGET, FD, KC, GC, false>;
            if constexpr (RF::template full<GC>() && !RF::kLeanOnly) if (gp.K == KC) kernel = group_kernel_reg<SAMPLER, TARGET, FD, KC, GC, true>;
            if constexpr (RF::template full<GC>() && KC == KREG) {             // the throughput kernels of the BASELINE shapes (d = 50 / G = 4, d = 100 / G = 8, ...)
                const SkelParams &q = gp.s;
                const bool plain = !(q.H || q.AR || q.EB || q.RJ || q.HH || q.EV || q.IA || q.count || q.chain_map || q.resume) && !(q.t_stop < __builtin_huge_val());
                if (gp.K == KC && plain && !(gp.host_flags & 1)) kernel = group_kernel_reg<SAMPLER, TARGET, FD, KC, GC, true, true>;
            }
        }
    } else {
        // ONE instantiation per (sampler, target, slope mode) takes grid_size at run time -- measured equal to a constant grid (StickyZigZag 35.5 / 35.6 ms,
        // SpeedUpZigZag 65.8 / 66.2) -- except where the constant buys a wave per SIMD (GroupOcc: Zig-Zag, banana, finite-difference slopes): there the
        // default grid (and Brent, which has no grid) keeps its own kernel
        constexpr bool kFixedToo = GroupOcc<SAMPLER, TARGET, FD, false>::kWaves != GroupOcc<SAMPLER, TARGET, FD, true>::kWaves;
        const bool fixed = kFixedToo && (gp.s.n_grid == GNG || gp.s.n_grid == 0);
        if constexpr (TAPE) kernel = group_kernel_tape<SAMPLER, TARGET, FD, true>;
        else {
            kernel = group_kernel<SAMPLER, TARGET, FD, true>;
            if constexpr (kFixedToo) { if (fixed) kernel = group_kernel<SAMPLER, TARGET, FD, false>; }
        }
    }
    if (!kernel) return hipErrorNotSupported;
    hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int per_cu = 0, dev = 0, cus = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE * wpb, lds);
    if (e != hipSuccess) return e;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (per_cu < 1) per_cu = 1;
    if (cus < 1) cus = 256;
    gp.nb = (gp.s.C + cpw - 1) / cpw;
    const int64_t slots = (int64_t)cus * per_cu * wpb;
    group_schedule(gp.nb, slots, gp.s.n_sk, gp.nseg, gp.nseg, gp.seg_len);
    int64_t blocks = (gp.nb + wpb - 1) / wpb;
    if (blocks > (int64_t)cus * per_cu) blocks = (int64_t)cus * per_cu;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(WAVE * wpb), lds, stream, gp);
    return hipGetLastError();
}

#define PDMP_DEFINE_GROUP_TAPE_LAUNCHER(S, T)                                                       \
    template <>                                                                                     \
    hipError_t launch_group_tape<S, T>(const GroupParams &gp, hipStream_t stream) {                 \
        return gp.s.deriv_mode != 0 ? launch_group_impl<S, T, true, 0, 0, false, true>(gp, stream)  \
                                    : launch_group_impl<S, T, false, 0, 0, false, true>(gp, stream); \
    }

#define PDMP_DEFINE_GROUP_LAUNCHER(S, T)                                                            \
    template <>                                                                                     \
    hipError_t launch_group<S, T>(const GroupParams &gp, hipStream_t stream) {                      \
        return gp.s.deriv_mode != 0 ? launch_group_impl<S, T, true>(gp, stream)                     \
                                    : launch_group_impl<S, T, false>(gp, stream);                   \
    }

#endif  // __HIPCC_RTC__

PDMP_NS_END
