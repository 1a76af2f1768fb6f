// skeleton_kernel.hpp -- the hot path: one wavefront per chain runs the whole thinning state machine
// of PDMPFlux.jl's sample_skeleton() (src/sample.jl:253-284) with the chain state in registers.
//
//   lane l owns coordinates l, l+64, ... of x and v (K = ceil(d/64) registers each);
//   wave-uniform scalars (t, horizon, tp, ...) live replicated in VGPRs / SGPRs;
//   the BoundBox (src/Composites.jl:15-20) is distributed over lanes: lane j holds box_max[j] and
//   cum_sum[j] (grid_size <= 64), so next_event is one ballot + popcount + three readlanes;
//   sums over coordinates are fixed-order wave reductions (wave_math.hpp);
//   every event is two coalesced d*8-byte stores (x, v) straight into the PDMPHistory layout.
//
// Citations: file:line under the reference tree.  The control flow follows
// src/SamplingLoopInplace.jl:27-217 statement by statement; quirks Q1-Q6 of SURVEY.md 3.7 are kept.
#pragma once
#include "targets.hpp"

PDMP_NS_BEGIN

enum : int { S_ZIGZAG = 0, S_BPS = 1, S_BOOMERANG = 2, S_FECMC = 3, S_SUZZ = 4, S_RHMC = 5, S_STICKY = 6 };   // ids >= 3: group kernel only

struct SkelParams {
    int32_t d, n_grid, signed_bound, adaptive, gaussian_velocity, deriv_mode, vectorized;
    int64_t C, n_sk, chain0, cap;
    uint64_t seed;
    double tmax, refresh;
    const double *xinit, *vinit, *tparams;
    double *X, *V, *t, *H, *AR, *EV;
    int32_t *EB, *RJ, *HH;
    uint8_t *IA;       // StickyZigZag: PDMPHistory.is_active, d x n_sk x C bytes, or null
    int32_t *status;   // [0] chains that hit the iteration cap, [1] Q11 guard events, [2] lost hand-offs, [3] chains whose tape ran out
    unsigned long long *stamps;   // diagnostic build only (-DPDMP_STAMP): per-phase cycle sums, else null
    // tape source: replay pre-recorded variates instead of Philox; chain-major, indexed by global chain id
    const double *tape_exp, *tape_uni, *tape_nrm;
    int64_t tape_n_exp, tape_n_uni, tape_n_nrm;
    // ForwardECMC keywords (ForwardEventChainMonteCarlo.jl:290-292)
    double fe_mix_p;
    int32_t fe_ran_p, fe_switch, fe_positive;
    // RHMC keywords (RandomizedHamiltonianMonteCarlo.jl:48-57)
    double rh_step_size, rh_phi;
    double fe_speed;   // ForwardECMC speed_factor
    // Time-horizon variant sample_skeleton(sampler, T::Float64, ...) (src/sample.jl:323-439) as a kernel feature (group / lane kernels):
    //   t_stop   a chain stops after recording its first event with t >= t_stop (`while state.t < T`, :372); +Inf: fixed count
    //   resume   != 0: every chain starts from its hand-off state (x, v, t, horizon, ar, stream position: the last event of the
    //            previous chunk, recorded again as column 0 of this one) instead of (xinit, vinit, 0, tmax, 0, 0)
    //   chain_map  null, or chain_map[c] = index of launch chain c in the ORIGINAL ensemble: its state slot, its xinit / vinit row
    //            and (+ chain0) its random stream; outputs are indexed by the launch's own c (dense over the active chains)
    //   count    null, or count[c] = columns chain c recorded in this launch (n_sk when it used the whole capacity)
    double t_stop;
    int32_t resume;
    int32_t tflags;    // target facts found by the host when the handle is made: bit 0 = every precision of the diagonal Gaussian is >= 0 (GTarget<T_GAUSS_DIAG>::nonneg)
    const int64_t *chain_map;
    int64_t *count;
};

__device__ __forceinline__ bool uni(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }

struct Bound {       // lane-distributed BoundBox
    double box;      // lane j: box_max[j]   (j < n-1)
    double cum;      // lane j: cum_sum[j]   (j < n)
    double h;        // step_size (uniform)
};

template <int SAMPLER, int TARGET, int K>
struct Chain {
    const SkelParams &p;
    const int lane;
    Target<TARGET, K> tgt;
    ChainRng rng;
    double x[K], v[K];
    bool act[K];
    int32_t guard = 0;
    const double *te = nullptr, *tu = nullptr, *tn = nullptr;   // this chain's tapes (null: Philox)
    int64_t ie = 0, iu = 0, in = 0;
    bool overrun = false;

    // random variates: the chain's Philox stream, or its tapes when a tape is set (an exhausted tape yields
    // 1.0 / 0.5 / 0.0 and flags the launch)
    __device__ __forceinline__ double r_exp() {
        if (te) { if (ie < p.tape_n_exp) return te[ie++]; overrun = true; return 1.0; }
        return rng.exponential();
    }
    __device__ __forceinline__ double r_uni() {
        if (tu) { if (iu < p.tape_n_uni) return tu[iu++]; overrun = true; return 0.5; }
        return rng.uniform();
    }
    __device__ __forceinline__ double r_normal(int i) {          // coordinate i of a d-vector of normals
        if (tn) { if (in + i < p.tape_n_nrm) return tn[in + i]; overrun = true; return 0.0; }
        return rng.normal_coord(i);
    }
    __device__ __forceinline__ void r_normals_done(int d) {
        if (tn) { in += d; return; }
        rng.normals_done(d);
    }

    __device__ __forceinline__ Chain(const SkelParams &pp, int l) : p(pp), lane(l) {}

    // ---- flow with dual time (t,1): ZigZagSamplers.jl:80 / BouncyParticleSamplers.jl:32 /
    //      BoomerangSamplers.jl:31
    __device__ __forceinline__ void flow_dual(double t, Dual (&xt)[K], Dual (&vt)[K]) const {
        if constexpr (SAMPLER == S_BOOMERANG) {
            double s, c;
            det_sincos(t, s, c);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                xt[k] = {x[k] * c + v[k] * s, x[k] * (-s) + v[k] * c};
                vt[k] = {(-x[k]) * s + v[k] * c, (-x[k]) * c + v[k] * (-s)};
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                xt[k] = {x[k] + v[k] * t, v[k]};
                vt[k] = {v[k], 0.0};
            }
        }
    }
    __device__ __forceinline__ void flow_plain(double t, double (&xt)[K], double (&vt)[K]) const {
        if constexpr (SAMPLER == S_BOOMERANG) {
            double s, c;
            det_sincos(t, s, c);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                xt[k] = x[k] * c + v[k] * s;
                vt[k] = (-x[k]) * s + v[k] * c;
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) { xt[k] = x[k] + v[k] * t; vt[k] = v[k]; }
        }
    }
    __device__ __forceinline__ double lane_sum(const double (&a)[K]) const {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += act[k] ? a[k] : 0.0;
        return s;
    }

    // ---- bounded function, vector form: signed_rate_vect ZigZagSamplers.jl:95-98 / rate_vect :88-91
    template <bool SLOPE>
    __device__ __forceinline__ void func_vect(double t, double (&val)[K], double (&slp)[K]) const {
        if constexpr (SLOPE) {
            Dual xt[K], vt[K], g[K];
            flow_dual(t, xt, vt);
            tgt.grad(xt, g, lane);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double s = g[k].v * vt[k].v;
                double sd;
                if constexpr (SAMPLER == S_BOOMERANG) sd = g[k].e * vt[k].v + g[k].v * vt[k].e;
                else sd = g[k].e * vt[k].v;
                if (p.signed_bound) { val[k] = s; slp[k] = sd; }
                else { val[k] = pmax(0.0, s); slp[k] = (0.0 < s) ? sd : 0.0; }
            }
        } else {
            double xt[K], vt[K], g[K];
            flow_plain(t, xt, vt);
            tgt.grad(xt, g, lane);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double s = g[k] * vt[k];
                val[k] = p.signed_bound ? s : pmax(0.0, s);
            }
        }
    }
    // ---- bounded function, scalar form: BouncyParticleSamplers.jl:39-47, BoomerangSamplers.jl:38-46
    template <bool SLOPE>
    __device__ __forceinline__ double func_scalar(double t, double &slp) const {
        double dv, de = 0.0;
        if constexpr (SLOPE) {
            Dual xt[K], vt[K], g[K];
            flow_dual(t, xt, vt);
            tgt.grad(xt, g, lane);
            double a[K], b[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                a[k] = g[k].v * vt[k].v;
                if constexpr (SAMPLER == S_BOOMERANG) b[k] = g[k].e * vt[k].v + g[k].v * vt[k].e;
                else b[k] = g[k].e * vt[k].v;
                if constexpr (SAMPLER == S_ZIGZAG) {      // non-vectorized Zig-Zag bounds rate = sum(max.(0,.)), ZigZagSamplers.jl:83-86
                    b[k] = (0.0 < a[k]) ? b[k] : 0.0;
                    a[k] = pmax(0.0, a[k]);
                }
            }
            dv = wave_sum(lane_sum(a));
            de = wave_sum(lane_sum(b));
        } else {
            double xt[K], vt[K], g[K], a[K];
            flow_plain(t, xt, vt);
            tgt.grad(xt, g, lane);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                a[k] = g[k] * vt[k];
                if constexpr (SAMPLER == S_ZIGZAG) a[k] = pmax(0.0, a[k]);
            }
            dv = wave_sum(lane_sum(a));
        }
        if constexpr (SAMPLER == S_ZIGZAG) { slp = de; return dv; }
        if (p.signed_bound) { slp = de; return dv + p.refresh; }
        slp = (0.0 < dv) ? de : 0.0;
        return pmax(0.0, dv) + p.refresh;
    }

    // finite_difference_derivative sample points, UpperBound.jl:57-61 (start = 0)
    __device__ __forceinline__ static void fd_points(double t, double horizon, double &tm, double &tp) {
        const double h = 1.4901161193847656e-08 * pmax(1.0, fabs(t));
        tm = pmax(0.0, t - h);
        tp = (horizon < t + h) ? horizon : t + h;
    }

    // ---- unsigned total rate for the accept test (always `sampler.rate`, SamplingLoopInplace.jl:116)
    __device__ __forceinline__ double true_rate(double t) const {
        double xt[K], vt[K], g[K], a[K];
        flow_plain(t, xt, vt);
        tgt.grad(xt, g, lane);
        if constexpr (SAMPLER == S_ZIGZAG) {                  // ZigZagSamplers.jl:83-86
#pragma unroll
            for (int k = 0; k < K; ++k) a[k] = pmax(0.0, g[k] * vt[k]);
            return wave_sum(lane_sum(a));
        } else {                                              // BPS :39-42, Boomerang :38-41
#pragma unroll
            for (int k = 0; k < K; ++k) a[k] = g[k] * vt[k];
            return pmax(0.0, wave_sum(lane_sum(a))) + p.refresh;
        }
    }

    // ---- upper_bound_grid_vect (UpperBound.jl:203-247) / upper_bound_grid (:92-137)
    __device__ __forceinline__ Bound upper_bound(double horizon) const {
        const int n = p.n_grid;
        Bound B;
        B.h = horizon / (double)(n - 1);
        B.box = 0.0;
        B.cum = 0.0;
        const double h = B.h;
        double acc = 0.0;
        if (p.vectorized) {
            double v0[K], g0[K], v1[K], g1[K];
            eval_vect(0.0, horizon, v0, g0);
            for (int j = 0; j + 1 < n; ++j) {
                const double t0 = (double)j * h;
                const double t1 = (j + 2 == n) ? horizon : (double)(j + 1) * h;
                eval_vect(t1, horizon, v1, g1);
                double cell[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    // :229 absolute-time tangent intersection, :234 NaN -> 0, :235 clamp (Q1)
                    double q = (((v0[k] - v1[k]) + g1[k] * t1) - g0[k] * t0) / (g1[k] - g0[k]);
                    if (q != q) q = 0.0;
                    q = (q > h) ? h : ((q < 0.0) ? 0.0 : q);
                    const double inter = v0[k] + g0[k] * q;                            // :237
                    cell[k] = pmax(pmax(pmax(v0[k], v1[k]), inter), 0.0);              // :239-241
                    v0[k] = v1[k];
                    g0[k] = g1[k];
                }
                const double bj = wave_sum(lane_sum(cell));                            // :246
                acc += bj;
                if (lane == j) B.box = bj;
                if (lane == j + 1) B.cum = acc * h;                                    // :243-244
            }
        } else {
            const double refresh = p.signed_bound ? p.refresh : 0.0;                   // AbstractPDMP.jl:104-112
            double f0, s0, f1, s1;
            eval_scalar(0.0, horizon, f0, s0);
            for (int j = 0; j + 1 < n; ++j) {
                const double t1 = (j + 2 == n) ? horizon : (double)(j + 1) * h;
                eval_scalar(t1, horizon, f1, s1);
                double q = ((f0 - f1) + s1 * h) / (s1 - s0);                           // :123 relative form
                if (q != q) q = 0.0;                                                   // :124
                q = (q > h) ? h : ((q < 0.0) ? 0.0 : q);                               // :125
                const double inter = f0 + s0 * q;                                      // :127
                double bj = pmax(pmax(pmax(f0, f1), inter), 0.0);                      // :128-130
                bj += refresh;                                                         // :131 (Q2)
                acc += bj;
                if (lane == j) B.box = bj;
                if (lane == j + 1) B.cum = acc * h;                                    // :133-134
                f0 = f1;
                s0 = s1;
            }
        }
        return B;
    }
    // value + slope at one grid time, forward-mode (UpperBound.jl:208-210) or finite differences (:223-224)
    __device__ __forceinline__ void eval_vect(double t, double horizon, double (&val)[K], double (&slp)[K]) const {
        if (p.deriv_mode == 0) { func_vect<true>(t, val, slp); return; }
        double fa[K], fb[K], dummy[K];
        func_vect<false>(t, val, dummy);                          // fx, :56
        double tm, tp;
        fd_points(t, horizon, tm, tp);
        if (tp == tm) {
#pragma unroll
            for (int k = 0; k < K; ++k) slp[k] = val[k] - val[k];
        } else if (tm == t) {
            func_vect<false>(tp, fa, dummy);
#pragma unroll
            for (int k = 0; k < K; ++k) slp[k] = (fa[k] - val[k]) / (tp - t);
        } else if (tp == t) {
            func_vect<false>(tm, fb, dummy);
#pragma unroll
            for (int k = 0; k < K; ++k) slp[k] = (val[k] - fb[k]) / (t - tm);
        } else {
            func_vect<false>(tp, fa, dummy);
            func_vect<false>(tm, fb, dummy);
#pragma unroll
            for (int k = 0; k < K; ++k) slp[k] = (fa[k] - fb[k]) / (tp - tm);
        }
    }
    __device__ __forceinline__ void eval_scalar(double t, double horizon, double &val, double &slp) const {
        if (p.deriv_mode == 0) { val = func_scalar<true>(t, slp); return; }
        double dummy;
        val = func_scalar<false>(t, dummy);
        double tm, tp;
        fd_points(t, horizon, tm, tp);
        if (tp == tm) slp = val - val;
        else if (tm == t) slp = (func_scalar<false>(tp, dummy) - val) / (tp - t);
        else if (tp == t) slp = (val - func_scalar<false>(tm, dummy)) / (t - tm);
        else {
            const double fp = func_scalar<false>(tp, dummy), fm = func_scalar<false>(tm, dummy);
            slp = (fp - fm) / (tp - tm);
        }
    }

    // ---- next_event, UpperBound.jl:264-273
    __device__ __forceinline__ void next_event(const Bound &B, double E, double &tp, double &lb) const {
        const int n = p.n_grid;
        const unsigned long long m = __ballot((lane < n) && (B.cum < E));
        int idx = __builtin_amdgcn_readfirstlane(__popcll(m));     // searchsortedfirst (0-based)
        if (idx >= n) {
            tp = __builtin_huge_val();
            lb = readlane_d(B.box, n - 2);
            return;
        }
        if (idx == 0) idx = 1;
        const double c0 = readlane_d(B.cum, idx - 1), c1 = readlane_d(B.cum, idx);
        tp = (double)(idx - 1) * B.h + (E - c0) / (c1 - c0) * B.h;
        lb = readlane_d(B.box, idx - 1);
    }

    __device__ __forceinline__ void flow_state(double t) {
        double xt[K], vt[K];
        flow_plain(t, xt, vt);
#pragma unroll
        for (int k = 0; k < K; ++k) { x[k] = xt[k]; v[k] = vt[k]; }
    }

    // ---- velocity_jump: ZigZagSamplers.jl:101-107, BouncyParticleSamplers.jl:50-74, BoomerangSamplers.jl:49-67
    __device__ __forceinline__ void velocity_jump() {
        double g[K];
        tgt.grad(x, g, lane);
        if constexpr (SAMPLER == S_ZIGZAG) {
            double lam[K];
#pragma unroll
            for (int k = 0; k < K; ++k) lam[k] = act[k] ? pmax(0.0, g[k] * v[k]) : 0.0;
            const double S = wave_sum(lane_sum(lam));
            const double u = r_uni();                                // Categorical draw
            if (!uni(S > 0.0)) { guard++; return; }                  // Q11
            double carry = 0.0;
            int m = -1;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double inc = wave_scan_incl(lam[k] / S, lane);
                const unsigned long long hit = __ballot(act[k] && (carry + inc > u));
                if (m < 0 && hit != 0ull) m = WAVE * k + (int)__builtin_ctzll(hit);
                carry = carry + readlane_d(inc, WAVE - 1);
            }
            if (m < 0) m = p.d - 1;
            m = __builtin_amdgcn_readfirstlane(m);
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (lane + WAVE * k == m) v[k] = -v[k];              // v[m] *= -1
            return;
        } else {
            if constexpr (SAMPLER == S_BOOMERANG) {
#pragma unroll
                for (int k = 0; k < K; ++k) g[k] = g[k] - x[k];      // grad-U_effective, :51-52 (Q8)
            }
            double a[K];
#pragma unroll
            for (int k = 0; k < K; ++k) a[k] = g[k] * v[k];
            const double gv = wave_sum(lane_sum(a));
            const double bounce = pmax(0.0, gv);
            const double prob = bounce / (bounce + p.refresh);       // NaN if both 0 (Q9)
            const double u = r_uni();
            if (uni(u < prob)) {
#pragma unroll
                for (int k = 0; k < K; ++k) a[k] = g[k] * g[k];
                const double gg = wave_sum(lane_sum(a));
                if constexpr (SAMPLER == S_BPS) {
                    if (uni(gg == 0.0)) return;
                    const double scale = 2.0 * gv / gg;              // :60
#pragma unroll
                    for (int k = 0; k < K; ++k) v[k] = v[k] - scale * g[k];
                } else {
                    const double nrm = __dsqrt_rn(gg);               // :58
                    double e[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) { e[k] = g[k] / nrm; a[k] = v[k] * e[k]; }
                    const double ve = 2.0 * wave_sum(lane_sum(a));
#pragma unroll
                    for (int k = 0; k < K; ++k) v[k] = v[k] - ve * e[k];   // :59
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) v[k] = act[k] ? r_normal(lane + WAVE * k) : 0.0;
                r_normals_done(p.d);
                if (SAMPLER == S_BPS && !p.gaussian_velocity) {      // normalize!, :71
#pragma unroll
                    for (int k = 0; k < K; ++k) a[k] = v[k] * v[k];
                    const double inv = 1.0 / __dsqrt_rn(wave_sum(lane_sum(a)));
#pragma unroll
                    for (int k = 0; k < K; ++k) v[k] = v[k] * inv;
                }
            }
        }
    }

    // ---- record!, Composites.jl:239-260
    __device__ __forceinline__ void record(int64_t c, int64_t k, double t, double horizon, double ar,
                                           int32_t errored, const double (&ev)[5], int32_t rejected,
                                           int32_t hits) const {
        const size_t e = (size_t)c * (size_t)p.n_sk + (size_t)k;
        double *Xe = p.X + e * (size_t)p.d, *Ve = p.V + e * (size_t)p.d;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int i = lane + WAVE * kk;
            if (act[kk]) { Xe[i] = x[kk]; Ve[i] = v[kk]; }
        }
        if (lane == 0) {
            p.t[e] = t;
            if (p.H) p.H[e] = horizon;
            if (p.AR) p.AR[e] = ar;
            if (p.EB) p.EB[e] = errored;
            if (p.RJ) p.RJ[e] = rejected;
            if (p.HH) p.HH[e] = hits;
        }
        if (p.EV && lane < 5) {
            const double val = lane == 0 ? ev[0] : lane == 1 ? ev[1] : lane == 2 ? ev[2] : lane == 3 ? ev[3] : ev[4];
            p.EV[5 * e + lane] = val;
        }
    }

    // ---- sample_skeleton for one chain: sample.jl:253-284
    __device__ __forceinline__ void run(int64_t c) {
        const int d = p.d;
        tgt.load(p.tparams, d, lane);
        rng.init(p.seed, (uint64_t)(p.chain0 + c));
        if (p.tape_exp) {
            const size_t gc = (size_t)(p.chain0 + c);
            te = p.tape_exp + gc * (size_t)p.tape_n_exp;
            tu = p.tape_uni + gc * (size_t)p.tape_n_uni;
            tn = p.tape_nrm + gc * (size_t)p.tape_n_nrm;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = lane + WAVE * k;
            act[k] = i < d;
            x[k] = act[k] ? p.xinit[(size_t)c * d + i] : 0.0;       // copies: inputs are never mutated (Q7)
            v[k] = act[k] ? p.vinit[(size_t)c * d + i] : 0.0;
        }
        double t = 0.0, horizon = p.tmax, ts = 0.0, tp = 0.0, E = 0.0, lb = 0.0, ar = 0.0;
        const bool adaptive = p.adaptive != 0;
        Bound B;
        B.box = 0.0; B.cum = 0.0; B.h = 0.0;
        {
            const double ev0[5] = {0, 0, 0, 0, 0};
            record(c, 0, t, horizon, ar, 0, ev0, 0, 0);               // sample.jl:272
        }
        for (int64_t k = 1; k < p.n_sk; ++k) {                        // sample.jl:275-280
            int32_t errored = 0, rejected = 0, hits = 0;              // SamplingLoopInplace.jl:28-31
            double ev[5] = {0, 0, 0, 0, 0};
            bool accept = false;
            int64_t it = 0;
            while (!accept) {                                         // :33-35
                if (++it > p.cap) { if (lane == 0) atomicAdd(&p.status[0], 1); return; }
                if (uni(overrun)) { if (lane == 0) atomicAdd(&p.status[3], 1); return; }   // tape exhausted: stop this chain
                // one_step_of_thinning!, :65-85
                B = upper_bound(horizon);
                E = r_exp();
                next_event(B, E, tp, lb);
                if (uni(tp > horizon)) {                              // move_to_horizon!, :87-101
                    flow_state(horizon);
                    ts += horizon;
                    hits++;
                    if (adaptive) horizon = horizon * 1.01;
                } else {                                              // moves_until_horizon!, :103-111
                    while (uni(tp < horizon) && !accept) {
                        if (++it > p.cap) { if (lane == 0) atomicAdd(&p.status[0], 1); return; }
                        if (uni(overrun)) { if (lane == 0) atomicAdd(&p.status[3], 1); return; }
                        const double lam = true_rate(tp);             // ac_step!, :113-129
                        ar = lam / lb;
                        if (uni(ar > 1.0)) {                          // erroneous_acceptance_rate!, :131-151
                            const double h2 = horizon / 2;
                            B = upper_bound(h2);
                            E = r_exp();
                            next_event(B, E, tp, lb);
                            if (adaptive) horizon = h2;
                            errored++;
                            const int slot = errored % 5;             // :148 (1-based errored%5+1)
#pragma unroll
                            for (int q = 0; q < 5; ++q) if (q == slot) ev[q] = ar;
                        } else {                                      // ac_step_with_proxy!, :153-168
                            const double u = r_uni();
                            accept = uni(u < ar);
                            if (accept) {                             // if_accept!, :170-186
                                flow_state(tp);
                                velocity_jump();
                                t = t + tp + ts;
                                ts = 0.0;
                                tp = 0.0;
                            } else {                                  // if_reject!, :188-203
                                E = E + r_exp();
                                next_event(B, E, tp, lb);
                                if (adaptive) horizon = horizon / 1.04;
                                rejected++;
                            }
                            if (!accept && uni(tp > horizon)) {       // move_to_horizon2!, :205-217
                                flow_state(horizon);
                                ts += horizon;
                                hits++;
                            }
                        }
                    }
                }
            }
            record(c, k, t, horizon, ar, errored, ev, rejected, hits);
        }
        if (guard && lane == 0) atomicAdd(&p.status[1], guard);
        if (overrun && lane == 0) atomicAdd(&p.status[3], 1);
    }
};

template <int SAMPLER, int TARGET, int K>
__global__ __launch_bounds__(256) void skeleton_kernel(const SkelParams p) {
    const int64_t c = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (c >= p.C) return;
    Chain<SAMPLER, TARGET, K> chain(p, lane_id());
    chain.run(c);
}

#ifndef __HIPCC_RTC__
// host-side launcher, one per (sampler, target); defined in inst_*.hip
template <int SAMPLER, int TARGET>
hipError_t launch_skeleton(const SkelParams &p, hipStream_t stream);

#define PDMP_DEFINE_LAUNCHER(S, T)                                                                  \
    template <>                                                                                     \
    hipError_t launch_skeleton<S, T>(const SkelParams &p, hipStream_t stream) {                     \
        const dim3 block(256), grid((unsigned)((p.C + 3) / 4));                                     \
        const int K = (p.d + WAVE - 1) / WAVE;                                                      \
        switch (K) {                                                                                \
        case 1: hipLaunchKernelGGL((skeleton_kernel<S, T, 1>), grid, block, 0, stream, p); break;   \
        case 2: hipLaunchKernelGGL((skeleton_kernel<S, T, 2>), grid, block, 0, stream, p); break;   \
        case 3: case 4: hipLaunchKernelGGL((skeleton_kernel<S, T, 4>), grid, block, 0, stream, p); break; \
        default: return hipErrorInvalidValue;                                                       \
        }                                                                                           \
        return hipGetLastError();                                                                   \
    }
#endif  // __HIPCC_RTC__

PDMP_NS_END
