// lane_bps_kernel.hpp -- ONE LANE PER CHAIN for the Bouncy Particle Sampler (BASELINE config 3: BPS, d = 50, banana; round 3).
// The large-ensemble form of sample_skeleton (src/sample.jl:253-284) for BPS (src/Samplers/BouncyParticleSamplers.jl) with the
// constructors' default bound (grid_size = 10, scalar grid bound, forward-mode slopes) on the isotropic Gaussian and the banana,
// 25 <= d <= 52 (the dimensions the engine runs with G = 4 lanes per chain otherwise).
//
// Why (profiles/r2, r3): the group kernel (G = 4, 16 chains per wave) issues 194 VALU wave-instructions per BPS event at config 3 --
// bound 42 %, velocity jump + refresh 36 %, record 12 % of the wave cycles (scripts/stamps.py) -- and nearly all of it is per-CHAIN
// work (the affine-tail bound is O(1) per grid time: ten forward-mode evaluations of the banana's two twisted coordinates, nine
// tangent cells with a division each, the cumulative search; the random stream, log, four more divisions) that the four lanes of a
// chain execute redundantly, plus ~25 % of redundant coordinate work at 13 slots.  With one chain per lane that work is shared by 64
// chains per instruction and no sum crosses lanes.
//
// Arithmetic: BIT-IDENTICAL to the group kernel with G = 4 (the "GROUP(4)" order of the checker, affine tail included): a lane keeps
// four partial sums -- coordinate i = 4 s + g goes to partial g, slots ascending -- combined as ((p0 + p1) + (p2 + p3)); every other
// expression is GroupChain's, operation for operation (group_kernel.hpp: lr_prepare / lr_eval, bound_invert_scalar, true_rate,
// velocity_jump, refresh_velocities).  The choice between the two kernels cannot be seen in the results.
//
// State: x and v as doubles in registers (2 x 52; one wave per SIMD, 512 registers per lane).  record! (Composites.jl:239-260): each
// lane stores its own two rows straight from its registers -- unlike the Zig-Zag lane kernel (lane_kernel.hpp), whose launch is bound by
// the HBM write rate and therefore assembles whole 64-byte lines in LDS, this launch is bound by its arithmetic (at ~25 ms it writes
// 80.7 GB at ~3.2 TB/s), and the L2 merges the 8-byte pieces of a row.  Velocity refreshes (BouncyParticleSamplers.jl:65-71) are
// drawn by the whole wave: the Box-Muller pairs of all refreshing chains of the pass, back to back over the 64 lanes, handed to the
// owning lanes through LDS.
#pragma once
#include "lane_kernel.hpp"

namespace pdmp {

constexpr int lane_bps_lds_per_wave(int lk) { return WAVE * ((4 * lk) | 1) + WAVE + 5 * WAVE; }   // doubles: 64 rows of normals (odd stride), the list of refreshing lanes, error_value_ar's five-slot ring per chain

// The 2 x 52 state doubles leave ~300 registers: the terms of a sum over the coordinates are computed slot by slot -- without this the
// scheduler hoists all 52 independent products in front of their additions (104 more live registers: spills).
#define BPS_SLOT_FENCE(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))

template <int TARGET, int LK, bool DIAG>
struct LaneBps {
    static_assert(TARGET == T_GAUSS_ISO || TARGET == T_BANANA, "BPS lane kernel: isotropic Gaussian and banana");
    static_assert(LK >= LANE_LK_MIN && LK <= LANE_LK_MAX, "BPS lane kernel: 7 .. 13 slots of four coordinates");
    static constexpr int LD = 4 * LK;
    static constexpr int LANE_DMIN = LD - 3;             // the last slot may be partly padding, no other
    static constexpr int LN = LD | 1;                    // LDS row stride of the normals
    static constexpr bool kBanana = TARGET == T_BANANA;
    static constexpr int kAff = kBanana ? 2 : 0;         // GTarget::kAffineFrom: coordinates i >= kAff have d grad_i / dx_j = delta_ij
    using Tgt = GTarget<TARGET>;
    const SkelParams &p;
    const int lane, d;
    Tgt tgt;
    double x[LD], v[LD];
    ChainRng rng;
    int32_t guard = 0;
    double *tile;                 // this wave's LDS: 64 rows of LN doubles (refreshed velocities), then 64 ints (refreshing lanes)
    double lr_A = 0.0, lr_B = 0.0;
    bool lr_ok = false;           // (lr_A, lr_B) belong to the current (x, v)

    __device__ __forceinline__ LaneBps(const SkelParams &sp, int l, double *t_) : p(sp), lane(l), d(sp.d), tile(t_) { tgt.init(nullptr, sp.d); }
    __device__ __forceinline__ bool in_dim(int i) const { return i < LANE_DMIN || i < d; }
    __device__ __forceinline__ static double gsum4(const double (&a)[4]) { return (a[0] + a[1]) + (a[2] + a[3]); }   // the xor-1, xor-2 butterfly

    // ---- random stream, prefetched per pass (as LaneChain): the pass's exponential, acceptance uniform and jump uniform are stream
    // positions n, n+1, n+2 = Philox blocks n >> 1 and (n >> 1) + 1
    struct Prefetch { uint32_t a0, a1, a2, a3, b0, b1, b2, b3; uint64_t blk; };
    __device__ __forceinline__ Prefetch rng_prefetch() const {
        Prefetch f;
        f.blk = rng.n >> 1;
        uint32_t w[4];
        rng.block_at(f.blk, w);
        f.a0 = w[0]; f.a1 = w[1]; f.a2 = w[2]; f.a3 = w[3];
        uint32_t u[4];
        rng.block_at(f.blk + 1, u);
        f.b0 = u[0]; f.b1 = u[1]; f.b2 = u[2]; f.b3 = u[3];
        return f;
    }
    __device__ __forceinline__ double draw_uniform(const Prefetch &f) {
        const uint32_t idx = (uint32_t)(rng.n - 2 * f.blk);
        const bool hi2 = (idx & 2u) != 0u, odd = (idx & 1u) != 0u;
        const uint32_t lo_a = odd ? f.a2 : f.a0, hi_a = odd ? f.a3 : f.a1, lo_b = odd ? f.b2 : f.b0, hi_b = odd ? f.b3 : f.b1;
        ++rng.n;
        return bits_to_u(hi2 ? lo_b : lo_a, hi2 ? hi_b : hi_a);
    }

    // ---- "affine tail" (GroupChain::lr_prepare / lr_eval): g(x + v t).v = head(t) + A + B t
    __device__ __forceinline__ void lr_prepare() {
        if (__ballot(!lr_ok) == 0ull) return;
        double pa[4] = {0.0, 0.0, 0.0, 0.0}, pb[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < LK; ++s) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = 4 * s + g;
                if (i >= kAff) {                                  // (the group kernel adds +0.0 for the head and the padding: the identity on these sums)
                    const double ta = x[i] * v[i], tb = v[i] * v[i];          // grad_i = x_i on the affine coordinates: gd = {x_i, v_i}
                    pa[g] += in_dim(i) ? ta : 0.0;
                    pb[g] += in_dim(i) ? tb : 0.0;
                }
            }
            BPS_SLOT_FENCE(pa); BPS_SLOT_FENCE(pb);
        }
        const double a = gsum4(pa), b = gsum4(pb);
        if (!lr_ok) { lr_A = a; lr_B = b; lr_ok = true; }
    }
    __device__ __forceinline__ void lr_eval(double t, double &fv, double &fe) const {
        fv = fma(lr_B, t, lr_A);
        fe = lr_B;
        if constexpr (kBanana) {                                  // the two twisted coordinates on forward-mode numbers (d >= 25 here)
            const Dual a{fma(v[0], t, x[0]), v[0]}, b{fma(v[1], t, x[1]), v[1]};
            const auto sh = tgt.template shared<Dual>(a, b);      // {g_0, g_1}
            const double hv = (0.0 + sh.g0.v * v[0]) + sh.r.v * v[1], he = (0.0 + sh.g0.e * v[0]) + sh.r.e * v[1];
            fv = hv + fv;
            fe = he + fe;
        }
    }
    // ---- GroupChain::bound_invert_scalar<false> for this chain (upper_bound_grid, UpperBound.jl:92-137, + next_event, :264-273): the ten
    // grid times and nine cells in grid order, lazily (no chain of the wave needs a cell behind the one that covers its exponential)
    __device__ __forceinline__ void bound_scalar(double harg, double E, double &tpo, double &lbo) {
        const double h = harg / (double)(GNG - 1);
        const bool sgn = p.signed_bound != 0;
        const double refresh = sgn ? p.refresh : 0.0;                              // AbstractPDMP.jl:104-112
        lr_prepare();
        double acc = 0.0, cprev = 0.0, c0 = 0.0, c1 = 1.0, last = 0.0;
        int inv_j = 0;
        bool found = false;
        tpo = __builtin_huge_val();
        auto point = [&](double t, double &val, double &slp) {
            double dv, de;
            lr_eval(t, dv, de);
            val = (sgn ? dv : fmax(0.0, dv)) + p.refresh;                          // includes refresh_rate (Q2)
            slp = sgn ? de : ((0.0 < dv) ? de : 0.0);
        };
        // One wave per SIMD runs in order: the ten grid times are evaluated side by side, then the nine cells (nine independent
        // divisions), then the cumulative search -- written as a lazy cell-by-cell loop the pass was one long dependency chain.
        double f[GNG], sl[GNG], bj[GNG - 1];
#pragma unroll
        for (int j = 0; j < GNG; ++j) point((j == GNG - 1) ? harg : (double)j * h, f[j], sl[j]);
#pragma unroll
        for (int j = 0; j < GNG - 1; ++j) {
            const double num = fma(sl[j + 1], h, f[j] - f[j + 1]);                 // :123 relative form
            const double den = sl[j + 1] - sl[j];
            bool fast = false;
            if constexpr (Tgt::kLinear) fast = (__ballot(den != 0.0) == 0ull);     // (same bits either way: GroupChain::cell_q)
            const double q = fast ? ((num > 0.0) ? h : 0.0) : fmin(fmax(num / den, 0.0), h);   // :124-125
            const double inter = fma(sl[j], q, f[j]);                              // :127
            bj[j] = fmax(fmax(fmax(f[j], f[j + 1]), inter), 0.0) + refresh;        // :128-131
        }
#pragma unroll
        for (int j = 0; j < GNG - 1; ++j) {
            acc += bj[j];
            const double cj = acc * h;                                             // :133-134
            if (!found && !(cj < E)) { found = true; inv_j = j; c0 = cprev; c1 = cj; lbo = bj[j]; }   // searchsortedfirst, :265
            cprev = cj;
            last = bj[j];
        }
        if (found) tpo = (double)inv_j * h + (E - c0) / (c1 - c0) * h;             // next_event, :269
        else lbo = last;                                                           // box_max[end], :270
    }
    __device__ __forceinline__ double true_rate(double t, double &raw) {           // BouncyParticleSamplers.jl:39-42 (unsigned, Q3)
        lr_prepare();
        double fe;
        lr_eval(t, raw, fe);
        return fmax(0.0, raw) + p.refresh;
    }
    __device__ __forceinline__ void flow(bool move, double tf) {                   // BouncyParticleSamplers.jl:32: x + v t
        if (move) {
#pragma unroll
            for (int i = 0; i < LD; ++i) x[i] = fma(v[i], tf, x[i]);
            lr_ok = false;
        }
    }
    // grad-U at the stored state: coordinate i (banana: the shared part from x_0, x_1)
    __device__ __forceinline__ double grad_at(int i, const BananaSh<double> &sh) const {
        if constexpr (kBanana) return i == 0 ? sh.g0 : (i == 1 ? sh.r : x[i]);
        else return x[i];
    }
    // velocity_jump, BouncyParticleSamplers.jl:50-74, for the lanes with `mine`: reflection, or true = the chain asks for a refresh
    __device__ __forceinline__ bool velocity_jump(bool mine, double raw, double u) {
        bool want_refresh = false;
        if (mine) {
            lr_ok = false;
            BananaSh<double> sh{};
            if constexpr (kBanana) sh = tgt.template shared<double>(x[0], x[1]);
            const double gv = raw;                                                 // dot(grad-U(x), v): the accepted proposal's g.v, :51-52
            const double bounce = fmax(0.0, gv);
            const double prob = bounce / (bounce + p.refresh);                     // NaN if both 0 (Q9)
            if (u < prob) {
                double pg[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < LK; ++s) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int i = 4 * s + g;
                        const double gi = grad_at(i, sh);
                        const double sq = gi * gi;
                        pg[g] += in_dim(i) ? sq : 0.0;
                    }
                    BPS_SLOT_FENCE(pg);
                }
                const double gg = gsum4(pg);
                if (gg != 0.0) {
                    const double scale = 2.0 * gv / gg;                            // :60
#pragma unroll
                    for (int i = 0; i < LD; ++i) {
                        const double gi = grad_at(i, sh);
                        const double sg = scale * gi;
                        if (in_dim(i)) v[i] = v[i] - sg;
                    }
                }
            } else want_refresh = true;                                            // randn!, :67
        }
        return want_refresh;
    }
    // Velocity refresh of the chains whose lanes pass `mine` (:65-71), by the whole wave (GroupChain::refresh_velocities): the
    // Box-Muller pairs (pair P = normals 2P, 2P+1 of the chain's stream) of ALL refreshing chains, back to back over the 64 lanes
    // -- lane l of pass q0 takes flat pair q0 + l = (refreshing chain r, pair q - r P) -- written to the owning lane's LDS row.
    __device__ __forceinline__ void refresh_velocities(bool mine) {
        const int P = (d + 1) >> 1;
        const unsigned long long todo = __ballot(mine);
        const int R = (int)__builtin_popcountll(todo);
        int *list = (int *)(tile + WAVE * LN);
        int l = lane;
        asm volatile("" : "+v"(l));
        __builtin_amdgcn_wave_barrier();
        if (mine) list[__builtin_popcountll(todo & ((1ull << l) - 1ull))] = l;     // the r-th refreshing lane
        __builtin_amdgcn_wave_barrier();
        for (int q0 = 0; q0 < R * P; q0 += WAVE) {
            const int q = q0 + l;
            const bool act = q < R * P;
            const int r = act ? q / P : 0, pr = q - r * P;
            const int L = list[r];
            ChainRng r2 = rng;                                                     // seed words are the same in every lane; counters of lane L
            r2.c2 = (uint32_t)__shfl((int)rng.c2, L, WAVE);
            r2.c3 = (uint32_t)__shfl((int)rng.c3, L, WAVE);
            r2.n = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(rng.n >> 32), L, WAVE) << 32) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)rng.n, L, WAVE);
            if (act) {
                double z0, z1;
                r2.normal_pair(pr, z0, z1);
                double *row = tile + L * LN;
                row[2 * pr] = z0;
                row[2 * pr + 1] = z1;                                              // (an odd d: position d, never read)
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (mine) {
            lr_ok = false;
            rng.normals_done(d);
            const double *row = tile + l * LN;
            double pz[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < LD; ++i) {
                const double z = in_dim(i) ? row[i] : 0.0;
                v[i] = z;
                pz[i & 3] += z * z;                                                // (padding: + 0.0)
                if ((i & 3) == 3) BPS_SLOT_FENCE(pz);
            }
            if (!p.gaussian_velocity) {                                            // normalize!, :71
                const double inv = 1.0 / __dsqrt_rn(gsum4(pz));
#pragma unroll
                for (int i = 0; i < LD; ++i) v[i] = v[i] * inv;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- record!, Composites.jl:239-260: each lane its own rows, straight from the registers
    // error_value_ar's ring (PDMPState.error_value_ar, 5 slots; DIAG kernels only) lives in LDS, [slot][lane]: it is written on the
    // rare ar > 1 path and read once per recorded event -- ten registers that the 2 x 52 state doubles do not leave
    __device__ __forceinline__ double *ev_ring() const {
        int l = lane;
        asm volatile("" : "+v"(l));
        return tile + WAVE * LN + WAVE + l;
    }
    __device__ __forceinline__ void record(bool rec, int64_t c, int32_t k, double t, double horizon, double ar, int32_t errored,
                                           int32_t rejected, int32_t hits) {
        if (!rec) return;
        const size_t e = (size_t)c * (size_t)p.n_sk + (size_t)k;
        double *Xe = p.X + e * (size_t)d, *Ve = p.V + e * (size_t)d;
#pragma unroll
        for (int i = 0; i < LD; ++i) if (in_dim(i)) { Xe[i] = x[i]; Ve[i] = v[i]; }
        p.t[e] = t;
        if constexpr (DIAG) {
            if (p.H) p.H[e] = horizon;
            if (p.AR) p.AR[e] = ar;
            if (p.EB) p.EB[e] = errored;
            if (p.RJ) p.RJ[e] = rejected;
            if (p.HH) p.HH[e] = hits;
            if (p.EV) {
                double *ring = ev_ring();
#pragma unroll
                for (int q = 0; q < 5; ++q) { p.EV[5 * e + q] = ring[q * WAVE]; ring[q * WAVE] = 0.0; }   // (get_event_state!, :28-31: reset for the next event)
            }
        }
    }

    // ---- GroupChain::run_segment for the 64 chains of this wave.  Hand-off state of batch b (private to this kernel): word w of the
    // chain in lane l at state[(b (2d+4) + w) 64 + l] -- x_0..x_{d-1}, v_0..v_{d-1}, t, horizon, ar, the stream position.
    __device__ __forceinline__ void run_segment(int64_t b, int64_t c, bool chain_ok, int r, int64_t seg_len, double *state, int nseg) {
        const int64_t k_begin = 1 + (int64_t)r * seg_len;
        int64_t k_end = k_begin + seg_len;
        if (k_end > p.n_sk) k_end = p.n_sk;
        int64_t cs = c;                                   // the chain's index in the ORIGINAL ensemble (state slot, initial state, random stream)
        if constexpr (DIAG) { if (p.chain_map && chain_ok) cs = p.chain_map[c]; }
        double *st = state + (size_t)(cs >> 6) * (size_t)(2 * d + 4) * WAVE + (size_t)(cs & 63);
        double t = 0.0, horizon = p.tmax, ar = 0.0;
        uint64_t n0 = 0;
        bool from_state = r > 0;
        if constexpr (DIAG) from_state = from_state || p.resume != 0;
        if (from_state) {
#pragma unroll
            for (int i = 0; i < LD; ++i) {
                x[i] = in_dim(i) ? st[(size_t)i * WAVE] : 0.0;
                v[i] = in_dim(i) ? st[(size_t)(d + i) * WAVE] : 0.0;
            }
            t = st[(size_t)(2 * d) * WAVE]; horizon = st[(size_t)(2 * d + 1) * WAVE]; ar = st[(size_t)(2 * d + 2) * WAVE];
            n0 = (uint64_t)__double_as_longlong(st[(size_t)(2 * d + 3) * WAVE]);
        } else {
#pragma unroll
            for (int i = 0; i < LD; ++i) {
                const bool ok = chain_ok && in_dim(i);
                x[i] = ok ? p.xinit[(size_t)cs * d + i] : 0.0;                      // copies: inputs are never mutated (Q7)
                v[i] = ok ? p.vinit[(size_t)cs * d + i] : 0.0;
            }
        }
        rng.init_lane(p.seed, (uint64_t)(p.chain0 + cs));
        rng.restore(n0);
        lr_ok = false;
        double ts = 0.0, tp = 0.0, E = 0.0, lb = 0.0;
        int32_t errored = 0, rejected = 0, hits = 0;
        if constexpr (DIAG) {
            double *ring = ev_ring();
#pragma unroll
            for (int q = 0; q < 5; ++q) ring[q * WAVE] = 0.0;
        }
        const bool adaptive = p.adaptive != 0;
        bool done = !chain_ok;
        if (r == 0) record(!done, c, 0, t, horizon, ar, 0, 0, 0);                 // sample.jl:272
        int32_t k = (int32_t)k_begin, it = 0;
        const int32_t cap = (int32_t)p.cap, kend = (int32_t)k_end;
        bool fin_before = false, stopped = false;                                 // time-horizon variant: `while state.t < T`, sample.jl:372
        if constexpr (DIAG) { fin_before = !(t < p.t_stop); if (fin_before) done = true; }
        if (k >= kend) done = true;
        enum : int { FRESH = 0, ERR = 1, REUSE = 2 };
        bool propose = false;
        int mode = FRESH;
        double hB = horizon;
        while (__ballot(!done) != 0ull) {
            bool accepted = false, proposed = false, refresh = false, move = false;
            double tf = 0.0, raw = 0.0, ujump = 0.0;
            const Prefetch pf = rng_prefetch();
            if (!done && !propose) {
                // one_step_of_thinning! (:65-85), erroneous_acceptance_rate! (:131-151) or if_reject! (:188-203); the scalar bound is
                // rebuilt from the unchanged (x, v, hB) after a rejection (GroupChain: not resumable)
                const double harg = (mode == ERR) ? horizon / 2 : ((mode == REUSE) ? hB : horizon);
                hB = harg;
                const double e1 = -det_log(draw_uniform(pf));
                E = (mode == REUSE) ? E + e1 : e1;                                // cumulative only after a rejection (:190)
                bound_scalar(harg, E, tp, lb);
                if (mode == ERR) {
                    if (adaptive) horizon = harg;
                    errored++;
                    if constexpr (DIAG) ev_ring()[(errored % 5) * WAVE] = ar;     // :148 (1-based errored%5+1)
                    propose = tp < horizon;                                       // moves_until_horizon! loop test, :106
                } else if (mode == REUSE) {
                    if (adaptive) horizon = horizon / 1.04;                       // :194
                    rejected++;
                    if (tp > horizon) { move = true; tf = horizon; ts += horizon; hits++; }   // move_to_horizon2!, :205-217
                    propose = tp < horizon;
                } else if (tp > horizon) {                                        // move_to_horizon!, :87-101
                    move = true; tf = horizon;
                    ts += horizon;
                    hits++;
                    if (adaptive) horizon = horizon * 1.01;
                } else {
                    propose = tp < horizon;                                       // tp == horizon: retry (Q6)
                }
                mode = FRESH;
                if (++it > cap) { done = true; atomicAdd(&p.status[0], 1); }
            }
            if (!done && propose) {
                const double lam = true_rate(tp, raw);                            // ac_step!, :113-129
                ar = lam / lb;
                propose = false;
                if (ar > 1.0) {
                    mode = ERR;
                } else {                                                          // ac_step_with_proxy!, :153-168
                    const double u = draw_uniform(pf);
                    if (u < ar) {                                                 // if_accept!, :170-186
                        move = true; tf = tp;
                        ujump = draw_uniform(pf);
                        accepted = true;
                    } else {
                        mode = REUSE;                                             // if_reject! continues in the BOUND phase
                    }
                }
                proposed = true;
            }
            flow(move, tf);                                                       // to the horizon, or to the accepted proposal
            refresh = velocity_jump(accepted, raw, ujump);
            if (__ballot(refresh) != 0ull) refresh_velocities(refresh);
            if (accepted) {
                t = t + tp + ts;
                ts = 0.0;
                tp = 0.0;
            }
            record(accepted, c, k, t, horizon, ar, errored, rejected, hits);
            if (accepted) {
                errored = 0; rejected = 0; hits = 0;                              // get_event_state!, :28-31 (the error ring: reset in record)
                it = 0;
                if (++k >= kend) done = true;
                if constexpr (DIAG) { if (!(t < p.t_stop)) { done = true; stopped = true; } }
            }
            if (proposed && ++it > cap && !done) { done = true; atomicAdd(&p.status[0], 1); }
        }
        if (chain_ok) {                                                           // hand-off / final state (write-through stores, as the group kernel's)
            // (the slot's address is rebuilt from an opaque copy of the chain index: computed before the event loop, the 2 x 52 store
            // addresses would be kept in registers for the whole segment)
            int64_t cs2 = cs;
            asm volatile("" : "+v"(cs2));
            double *so = state + (size_t)(cs2 >> 6) * (size_t)(2 * d + 4) * WAVE + (size_t)(cs2 & 63);
#pragma unroll
            for (int i = 0; i < LD; ++i) if (in_dim(i)) { publish(so + (size_t)i * WAVE, x[i]); publish(so + (size_t)(d + i) * WAVE, v[i]); }
            publish(so + (size_t)(2 * d) * WAVE, t); publish(so + (size_t)(2 * d + 1) * WAVE, horizon); publish(so + (size_t)(2 * d + 2) * WAVE, ar);
            publish(so + (size_t)(2 * d + 3) * WAVE, __longlong_as_double((long long)rng.n));
        }
        if (guard) { atomicAdd(&p.status[1], guard); guard = 0; }
        if constexpr (DIAG) {
            if (p.count && chain_ok && !fin_before && (stopped || r == nseg - 1)) p.count[c] = (int64_t)k;
            if (p.count && chain_ok && fin_before && r == 0) p.count[c] = 1;
        }
    }
};

template <int TARGET, int LK, bool DIAG>
__global__ __launch_bounds__(256, 1) void lane_bps_kernel(const GroupParams gp) {
    extern __shared__ double smem[];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    double *tile = smem + (size_t)wave * lane_bps_lds_per_wave(LK);
    LaneBps<TARGET, LK, DIAG> chain(gp.s, lane, tile);
    const int nb = (int)gp.nb, n_units = nb * gp.nseg;
    bool lost = false;
    for (int pulled = 0; pulled <= n_units && !lost; ++pulled) {
        int u = 0;
        if (lane == 0) u = atomicAdd(gp.ctr, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= n_units) break;
        const int r = u / nb, b = u - r * nb;
        if (r > 0) lost = !wait_progress(gp.progress + b, r);
        if (!lost) {
            const int64_t c = (int64_t)b * WAVE + lane;
            chain.run_segment(b, c, c < gp.s.C, r, gp.seg_len, gp.state, gp.nseg);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&gp.progress[b], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lost && lane == 0) atomicAdd(&gp.s.status[2], 1);
}

#ifndef __HIPCC_RTC__
template <int TARGET, int LK, bool DIAG>
inline hipError_t launch_lane_bps(const GroupParams &gp_in, int rounds, hipStream_t stream) {
    GroupParams gp = gp_in;
    constexpr int wpb = 4;
    const size_t lds = (size_t)wpb * lane_bps_lds_per_wave(LK) * sizeof(double);
    void (*kernel)(const GroupParams) = lane_bps_kernel<TARGET, LK, DIAG>;
    int per_cu = 0, dev = 0, cus = 0;
    hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE * wpb, lds);
    if (e != hipSuccess) return e;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (per_cu < 1) per_cu = 1;
    if (cus < 1) cus = 256;
    gp.nb = (gp.s.C + WAVE - 1) / WAVE;
    const int64_t slots = (int64_t)cus * per_cu * wpb;
    lane_schedule(gp.nb, slots, gp.s.n_sk, gp.nseg, rounds, gp.nseg, gp.seg_len);
    int64_t blocks = (gp.nb * gp.nseg + wpb - 1) / wpb;
    if (blocks > (int64_t)cus * per_cu) blocks = (int64_t)cus * per_cu;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(WAVE * wpb), lds, stream, gp);
    return hipGetLastError();
}
template <int TARGET, int LK0, int LK1>
inline hipError_t launch_lane_bps_by_slots(int lk, const GroupParams &gp, int rounds, hipStream_t stream) {
    if constexpr (LK0 > LK1) return hipErrorNotSupported;
    else {
        const bool diag = gp.s.H || gp.s.AR || gp.s.EB || gp.s.RJ || gp.s.HH || gp.s.EV ||
                          gp.s.count || gp.s.chain_map || gp.s.resume || gp.s.t_stop < __builtin_huge_val();   // (the time-horizon features: DIAG kernels)
        return lk != LK0 ? launch_lane_bps_by_slots<TARGET, LK0 + 1, LK1>(lk, gp, rounds, stream)
                         : (diag ? launch_lane_bps<TARGET, LK0, true>(gp, rounds, stream) : launch_lane_bps<TARGET, LK0, false>(gp, rounds, stream));
    }
}
template <int LK0, int LK1>
inline hipError_t launch_lane_bps_range(int target, const GroupParams &gp, int rounds, hipStream_t stream) {
    const int lk = (gp.s.d + 3) / 4;
    switch (target) {
    case T_GAUSS_ISO: return launch_lane_bps_by_slots<T_GAUSS_ISO, LK0, LK1>(lk, gp, rounds, stream);
    case T_BANANA: return launch_lane_bps_by_slots<T_BANANA, LK0, LK1>(lk, gp, rounds, stream);
    default: return hipErrorNotSupported;
    }
}
hipError_t launch_lane_bps_13(int target, const GroupParams &gp, int rounds, hipStream_t stream);      // LK = 13 (d = 49 .. 52)
hipError_t launch_lane_bps_a(int target, const GroupParams &gp, int rounds, hipStream_t stream);       // LK = 7 .. 9
hipError_t launch_lane_bps_b(int target, const GroupParams &gp, int rounds, hipStream_t stream);       // LK = 10 .. 12
#endif

}  // namespace pdmp
