// lane_kernel.hpp -- ONE LANE PER CHAIN: the large-ensemble form of the Zig-Zag hot path (sample_skeleton, src/sample.jl:253-284) for the
// constructors' default bound (grid_size = 10, vectorized signed bound, forward-mode slopes) on the isotropic Gaussian and the banana,
// unit speeds (|v_i| = 1, the reference's initial velocities), 27 <= d <= 52 (the dimensions the engine runs with G = 4 lanes per chain
// otherwise; one instantiation per number of four-coordinate slots).
//
// Why (profiles/r2): the group kernel (G = 4 lanes per chain, 16 chains per wave) is bound by fp64 VALU issue -- 81 wave-instructions
// per event at config 2 -- and two thirds of those are per-CHAIN scalar work (random stream, log, cumulative sums, inversion, the
// Categorical's bookkeeping, the state machine) that the four lanes of a chain execute redundantly.  With one chain per lane that work
// is shared by 64 chains per instruction and no sum over coordinates crosses lanes any more (no DPP butterflies, no ballots in the
// Categorical walk).
//
// State ("signed positions"): with v_i = +-1 the Zig-Zag dynamics only ever needs  y_i = x_i v_i:
//   rate_i(t) = max(0, (x_i + v_i t) v_i) = max(0, y_i + t),   flow: y_i <- y_i + t,   flip of coordinate m: y_m <- -y_m
// -- in IEEE arithmetic, not only on paper: multiplying by +-1 is exact and rounding to nearest is symmetric, so fl(x - t) (-1) =
// fl(-x + t).  A lane holds y_i in VGPRs (52 doubles) and the signs of v packed in two words; x_i = y_i v_i and v_i are rebuilt (sign-bit
// xor) where they are needed: in record!, and for the banana's two twisted coordinates.  The only visible difference to carrying x and
// v: when x_i + v_i t cancels to exactly 0 with v_i = -1 the zero's sign differs (an event of probability ~2^-52 per operation).
//
// Arithmetic: BIT-IDENTICAL to the group kernel with G = 4 (the "GROUP(4)" order of the checker).  A lane keeps four partial sums --
// coordinate i = 4 s + g goes to partial g, slots s in ascending order -- and combines them as ((p0 + p1) + (p2 + p3)), which is what the
// xor-1 / xor-2 butterfly over the four lanes computes; the Kogge-Stone scan inside the Categorical's hit slot is written out for four
// elements; every other expression is GroupChain's (group_kernel.hpp), operation for operation.  So the choice between the two
// kernels -- made per launch from the number of chains -- cannot be seen in the results (tests/test_gpu_lane_kernel.py compares them,
// and both with the checker), and sharding invariance holds across it.
//
// record! (Composites.jl:239-260) goes through the chain's LDS row so that a store instruction writes 16 consecutive doubles of each of
// four chains (a lane's own row lies n_sk d 8 bytes from its neighbour's), and only WHOLE lines -- of 128 bytes, the L2's, since round 5
// (lane_line()): see `record` below.
// One wave per SIMD (512 registers per lane); the launch is bound by the rate at which the events can be written.
#pragma once
#include "group_kernel.hpp"
#include <type_traits>
#ifndef __HIPCC_RTC__
#include <cstdio>
#include <cstdlib>
#endif

#ifndef LANE_WAVES
#define LANE_WAVES 1          // waves per SIMD the kernel is compiled for (512 registers per lane at 1)
#endif
#ifndef LANE_BATCH
#define LANE_BATCH 8          // stores whose LDS reads are in flight together (record)
#endif
#ifndef LANE_STORE_AUX
#define LANE_STORE_AUX 2      // nt: streaming stores (measured 16.6 -> 15.1 ms at config 2); cache-policy bits of the event stores (buffer_store ... sc0 = 1 / nt = 2 / sc1 = 16)
#endif

PDMP_NS_BEGIN

// LK = slots of the emulated G = 4 layout: one instantiation per LK = 7 .. 13 serves d = 4 LK - 3 .. 4 LK (25 .. 52; the engine's default
// layout is G = 4 from d = 27 on)
constexpr int LANE_LK_MIN = 1, LANE_LK_MAX = 13;          // (round 4: 3 .. 6 slots, d = 9 .. 24, added: inst_lane_zigzag_c.hip, inst_lane2q_zigzag_c.hip;
                                                         //  then 1 and 2 slots, d = 1 .. 8 -- rows shorter than a 64-byte line, LaneChain::kShort: inst_lane_zigzag_d.hip, inst_lane2q_zigzag_d.hip)
// gl = lanes per chain (1; 2: the chain's coordinates split over a lane pair, lane_kernel.hpp: GL)
// pg = coordinates of a slot per lane (4; 2: the "quad pair" form, GL = 2 at d <= 52), dmax = pg gl lk = a chain's coordinates with padding
// Doubles per written line (LaneChain::LN): a store instruction writes whole lines of this length only.  64 bytes until round 5; 128 bytes -- the
// L2's line -- for one lane per chain at d >= 25 and for the pair form at d = 53 .. 104 (LANE_LINE16=0: A/B builds): a 128-byte line that two store instructions fill half by
// half is written to memory 1.19 times on average (scripts/probes/oob_probe.hip; the launch of config 2: 14.04 64-byte lines per event for
// 12.625 of events, 0.375 too many per row of X and of V, PDMP_LANE_DEBUG).
#ifndef LANE_LINE16
#define LANE_LINE16 1
#endif
constexpr int lane_line(int lk, int gl = 1, int pg = 4) { return (LANE_LINE16 && pg == 4 && lk >= 7) ? 16 : 8; }   // (the pair form at d = 53 .. 104 too; the quad pairs -- pg = 2, two waves per SIMD -- keep 8)
constexpr int lane_row_stride(int lk, int gl = 1, int pg = 4) { return (lane_line(lk, gl, pg) - 1 + pg * gl * lk) | 1; }   // LDS row: up to LN - 1 carried doubles + dmax coordinates, odd stride
constexpr bool lane_wide(int lk, int gl, int pg) { return 7 + pg * gl * lk > 64; }                 // a row's sign bits need more than one 64-bit word
constexpr int lane_meta_doubles(int lk, int gl = 1, int pg = 4) { return lane_wide(lk, gl, pg) ? 4 : 2; }   // store parameters of a chain: 16 bytes (64 sign bits) / 32 bytes (128)
constexpr int lane_lds_per_wave(int lk, int gl = 1, int pg = 4) { return (WAVE / gl) * (lane_row_stride(lk, gl, pg) + lane_meta_doubles(lk, gl, pg) + 8) + (lane_line(lk, gl, pg) == 16 && gl == 1 ? 8 : 0); } // doubles: a row + the store parameters + 8 event times per chain (+ 64 bytes: the list of the chains whose row completes a fourth line, store_tile_listed)

// Two waves per SIMD (256 registers): slot loops run slot by slot -- without this the scheduler hoists the 52 independent terms of a sum
// in front of their additions (104 more live registers, spills).  One wave per SIMD (512 registers, the build that ships): no fence, the
// terms of all slots are in flight together (banana 17.2 -> 16.7 ms).
#if LANE_WAVES == 1 && !defined(LANE_FENCE)
#define LANE_NO_FENCE
#endif
#if defined(PDMP_STAMP)        // diagnostic build (make stamp): wave cycles per phase of the state machine, summed into p.stamps
#define LANE_MARK(n) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[LANE_PREV_##n] += now_ - st_last; st_last = now_; } while (0)
// the phase that ENDS at mark n (static: the marks are passed in program order; 8 = loop top)
#define LANE_PREV_0 8
#define LANE_PREV_1 0
#define LANE_PREV_2 1
#define LANE_PREV_3 2
#define LANE_PREV_4 3
#define LANE_PREV_5 4
#define LANE_PREV_6 5
#define LANE_PREV_7 15
#define LANE_PREV_8 7
#define LANE_PREV_9 6
#define LANE_PREV_10 9
#define LANE_PREV_11 10
#define LANE_PREV_12 11
#define LANE_PREV_13 12
#define LANE_PREV_14 13
#define LANE_PREV_15 14
#elif defined(LANE_MARKS)      // static instruction counts per phase: markers in the assembly
#define LANE_MARK(n) asm volatile("; LANEMARK " #n ::: "memory")
#else
#define LANE_MARK(n)
#endif
#ifdef LANE_NO_FENCE
#define LANE_SLOT_FENCE(a, tt)
#else
#define LANE_SLOT_FENCE(a, tt) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(tt))
#endif

// DIAG: the launch records PDMPHistory's diagnostic arrays (horizon, ar, errored_bound, error_value_ar, rejected, hitting_horizon);
// without them (the throughput case) their pointers, counters and the five-slot ring of error values are not compiled in at all.
// TAPE: pre-recorded variates instead of the Philox stream (pdmp_set_tape; GroupChain's tape source, same format) -- separate
// instantiations (inst_tape_lane.hip), one segment per chain.
//
// GL = 2 (round 3): d = 53 .. 104, the dimensions the engine runs with G = 8 lanes per chain otherwise.  A chain is a PAIR of lanes (32
// chains per wave); lane h of the pair holds coordinates 8 s + 4 h + g (g = 0 .. 3) of slot s = 0 .. LK - 1 in the same 4 LK registers,
// with the same four partial sums: its ((p0 + p1) + (p2 + p3)) is what the xor-1 / xor-2 stages of the eight-lane butterfly leave in a
// quad, and ONE exchange with the partner lane (DPP quad_perm [1,0,3,2]) is the xor-4 stage -- bit-identical to the G = 8 group kernel
// ("GROUP(8)").  Everything per chain (random stream, exponentials, bound bookkeeping, the state machine) runs identically in both
// lanes, as it does in the eight lanes of the group kernel -- but for 32 chains per instruction instead of 8.
struct LaneBits128 { uint64_t lo, hi; };
__device__ __forceinline__ uint64_t lb_shl(uint64_t a, int n) { return a << n; }                 // n in 0 .. 63
__device__ __forceinline__ uint64_t lb_shr(uint64_t a, int n) { return a >> n; }
__device__ __forceinline__ uint64_t lb_or(uint64_t a, uint64_t b) { return a | b; }
__device__ __forceinline__ uint32_t lb_low(uint64_t a) { return (uint32_t)a; }
__device__ __forceinline__ LaneBits128 lb_shl(LaneBits128 a, int n) {                            // n in 0 .. 63
    return {a.lo << n, (a.hi << n) | ((a.lo >> 1) >> (63 - n))};
}
__device__ __forceinline__ LaneBits128 lb_shr(LaneBits128 a, int n) {                            // n in 0 .. 127
    if (n >= 64) { a.lo = a.hi; a.hi = 0; n -= 64; }
    return {(a.lo >> n) | ((a.hi << 1) << (63 - n)), a.hi >> n};
}
__device__ __forceinline__ LaneBits128 lb_or(LaneBits128 a, LaneBits128 b) { return {a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ uint32_t lb_low(LaneBits128 a) { return (uint32_t)a.lo; }
__device__ __forceinline__ uint32_t dpp_xor1_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, DPP_QUAD_XOR1, 0xF, 0xF, true); }

//
// GL = 2 with PG = 2 (round 3, the 4-GPU shard of the headline ensemble): d = 25 .. 52 again, for launches of ~20 000 - 40 000 chains that
// leave most SIMDs without a wave when a wave holds 64 chains.  Lane h of the pair holds coordinates 4 s + 2 h + g (g = 0, 1): its
// p0 + p1 is the xor-1 stage of the FOUR-lane butterfly, the exchange with the partner the xor-2 stage: GROUP(4) again, bit-identical
// to the one-lane form and to the G = 4 group kernel.  26 coordinate registers per lane: two waves per SIMD, no AGPRs.
template <int TARGET, int LK, bool DIAG, bool TAPE = false, int GL = 1, int PG = 4>
struct LaneChain {
    static_assert(TARGET == T_GAUSS_ISO || TARGET == T_BANANA, "lane kernel: isotropic Gaussian and banana");
    static_assert(LK >= LANE_LK_MIN && LK <= LANE_LK_MAX, "lane kernel: 1 .. 13 slots of four coordinates");
    static_assert(GL == 1 || GL == 2, "lane kernel: one lane per chain, or a pair");
    static_assert(PG == 4 || (PG == 2 && GL == 2), "lane kernel: four coordinates of a slot per lane, or two in a pair");
    static_assert(!(TAPE && GL != 1), "tape replay: one lane per chain");
    static constexpr int GE = PG * GL;                   // lanes per chain of the group layout whose summation order this is
    static constexpr int LGE = GE == 4 ? 2 : 3;          // log2(GE)
    static constexpr int LD = PG * LK;                   // coordinate registers per lane
    static constexpr int DMAX = GE * LK;                 // coordinates of a chain, padding included
    static constexpr int CW = WAVE / GL;                 // chains per wave
    static constexpr int LR = lane_row_stride(LK, GL, PG);   // LDS row stride
    static constexpr int LN = lane_line(LK, GL, PG);     // doubles per written line: 8 (64 bytes) or 16 (128 bytes, the L2's line)
    static constexpr bool kWide = lane_wide(LK, GL, PG); // 128 sign bits per row (else 64)
    static constexpr int MD = lane_meta_doubles(LK, GL, PG);   // doubles of store parameters per chain
    static constexpr int LANE_DMIN = DMAX - (GE - 1);    // the last slot may be partly padding, no other
    static constexpr bool kBanana = TARGET == T_BANANA;
    static constexpr bool kShort = DMAX <= 8;            // d <= 8: a row may be shorter than what precedes it in its 64-byte line (open_window: skip)
    using Tgt = GTarget<TARGET>;
    using SBits = typename std::conditional<!kWide, uint64_t, LaneBits128>::type;   // one sign bit per row position (7 carried + DMAX)
    const SkelParams &p;
    const int lane, d;
    int hq = 0;                   // GL = 2: PG h -- my half of every slot (h = lane & 1)
    int dl = PG;                  // GL = 2: how many coordinates of MY last PG are real: d - GE (LK - 1) - PG h (<= 0: none, >= PG: all)
    Tgt tgt;
    double y[LD];                 // y_i = x_i v_i; register j = PG s + g holds coordinate gi(j) = GE s + PG h + g
    uint32_t sg0 = 0, sg1 = 0, sg2 = 0, sg3 = 0;    // bit GE s + g of (sg3:sg2:sg1:sg0) set <=> v = -1 for register PG s + g (GL = 1: bit j, two words)
    ChainRng rng;
    int32_t guard = 0;
    double *tile;                 // this wave's LDS: CW rows of LR doubles, then CW x (sign bits, window offset, valid range), then the event-time lines

    __device__ __forceinline__ LaneChain(const SkelParams &sp, int l, double *t_) : p(sp), lane(l), d(sp.d), tile(t_) {
        tgt.init(nullptr, sp.d);
        if constexpr (GL == 2) { hq = PG * (l & 1); dl = sp.d - GE * (LK - 1) - hq; }
    }
    static constexpr int js(int j) { return j / PG; }    // slot of register j
    static constexpr int jg(int j) { return j % PG; }    // its place among my coordinates of the slot
    static constexpr int lbit(int j) { return GE * (j / PG) + (j % PG); }   // its sign bit
    __device__ __forceinline__ bool head() const { if constexpr (GL == 1) return true; else return hq == 0; }   // the lane that does the chain's scalar stores
    __device__ __forceinline__ int gi(int j) const { if constexpr (GL == 1) return j; else return lbit(j) + hq; }   // coordinate held by register j

    __device__ __forceinline__ static double sgn(double a, uint32_t s) {          // a * v for v = +-1 (exact): flip the sign bit
        return __hiloint2double(__double2hiint(a) ^ (int)s, __double2loint(a));
    }
    __device__ __forceinline__ uint32_t sw(int i) const {                         // sign word of register i's v: 0 or 0x80000000 (i is a constant after unrolling)
        const int lb = lbit(i);
        const uint32_t w = (lb < 32) ? sg0 : ((lb < 64) ? sg1 : ((lb < 96) ? sg2 : sg3));
        const int b = lb & 31;
        return (b == 31) ? (w & 0x80000000u) : ((w << (31 - b)) & 0x80000000u);
    }
    __device__ __forceinline__ double X(int i) const { return sgn(y[i], sw(i)); }
    __device__ __forceinline__ double V(int i) const { return __hiloint2double((int)(0x3ff00000u | sw(i)), 0); }
    __device__ __forceinline__ bool in_dim(int i) const {                         // register i holds a real coordinate
        if constexpr (GL == 1) return i < LANE_DMIN || i < d;
        else return i < LD - PG || jg(i) < dl;
    }
    __device__ __forceinline__ static double gsum4(const double (&a)[PG]) {       // my coordinates of a slot: the first stage(s) of the butterfly
        if constexpr (PG == 4) return (a[0] + a[1]) + (a[2] + a[3]); else return a[0] + a[1];
    }
    __device__ __forceinline__ static double csum(const double (&a)[PG]) {        // the chain's sum: GL = 2 adds the last stage (the partner lane's)
        double r = gsum4(a);
        if constexpr (GL == 2) r = r + dpp_d<DPP_QUAD_XOR1>(r);
        return r;
    }
    // banana, slot 0 of a lane pair: coordinates 0, 1 (twisted) are registers 0, 1 of the head lane; the partner's are plain
    template <class T> __device__ __forceinline__ T hgrad(int i, T x, const BananaSh<T> &sh) const {
        if constexpr (GL == 1) return tgt.grad(i, x, sh);
        else {
            if (i >= 2) return x;
            const T tw = (i == 0) ? sh.g0 : sh.r;
            return head() ? tw : x;
        }
    }

    // ---- random stream, prefetched per pass of the state machine (as GroupChain's, without the lane pairing): a pass draws at most three
    // variates of a chain -- the bound's exponential, the acceptance uniform, the jump's uniform: stream positions n, n+1, n+2 = Philox
    // blocks n >> 1 and (n >> 1) + 1.  Drawn one by one each draw would run the ten Philox rounds whenever ANY of the 64 chains starts a
    // new block, i.e. three times per pass; here both blocks are computed once at the top of the pass and a draw picks its two words.
    // (The eight words are locals of run_segment, not members: a select between members turns into a dynamic address into this object
    // and keeps all of it -- y included -- in scratch memory.)
    struct Prefetch { uint32_t a0, a1, a2, a3, b0, b1, b2, b3; uint64_t blk; };
    const double *tp_e = nullptr, *tp_u = nullptr;       // tape source (TAPE instantiations): this chain's Exp(1) and U(0,1) tapes
    int32_t tp_ie = 0, tp_iu = 0;
    bool tp_over = false;
    __device__ __forceinline__ double tape_uniform() { if (tp_iu < p.tape_n_uni) return tp_u[tp_iu++]; tp_over = true; return 0.5; }
    __device__ __forceinline__ double tape_exponential() { if (tp_ie < p.tape_n_exp) return tp_e[tp_ie++]; tp_over = true; return 1.0; }
    __device__ __forceinline__ Prefetch rng_prefetch() const {
        Prefetch f;
        if constexpr (TAPE) { f.a0 = f.a1 = f.a2 = f.a3 = f.b0 = f.b1 = f.b2 = f.b3 = 0u; f.blk = 0; return f; }
        f.blk = rng.n >> 1;
        uint32_t w[4];
        rng.block_at(f.blk, w);
        f.a0 = w[0]; f.a1 = w[1]; f.a2 = w[2]; f.a3 = w[3];
        uint32_t v[4];
        rng.block_at(f.blk + 1, v);
        f.b0 = v[0]; f.b1 = v[1]; f.b2 = v[2]; f.b3 = v[3];
        return f;
    }
    __device__ __forceinline__ double draw_uniform(const Prefetch &f) {
        if constexpr (TAPE) return tape_uniform();
        const uint32_t idx = (uint32_t)(rng.n - 2 * f.blk);                       // 0..3: the first three draws of a pass (a fourth would repeat idx 3: none is made)
        const bool hi2 = (idx & 2u) != 0u, odd = (idx & 1u) != 0u;
        const uint32_t lo_a = odd ? f.a2 : f.a0, hi_a = odd ? f.a3 : f.a1, lo_b = odd ? f.b2 : f.b0, hi_b = odd ? f.b3 : f.b1;
        ++rng.n;
        return bits_to_u(hi2 ? lo_b : lo_a, hi2 ? hi_b : hi_a);
    }

    // ---- the lazily built vectorized bound, GroupChain::bound_advance_vect<true, false> for this chain
    int b_next = 0;
    double b_acc = 0.0, b_cprev = 0.0, b_last = 0.0, b_cbefore = 0.0;
    bool b_recheck = false;
    BananaSh<Dual> b_shf{};
    __device__ __forceinline__ void bound_begin(double &tpo) {
        b_next = 0; b_acc = 0.0; b_cprev = 0.0; b_last = 0.0; b_cbefore = 0.0; b_recheck = false;
        tpo = __builtin_huge_val();
    }
    __device__ __forceinline__ void bound_resume(double &tpo) { b_recheck = true; tpo = __builtin_huge_val(); }

    __device__ __forceinline__ BananaSh<Dual> head_dual(double t) const {         // GroupChain::shared_at<Dual>: x_0, x_1 flowed, on duals in t
        const double v0 = V(0), v1 = V(1);
        const Dual a{fma(v0, t, X(0)), v0}, b{fma(v1, t, X(1)), v1};
        return tgt.template shared<Dual>(a, b);
    }
    // one general tangent-intersection cell of coordinate i (the banana's slot 0), GroupChain::eval_vect<true> + cell_vect
    __device__ __forceinline__ double cell_general(int i, double t0, double t1, double h, const BananaSh<Dual> &sh0, const BananaSh<Dual> &sh1) const {
        const double v = V(i), x = X(i);
        const Dual xt0{fma(v, t0, x), v}, xt1{fma(v, t1, x), v};
        const Dual gd0 = hgrad(i, xt0, sh0), gd1 = hgrad(i, xt1, sh1);
        const double v0 = gd0.v * v, g0 = gd0.e * v, v1 = gd1.v * v, g1 = gd1.e * v;
        const double num = fma(-g0, t0, fma(g1, t1, v0 - v1));
        const double den = g1 - g0;
        // cell_q: the clamp of num / 0 without the division; coordinates 2, 3 are affine with slope v v = 1 at both ends -- den is 0.0
        // whatever the state, so their division is not even compiled
        const double q = (i < 2 && den != 0.0) ? fmin(fmax(num / den, 0.0), h) : ((num > 0.0) ? h : 0.0);
        const double inter = fma(g0, q, v0);
        return fmax(fmax(fmax(v0, v1), inter), 0.0);
    }
    __device__ __forceinline__ bool bound_advance(double harg, double E, double &tpo, double &lbo, int budget) {
        const double h = harg / (double)(GNG - 1);
        bool found = false, found_now = false;
        if (b_recheck) {
            b_recheck = false;
            if (b_next > 0 && !(b_cprev < E)) {
                found = true;
                tpo = (double)(b_next - 1) * h + (E - b_cbefore) / (b_cprev - b_cbefore) * h;
                lbo = b_last;
            }
        }
        for (int pass = 0; pass < budget; ++pass) {
            const bool act = !found && b_next < GNG - 1;
            if (__ballot(act) == 0ull) break;
            if (act) {
                const int j = b_next;
                const double t0 = (double)j * h;
                double t1 = (j == GNG - 2) ? harg : (double)(j + 1) * h;
                double part[PG] = {};
                if constexpr (kBanana) {                                          // slot 0: x_0, x_1 twisted (and the rest of the slot with them, as the group kernel's slot)
                    if (j == 0) b_shf = head_dual(t0);
                    const BananaSh<Dual> sh0 = b_shf, sh1 = head_dual(t1);
#pragma unroll
                    for (int g = 0; g < PG; ++g) if (in_dim(g)) part[g] += cell_general(g, t0, t1, h, sh0, sh1);   // (d = 2, 3: the slot's padding)
                    b_shf = sh1;
                }
#pragma unroll
                for (int s = kBanana ? 1 : 0; s < LK; ++s) {
#pragma unroll
                    for (int g = 0; g < PG; ++g) {
                        const int i = PG * s + g;
                        if (in_dim(i)) part[g] += fmax(y[i] + t1, 0.0);           // affine cell, monotone: max(v0, v1, 0) = max(v1, 0), v1 = (x + v t1) v
                    }
                    LANE_SLOT_FENCE(part, t1);
                }
                const double bj = csum(part);
                b_acc += bj;
                const double cj = b_acc * h;
                if (!(cj < E)) { found = true; found_now = true; lbo = bj; }
                b_cbefore = b_cprev;
                b_cprev = cj;
                b_last = bj;
                b_next = j + 1;
            }
        }
        if (found_now) tpo = (double)(b_next - 1) * h + (E - b_cbefore) / (b_cprev - b_cbefore) * h;
        const bool fin = found || b_next >= GNG - 1;
        if (fin && !found) lbo = b_last;
        return fin;
    }

    // grad-U .* v of the twisted coordinate i = 0, 1 (plain: gv) -- the head lane's registers 0, 1
    __device__ __forceinline__ double twisted(int i, double gv, const BananaSh<double> &sh) const {
        const double gb = sgn(i == 0 ? sh.g0 : sh.r, sw(i));
        if constexpr (GL == 1) return gb; else return head() ? gb : gv;
    }
    // ---- GroupChain::true_rate for Zig-Zag: sum max(0, grad-U(x + v t) .* v) in the GROUP(4) order
    __device__ __forceinline__ double true_rate(double t) const {
        double part[PG] = {};
        BananaSh<double> sh{};
        if constexpr (kBanana) sh = tgt.template shared<double>(fma(V(0), t, X(0)), fma(V(1), t, X(1)));
#pragma unroll
        for (int s = 0; s < LK; ++s) {
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                const int i = PG * s + g;
                if (in_dim(i)) {
                    double gv = y[i] + t;                                         // (x + v t) v
                    if constexpr (kBanana) { if (i < 2) gv = twisted(i, gv, sh); }
                    part[g] += fmax(0.0, gv);
                }
            }
            LANE_SLOT_FENCE(part, t);
        }
        return csum(part);
    }
    // The state is written in ONE place per pass of the state machine, outside every divergent region (no phi webs over the 104
    // registers of y): flow by tf -- the horizon, the accepted proposal's time, or -0.0 for a lane that does not move (y + (-0.0) = y
    // for every double, zeros included) -- and flip of coordinate m (-1: none).
    __device__ __forceinline__ void flow_and_flip(double tf, int mglob) {
        // m: the sign bit / register position of the flipped coordinate in THIS lane, GE s + g (-1: none, or the partner lane's)
        int m = mglob;
        if constexpr (GL == 2) m = (mglob >= 0 && (mglob & PG) == hq) ? (mglob & ~PG) : -1;
        const uint32_t bit = 1u << (m & 31);                                      // v[m] *= -1
        sg0 ^= (m >= 0 && m < 32) ? bit : 0u;
        if constexpr (GL == 1) sg1 ^= (m >= 32) ? bit : 0u;
        else { sg1 ^= (m >= 32 && m < 64) ? bit : 0u; sg2 ^= (m >= 64 && m < 96) ? bit : 0u; sg3 ^= (m >= 96) ? bit : 0u; }
        // the flip: one compare per slot of four, then per coordinate a select of the lane's word for g = m % 4 and the xor
        const int ms = m >> LGE, mg = m & (PG - 1);                               // (m = -1: slot -1, no slot matches)
        uint32_t xs[PG], mk[LK];
#pragma unroll
        for (int g = 0; g < PG; ++g) xs[g] = (mg == g) ? 0x80000000u : 0u;
#pragma unroll
        for (int s = 0; s < LK; ++s) { mk[s] = (ms == s) ? 0xffffffffu : 0u; if constexpr (GL == 1) asm("" : "+v"(mk[s])); }   // (a mask per slot, opaque to the compiler: the coordinate's flip is then ONE three-input bit operation, hi ^ (mk & xs); left to itself it makes a select and an xor of it)
#pragma unroll
        for (int i = 0; i < LD; ++i) {                                            // padded coordinates are never read
            // y_i + tf (a plain IEEE addition; as an asm so that no pass of the compiler turns  y + (moved ? t : -0.0)  into a select over
            // the 104 registers).  Source and destination are SEPARATE operands: with the destination tied to the source ("+v") every
            // coordinate whose register changes over the loop body -- 51 of 52 -- cost a v_mov_b64 in front of the addition
            double yn;
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(yn) : "v"(y[i]), "v"(tf));
            if constexpr (GL == 1) y[i] = sgn(yn, mk[js(i)] & xs[jg(i)]);
            else y[i] = sgn(yn, (ms == js(i)) ? xs[jg(i)] : 0u);               // (the pair kernels, two waves per SIMD, have no thirteen registers for the masks)
        }
    }
    // Returns the coordinate to flip (-1: Q11 guard, no flip); reads the state flowed by t (y_i + t: the very doubles flow_and_flip stores).
    __device__ __forceinline__ int velocity_jump(double S, double t, double u) {           // u: the Categorical's uniform
        BananaSh<double> sh{};
        if constexpr (kBanana) sh = tgt.template shared<double>(fma(V(0), t, X(0)), fma(V(1), t, X(1)));
        if (!(S > 0.0)) { guard++; return -1; }                                   // Q11
        const double theta = u * S;
        double carry = 0.0, carry_at = 0.0, lh[PG] = {};
        int ms_hit = -1;
#pragma unroll
        for (int s = 0; s < LK; ++s) {
            double l[PG];
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                const int i = PG * s + g;
                double gv = y[i] + t;                                             // x v at the accepted point
                if constexpr (kBanana) { if (i < 2) gv = twisted(i, gv, sh); }
                l[g] = in_dim(i) ? fmax(0.0, gv) : 0.0;
            }
            const double T = csum(l);
            if (ms_hit < 0 && (carry + T > theta)) {
                ms_hit = s; carry_at = carry;
#pragma unroll
                for (int g = 0; g < PG; ++g) lh[g] = l[g];
            }
            carry = carry + T;
            LANE_SLOT_FENCE(lh, t);
#ifndef LANE_NO_FENCE
            asm volatile("" : "+v"(carry), "+v"(carry_at), "+v"(ms_hit));
#endif
        }
        int m = -1;
        if constexpr (GL == 1) {
            if (ms_hit >= 0) {                                                    // Kogge-Stone inclusive scan of the hit slot, written out
                const double a1 = lh[1] + lh[0], a2 = lh[2] + lh[1], a3 = lh[3] + lh[2];
                const double inc[4] = {lh[0], a1, a2 + lh[0], a3 + a1};
#pragma unroll
                for (int g = 3; g >= 0; --g) if ((4 * ms_hit + g < d) && (carry_at + inc[g] > theta)) m = 4 * ms_hit + g;
            }
        } else {
            // the hit slot's GE rates in both lanes of the pair, then the Kogge-Stone steps (offsets 1, 2 and, for eight, 4) written out
            double l8[GE];
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                const double ph = dpp_d<DPP_QUAD_XOR1>(lh[g]);                    // (zeros while no slot is hit; both lanes of a pair are in here together)
                l8[g] = head() ? lh[g] : ph;
                l8[PG + g] = head() ? ph : lh[g];
            }
            if (ms_hit >= 0) {
                double a[GE], b[GE], c[GE];
#pragma unroll
                for (int g = 0; g < GE; ++g) a[g] = (g >= 1) ? l8[g] + l8[g - 1] : l8[g];
#pragma unroll
                for (int g = 0; g < GE; ++g) b[g] = (g >= 2) ? a[g] + a[g - 2] : a[g];
#pragma unroll
                for (int g = 0; g < GE; ++g) c[g] = (g >= 4) ? b[g] + b[g - 4] : b[g];
#pragma unroll
                for (int g = GE - 1; g >= 0; --g) if ((GE * ms_hit + g < d) && (carry_at + c[g] > theta)) m = GE * ms_hit + g;
            }
        }
        if (m < 0) m = d - 1;
        return m;
    }

    // ---- record!, Composites.jl:239-260, for the lanes with `rec`: X and V rows through LDS, stored as WHOLE lines of LN doubles (64 or 128 bytes;
    // the text below says 8 and 64 bytes where the kernels with LN = 16 have 16 and 128).
    // X[c, k, :] of consecutive events of a chain are contiguous in memory (d doubles each), so a chain writes one continuous
    // stream per array.  An event row of 400 bytes straddles lines; written as it comes, the straddled lines leave the L2 half
    // written and are written twice (profiles: 93.6 GB for 80.7 GB of events).  Here the tail of a row that does not complete
    // a line stays in the chain's LDS row -- the "carry", `cl` doubles -- and goes out in front of the next event's row:
    //   LDS row = [carry | y_0 .. y_{d-1}] = stream positions A .. A + cl + d - 1, A a multiple of 8 doubles;
    //   the first nfull = 8 floor((cl + d) / 8) doubles are stored (lines A / 8 ...), the rest becomes the next carry.
    // Values are kept as (y, sign bit): x = y with the sign flipped where the bit is set, v = +-1 from the same bit; bit i of
    // the chain's 64-bit word S belongs to row position i (carried positions keep the bits they were recorded with).
    // Store (q, r) of a tile of W row positions: lane (grp = lane / (W / 2), sub = lane % (W / 2)) handles the two positions
    // q + 2 sub, + 1 of chain (128 / W) r + grp in one 16-byte store (store_tile2; store_tile: one position, 8 bytes -- the flush)
    // through buffer descriptors of the wave's window of X / V (64 consecutive chains, < 2^31 bytes -- the host checks); a chain that
    // does not record publishes an offset beyond the window and the hardware drops its lanes: no exec masks, no branches in the loop.
    // At the end of a segment the carry is flushed as a partial line; the next segment rebuilds it from the hand-off state.
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    static constexpr uint32_t kNoRow = 0x80000000u;
    __amdgpu_buffer_rsrc_t rsX, rsV;                                              // this unit's windows of p.X and p.V
#ifdef PDMP_STAMP
    unsigned long long st_acc[16], st_last = 0;
    int st_cur = 8;
#endif
    int cl = 0;                                                                   // carried doubles (row positions 0 .. cl - 1 of my chain's LDS row)
    int skip = 0;                                                                 // kShort: the first `skip` positions of the line being filled are not mine to write (open_window)
    SBits Sc{};                                                                   // sign bits of the carried doubles
    // sign bits of the chain's coordinates in coordinate order (row order).  GL = 2: my bits sit at 8 s + g; shifted by 4 h and
    // or-ed with the partner's they are the row's -- all lanes of the wave must be here (DPP).
    __device__ __forceinline__ SBits signs_row() const {
        if constexpr (GL == 1) return ((uint64_t)sg1 << 32) | (uint64_t)sg0;
        else if constexpr (!kWide) {
            const uint64_t o = (((uint64_t)sg1 << 32) | (uint64_t)sg0) << hq;
            const uint32_t q0 = dpp_xor1_u32((uint32_t)o), q1 = dpp_xor1_u32((uint32_t)(o >> 32));
            return o | (((uint64_t)q1 << 32) | (uint64_t)q0);
        } else {
            LaneBits128 o{((uint64_t)sg1 << 32) | (uint64_t)sg0, ((uint64_t)sg3 << 32) | (uint64_t)sg2};
            o = lb_shl(o, hq);
            const uint32_t w0 = (uint32_t)o.lo, w1 = (uint32_t)(o.lo >> 32), w2 = (uint32_t)o.hi, w3 = (uint32_t)(o.hi >> 32);
            const uint32_t q0 = dpp_xor1_u32(w0), q1 = dpp_xor1_u32(w1), q2 = dpp_xor1_u32(w2), q3 = dpp_xor1_u32(w3);
            return {o.lo | (((uint64_t)q1 << 32) | (uint64_t)q0), o.hi | (((uint64_t)q3 << 32) | (uint64_t)q2)};
        }
    }
    __device__ __forceinline__ static SBits low_bits(SBits x, int a) {            // the lowest a <= LN - 1 bits
        if constexpr (!kWide) return x & ((1ull << a) - 1ull);
        else return {x.lo & ((1ull << a) - 1ull), 0ull};
    }
    // my registers into the chain's LDS row whose coordinate 0 is at `row0` (GL = 2: the two lanes of a pair interleave fours)
    __device__ __forceinline__ void row_put(double *row0) const {
        double *w = row0 + hq;
#pragma unroll
        for (int j = 0; j < LD; ++j) w[lbit(j)] = y[j];
    }
    // A segment that starts at event k_first >= 1 of chain c starts a = (address of X[c, k_first, 0] / 8) mod 8 doubles into a 64-byte
    // line -- per chain: round 3 dropped the preconditions "X on a line, n_sk d a multiple of 8" under which a was the same for the
    // whole wave.  Those a doubles in front of it are the tail of the previous event's row, i.e. of the very state the segment starts
    // from (the hand-off state IS the last recorded event; a <= 7 < d): the carry is rebuilt from it and the line is written whole
    // again (the previous segment's flush, or record_initial, wrote the same bytes), so no row position ever has to be skipped.  The
    // one row whose head may share a line with something that is not the chain's -- column 0 -- never comes through here
    // (record_initial); the chain's last line goes out as a partial line (flush_carry).  V uses the same offsets from its own base: it
    // is line-exact when it has X's phase inside a line, correct anywhere.  Called with y, sg loaded, k_first >= 1, by every lane.
    // d <= 6 (kShort): a may exceed d -- the line then starts with rows older than the state (a later segment: the previous segment's
    // flush wrote them) or with bytes that are not the chain's (the first segment; a resumed chunk).  Those a - d positions are skipped:
    // the line is written from the state's row on, once, as a partial line (store_tile's lower check).
    __device__ __forceinline__ void open_window(int64_t b, int64_t c, int64_t k_first) {
        const size_t first = (size_t)b * CW * (size_t)p.n_sk * (size_t)d;
        int64_t chains = p.C - b * CW;
        if (chains > CW) chains = CW;
        const int bytes = (int)(chains * p.n_sk * d * 8);
        // (SkelParams::tflags bits 8 - 10 = PDMP_LANE_DEBUG, measurements only: a window of no bytes drops every store of that array)
        // (one lane per chain only: the pair kernels have no register to spare)
        const int dbg = (GL == 1) ? p.tflags : 0;
        rsX = __builtin_amdgcn_make_buffer_rsrc(p.X + first, 0, (dbg & 0x400) ? 0 : bytes, 0x00020000);
        rsV = __builtin_amdgcn_make_buffer_rsrc(p.V + first, 0, (dbg & 0x200) ? 0 : bytes, 0x00020000);
        const int a = (int)(((uintptr_t)(p.X + ((size_t)c * (size_t)p.n_sk + (size_t)k_first) * (size_t)d) >> 3) & (uintptr_t)(LN - 1));   // per chain
        cl = a;
        {                                                                         // (all lanes: a = 0 moves nothing that is read)
            int l = lane;
            asm volatile("" : "+v"(l));
            double *myrow = tile + (l >> (GL - 1)) * LR;
            __builtin_amdgcn_wave_barrier();
            row_put(myrow + (LN - 1));                                            // the state's coordinates at row positions LN - 1 .. LN - 1 + DMAX - 1
            __builtin_amdgcn_wave_barrier();
            const double *from = myrow + (LN - 1) + (d - a);                      // its last a coordinates
            double mv[LN - 1];
#pragma unroll
            for (int j = 0; j < LN - 1; ++j) mv[j] = from[j];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < LN - 1; ++j) myrow[j] = mv[j];                    // (positions >= a are overwritten by the next row; a pair writes the same values twice)
            if constexpr (kShort) {
                skip = a > d ? a - d : 0;
                Sc = low_bits(a > d ? lb_shl(signs_row(), a - d) : lb_shr(signs_row(), d - a), a);
            } else
            Sc = low_bits(lb_shr(signs_row(), d - a), a);
        }
        tfirst = (int)(((uintptr_t)(p.t + (size_t)c * (size_t)p.n_sk + (size_t)k_first) >> 3) & 7);
        tlast = -1;
    }
    // Event times: 8 bytes per event, eight events of a chain per 64-byte line -- stored one by one each would leave the L2 as a
    // partial line.  They wait in LDS (slot = the time's position in its line, [slot][chain of the wave]) until the line is complete.
    int tfirst = 0, tlast = -1;                                                   // first valid slot of the line being filled; last slot filled (-1: nothing waits)
    typedef double d2 __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ void store_times(double *line, const double *ring, int lo, int hi) {   // slots lo..hi of my line
        if (lo == 0 && hi == 7) {
            d2 *dst = (d2 *)line;
#pragma unroll
            for (int j = 0; j < 4; ++j) { d2 v; v.x = ring[(2 * j) * CW]; v.y = ring[(2 * j + 1) * CW]; dst[j] = v; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j >= lo && j <= hi) line[j] = ring[j * CW];
        }
    }
    // store parameters of a chain: S (GL = 1: two words; GL = 2: four), window offset of row position 0 (kNoRow: nothing to store),
    // nfull << 8 -- 16 bytes (GL = 2: 32, the last 8 unused).  A tile that starts at row position q reads the sign word of q.
    struct TileMeta { uint32_t s, off, nf; };
    __device__ __forceinline__ TileMeta tile_meta(int q, int grp, int ch) const { // chain ch + grp of the wave (ch a constant: folds into the DS offset)
        if constexpr (!kWide) {
            const u32x4 v = ((const u32x4 *)(tile + CW * LR) + grp)[ch];
            return {(q >= 32) ? v.y : v.x, v.z, v.w};
        } else {
            const uint32_t *mw = (const uint32_t *)(tile + CW * LR) + 8 * grp + 8 * ch;
            const u32x2 v = *(const u32x2 *)(mw + 4);
            return {mw[q >> 5], v.x, v.y};
        }
    }
    __device__ __forceinline__ void meta_put(int cwi, SBits S, uint32_t off, uint32_t nf) {
        if constexpr (!kWide) {
            u32x4 m;
            m.x = (uint32_t)S; m.y = (uint32_t)(S >> 32); m.z = off; m.w = nf;
            if (head()) *((u32x4 *)(tile + CW * LR) + cwi) = m;
        } else if (head()) {
            uint32_t *mw = (uint32_t *)(tile + CW * LR) + 8 * cwi;
            u32x4 m;
            m.x = (uint32_t)S.lo; m.y = (uint32_t)(S.lo >> 32); m.z = (uint32_t)S.hi; m.w = (uint32_t)(S.hi >> 32);
            *(u32x4 *)mw = m;
            u32x2 o;
            o.x = off; o.y = nf;
            *(u32x2 *)(mw + 4) = o;
        }
    }
    static constexpr bool defined_lane_no_list() {
#ifdef LANE_NO_LIST            // A/B builds: the checked tile by eight store pairs
        return true;
#else
        return false;
#endif
    }
    // the high word of +-1.0 with the sign of bit 31 of w: (w & 0x80000000) | 0x3ff00000 as one three-input bit operation
    __device__ __forceinline__ static uint32_t one_with_sign(uint32_t w) {
#if __has_builtin(__builtin_amdgcn_bitop3_b32)
        uint32_t k = 0x3ff00000u;
        asm("" : "+v"(k));                              // (in a vector register: the other constant takes the instruction's one scalar operand)
        return __builtin_amdgcn_bitop3_b32(w, 0x80000000u, k, 0xEA);      // (a & b) | c
#else
        return (w & 0x80000000u) | 0x3ff00000u;
#endif
    }
    template <int W, bool LO, bool HI>              // W positions x 64 / W chains per store; LO / HI: check position >= skip / < nfull
    __device__ __forceinline__ void store_tile(int q, int l) {                    // l = lane (opaque copy: see record)
        constexpr int CH = WAVE / W, NR = CW / CH;                                // chains per store: store r covers chains CH r .. CH r + CH - 1
        const int sub = l & (W - 1), grp = l / W;
        const uint32_t pos = (uint32_t)(q + sub);
        const uint32_t sh = 31u - (pos & 31u);                                    // the sign bit of position q + sub, moved to bit 31
        const double *tb = tile + grp * LR + pos;                                 // one base address per array; the chain of store r lies a
                                                                                  // constant further on (folds into the DS offset)
        // LDS reads of LANE_BATCH stores are issued together, then their stores: one wave per SIMD runs in order, and a read that
        // is waited for before every store costs its ~100-cycle round trip 56 times per event row
        constexpr int NB = NR < LANE_BATCH ? NR : LANE_BATCH;
#pragma unroll
        for (int r0 = 0; r0 < NR; r0 += NB) {
            TileMeta m[NB];
            double yv[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { m[b] = tile_meta(q, grp, CH * (r0 + b)); yv[b] = tb[CH * (r0 + b) * LR]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                uint32_t off = m[b].off + pos * 8u;
                if constexpr (LO) off = (pos >= (m[b].nf & 0xffu)) ? off : kNoRow;
                if constexpr (HI) off = (pos < (m[b].nf >> 8)) ? off : kNoRow;
                const uint32_t ws = m[b].s << sh;
                u32x2 xv, vv;
                xv.x = (uint32_t)__double2loint(yv[b]); xv.y = (uint32_t)__double2hiint(yv[b]) ^ (ws & 0x80000000u);
                vv.x = 0u; vv.y = one_with_sign(ws);
                __builtin_amdgcn_raw_buffer_store_b64(xv, rsX, (int)off, 0, LANE_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(vv, rsV, (int)off, 0, LANE_STORE_AUX);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // The same with TWO consecutive positions per lane and 16-byte stores (half the store instructions): tiles that need no lower
    // (a pair of positions (even, odd) lies on one side of nfull, a multiple of 8).  W positions per
    // tile, W / 2 lanes per chain, 128 / W chains per store.
    template <int W, bool HI>
    __device__ __forceinline__ void store_tile2(int q, int l) {
        constexpr int LPC = W / 2, CH = WAVE / LPC, NR = CW / CH;                 // lanes per chain, chains per store, stores per tile
        const int sub = l & (LPC - 1), grp = l / LPC;
        const uint32_t pos = (uint32_t)(q + 2 * sub);                             // even; pos and pos + 1 share a sign word
        const uint32_t sh = 31u - (pos & 31u);
        const double *tb = tile + grp * LR + pos;
        constexpr int NB = NR < LANE_BATCH ? NR : LANE_BATCH;
#pragma unroll
        for (int r0 = 0; r0 < NR; r0 += NB) {
            TileMeta m[NB];
            double ya[NB], yb[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { m[b] = tile_meta(q, grp, CH * (r0 + b)); ya[b] = tb[CH * (r0 + b) * LR]; yb[b] = tb[CH * (r0 + b) * LR + 1]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                uint32_t off = m[b].off + pos * 8u;
                if constexpr (HI) off = (pos < (m[b].nf >> 8)) ? off : kNoRow;
                const uint32_t w = m[b].s;
                const uint32_t wa = w << sh, wb = w << (sh - 1u);                      // the two positions' sign bits at bit 31
                u32x4 xv, vv;
                // x: the sign bit flips y's (v_bitop3_b32, which the compiler finds by itself); v: +-1.0's high word = (bit 31 of wa) | 0x3ff00000 --
                // one v_bitop3_b32 with its two constants in a scalar and a vector register where the compiler emits v_and + v_or + a copy
                // (gfx9 encodes no literal in a three-operand instruction): 10 -> 7 vector instructions per pair of stores, 32 pairs per event row
                xv.x = (uint32_t)__double2loint(ya[b]); xv.y = (uint32_t)__double2hiint(ya[b]) ^ (wa & 0x80000000u);
                xv.z = (uint32_t)__double2loint(yb[b]); xv.w = (uint32_t)__double2hiint(yb[b]) ^ (wb & 0x80000000u);
                vv.x = 0u; vv.y = one_with_sign(wa); vv.z = 0u; vv.w = one_with_sign(wb);
                __builtin_amdgcn_raw_buffer_store_b128(xv, rsX, (int)off, 0, LANE_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(vv, rsV, (int)off, 0, LANE_STORE_AUX);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // The one checked tile of a row of 128-byte lines (positions q .. q + 15, whole or not there at all: nfull = q or q + 16) for the chains
    // that need it ONLY.  At d = 50 one row in eight completes a fourth line, but the chains of a wave are at eight different phases, so every
    // one of the eight store pairs of store_tile2<16, true> has a chain to serve and seven lane groups out of eight carry a dropped offset:
    // ~100 vector instructions per pass.  Here the chains that need the tile are listed (ballot, mbcnt: their ranks; 64 bytes of LDS) and
    // store pair r serves the chains of ranks 8 r .. 8 r + 7 -- one pair instead of eight on average; when most chains need the tile (rows that
    // are a whole number of lines: d = 48, every row) the direct form is the cheaper one again.
    __device__ __forceinline__ void store_tile_listed(int q, int l, bool need) {
        static_assert(GL == 1 && !kWide, "one lane per chain, 64 sign bits per row");
        const unsigned long long mask = __ballot(need);
        const int n = __builtin_popcountll(mask);
        if (n == 0) return;
        if (n > 40) { store_tile2<16, true>(q, l); return; }
        unsigned char *list = (unsigned char *)(tile + CW * (LR + MD + 8));
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        __builtin_amdgcn_wave_barrier();
        if (need) list[rank] = (unsigned char)l;
        __builtin_amdgcn_wave_barrier();
        const int sub = l & 7, grp = l >> 3;
        const uint32_t pos = (uint32_t)(q + 2 * sub);
        const uint32_t sh = 31u - (pos & 31u);
        for (int r0 = 0; r0 < n; r0 += 8) {                                       // (wave-uniform)
            const int slot = r0 + grp;
            const bool ok = slot < n;
            const int ch = (int)list[ok ? slot : 0];
            const u32x4 mv = ((const u32x4 *)(tile + CW * LR))[ch];               // S (two words), window offset of row position 0, nfull << 8
            const double *tb = tile + ch * LR + pos;
            const double ya = tb[0], yb = tb[1];
            const uint32_t off = ok ? mv.z + pos * 8u : kNoRow;
            const uint32_t w = (q >= 32) ? mv.y : mv.x;
            const uint32_t wa = w << sh, wb = w << (sh - 1u);
            u32x4 xv, vv;
            xv.x = (uint32_t)__double2loint(ya); xv.y = (uint32_t)__double2hiint(ya) ^ (wa & 0x80000000u);
            xv.z = (uint32_t)__double2loint(yb); xv.w = (uint32_t)__double2hiint(yb) ^ (wb & 0x80000000u);
            vv.x = 0u; vv.y = one_with_sign(wa); vv.z = 0u; vv.w = one_with_sign(wb);
            __builtin_amdgcn_raw_buffer_store_b128(xv, rsX, (int)off, 0, LANE_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(vv, rsV, (int)off, 0, LANE_STORE_AUX);
        }
    }
    __device__ __forceinline__ void record(bool rec, int64_t c, int32_t k, double t, double horizon, double ar, int32_t errored,
                                           const double (&ev)[5], int32_t rejected, int32_t hits) {
        int l = lane;
        asm volatile("" : "+v"(l));               // LDS addresses are rebuilt here from an opaque copy of the lane id: hoisted out of the
                                                  // event loop they would occupy ~50 registers for the whole launch
        const int cwi = l >> (GL - 1);            // my chain's row
        double *myrow = tile + cwi * LR;
        const int total = cl + d, nfull = total & ~(LN - 1);
        const SBits S = lb_or(Sc, lb_shl(signs_row(), cl));
        const uint32_t off = rec ? (((uint32_t)cwi * (uint32_t)p.n_sk + (uint32_t)k) * (uint32_t)d - (uint32_t)cl) * 8u : kNoRow;
        __builtin_amdgcn_wave_barrier();
        LANE_MARK(9);
        meta_put(cwi, S, off, ((uint32_t)nfull << 8) | (kShort ? (uint32_t)skip : 0u));
        row_put(myrow + cl);
        __builtin_amdgcn_wave_barrier();
        LANE_MARK(10);
        // nfull = 8 floor((cl + d) / 8) lies in [NF_MIN, NF_MAX]: whole tiles of 16 positions below NF_MIN need no check, the rest goes out as
        // tiles of 8 positions checked against nfull
        constexpr int NF_MIN = LANE_DMIN & ~(LN - 1), NF_MAX = (DMAX + LN - 1) & ~(LN - 1), Q0 = NF_MIN & ~15;
        if constexpr (kShort) {      // one tile of eight positions, checked on both sides (a pair of positions may straddle `skip`: 8-byte stores)
            static_assert(Q0 == 0 && NF_MAX == 8, "short rows: one line per event at most");
            store_tile<8, true, true>(0, l);
        } else {
#ifdef LANE_STORE_B64              // A/B builds: 8-byte stores
#pragma unroll
        for (int q = 0; q < Q0; q += 16) store_tile<16, false, false>(q, l);
        LANE_MARK(13);
#pragma unroll
        for (int q = Q0; q < NF_MAX; q += LN) store_tile<LN, false, true>(q, l);
#else
#pragma unroll
        for (int q = 0; q < Q0; q += 16) store_tile2<16, false>(q, l);
        LANE_MARK(13);
        if constexpr (LN == 16 && GL == 1 && !defined_lane_no_list()) {
            static_assert(NF_MAX - Q0 == 16, "128-byte lines: one checked tile per row");
            store_tile_listed(Q0, l, rec && nfull > Q0);
        } else {
#pragma unroll
        for (int q = Q0; q < NF_MAX; q += LN) store_tile2<LN, true>(q, l);
        }
#endif
        }
        __builtin_amdgcn_wave_barrier();
        LANE_MARK(14);
        // (Reading y back from the row here -- so that the compiler need not park half of it in accumulation registers across the tile stores, ~100
        //  v_accvgpr moves per pass -- was tried: 56 vector instructions fewer per pass, 26 LDS reads more, and the launch 13.61 -> 13.86 ms: a wave
        //  alone on its SIMD waits out every LDS round trip.)
        if (rec) {
            if constexpr (kShort) { if (nfull) skip = 0; }
            const double *from = myrow + nfull;                                   // the row's tail that completes no line: the next carry
            double mv[LN - 1];
#pragma unroll
            for (int j = 0; j < LN - 1; ++j) mv[j] = from[j];
#pragma unroll
            for (int j = 0; j < LN - 1; ++j) myrow[j] = mv[j];
            // 128-byte lines: the row (15 + 52 positions at most) may be longer than the 64 bits of S, but what is stored of it is not
            // (nfull <= 64), and the new carry lies in the row's own coordinates (nfull >= 48 > cl): its bits come from their word
            if constexpr (LN == 16) Sc = lb_shr(signs_row(), nfull - cl);
            else
            Sc = lb_shr(S, nfull);
            cl = total - nfull;
            const size_t e = (size_t)c * (size_t)p.n_sk + (size_t)k;
            const int slot = (int)(((uintptr_t)(p.t + e) >> 3) & 7);
            if (head()) {
                double *ring = tile + CW * LR + MD * CW + cwi;
                ring[slot * CW] = t;
                if (slot == 7 && !(GL == 1 && (p.tflags & 0x100))) store_times(p.t + e - 7, ring, tfirst, 7);
                if constexpr (DIAG) {
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
            tlast = slot;
            if (slot == 7) { tfirst = 0; tlast = -1; }
        }
        LANE_MARK(15);
    }
    // Column 0 (sample.jl:272; a resumed chunk: the previous chunk's last event again): each lane its own coordinates, straight from the
    // registers -- once per chain, and the only row whose head can share a 64-byte line with bytes that are not the chain's (the
    // previous chain's last row, or whatever precedes X): nothing but the row itself is written.
    __device__ __forceinline__ void record_initial(bool rec, int64_t c, double t, double horizon, double ar) {
        const size_t e = (size_t)c * (size_t)p.n_sk;
        double *Xe = p.X + e * (size_t)d, *Ve = p.V + e * (size_t)d;
        if (rec) {
#pragma unroll
            for (int i = 0; i < LD; ++i) if (in_dim(i)) { Xe[gi(i)] = X(i); Ve[gi(i)] = V(i); }
        }
        if (rec && head()) {
            p.t[e] = t;
            if constexpr (DIAG) {
                if (p.H) p.H[e] = horizon;
                if (p.AR) p.AR[e] = ar;
                if (p.EB) p.EB[e] = 0;
                if (p.RJ) p.RJ[e] = 0;
                if (p.HH) p.HH[e] = 0;
                if (p.EV) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) p.EV[5 * e + q] = 0.0;
                }
            }
        }
    }
    // end of a segment: the carried doubles go out as the head of a line (the next segment writes that line whole again); k = the next event
    __device__ __forceinline__ void flush_carry(bool mine, int64_t c, int32_t k) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int cwi = l >> (GL - 1);
        const uint32_t off = (mine && cl > 0) ? (((uint32_t)cwi * (uint32_t)p.n_sk + (uint32_t)k) * (uint32_t)d - (uint32_t)cl) * 8u : kNoRow;
        __builtin_amdgcn_wave_barrier();
        meta_put(cwi, Sc, off, ((uint32_t)cl << 8) | (kShort ? (uint32_t)skip : 0u));
        __builtin_amdgcn_wave_barrier();
        store_tile<LN, kShort, true>(0, l);
        __builtin_amdgcn_wave_barrier();
        cl = 0; skip = 0;
        if (mine && head() && tlast >= 0)                                         // the event times of an incomplete line; k - 1 = my last event, in slot tlast
            store_times(p.t + ((size_t)c * (size_t)p.n_sk + (size_t)k - 1 - (size_t)tlast), tile + CW * LR + MD * CW + cwi, tfirst, tlast);
        tlast = -1;
    }

    // ---- GroupChain::run_segment for the CW chains of this wave.  Hand-off state (private to this kernel): word w of chain cs at
    // state[((cs / 64) (2d+4) + w) 64 + cs % 64] -- y_0..y_{d-1}, the sign words (GL = 1: one double; GL = 2: two per lane of the
    // pair, each lane its own), t, horizon, ar, the stream position.
    __device__ __forceinline__ void run_segment(int64_t b, int64_t c, bool chain_ok, int r, int64_t seg_len, double *state, int budget, int nseg = 1) {
        const int64_t k_begin = 1 + (int64_t)r * seg_len;
        int64_t k_end = k_begin + seg_len;
        if (k_end > p.n_sk) k_end = p.n_sk;
        // cs: the chain's index in the ORIGINAL ensemble (state slot: batch cs / 64, lane cs % 64; initial state; random stream);
        // c = 64 b + lane: its index in this launch (outputs).  A padding lane of a ragged batch uses its own launch slot.
        // (The time-horizon features -- chain map, resume, stop at t_stop, column counts -- live in the DIAG instantiations only: the
        // throughput kernels keep the registers and instructions they had; the host routes every such launch to a DIAG kernel.)
        int64_t cs = c;
        if constexpr (DIAG) { if (p.chain_map && chain_ok) cs = p.chain_map[c]; }
        double *st = state + (size_t)(cs >> 6) * (size_t)lane_state_words(d) * WAVE + (size_t)(cs & 63);
        double t = 0.0, horizon = p.tmax, ar = 0.0;
        uint64_t n0 = 0;
        sg0 = 0; sg1 = 0; sg2 = 0; sg3 = 0;
        constexpr int NSD = kWide ? 2 : 1;                                        // doubles of sign words per lane
        const int w_sg = d + (GL == 1 ? 0 : NSD * (hq / PG)), w_sc = d + NSD * GL;   // my sign words; t, horizon, ar, stream position
        bool from_state = r > 0;
        if constexpr (DIAG) from_state = from_state || p.resume != 0;
        if (from_state) {
#pragma unroll
            for (int i = 0; i < LD; ++i) y[i] = in_dim(i) ? st[(size_t)gi(i) * WAVE] : 0.0;
            const uint64_t sgw = (uint64_t)__double_as_longlong(st[(size_t)w_sg * WAVE]);
            sg0 = (uint32_t)sgw; sg1 = (uint32_t)(sgw >> 32);
            if constexpr (kWide) {
                const uint64_t sgx = (uint64_t)__double_as_longlong(st[(size_t)(w_sg + 1) * WAVE]);
                sg2 = (uint32_t)sgx; sg3 = (uint32_t)(sgx >> 32);
            }
            t = st[(size_t)w_sc * WAVE]; horizon = st[(size_t)(w_sc + 1) * WAVE]; ar = st[(size_t)(w_sc + 2) * WAVE];
            n0 = (uint64_t)__double_as_longlong(st[(size_t)(w_sc + 3) * WAVE]);
        } else {
#pragma unroll
            for (int i = 0; i < LD; ++i) {
                const bool ok = chain_ok && in_dim(i);
                const double xi = ok ? p.xinit[(size_t)cs * d + gi(i)] : 0.0;
                const double vi = ok ? p.vinit[(size_t)cs * d + gi(i)] : 1.0;
                const uint32_t s = (uint32_t)__double2hiint(vi) & 0x80000000u;
                y[i] = sgn(xi, s);
                const int lb = lbit(i);
                if (lb < 32) sg0 |= (s >> 31) << (lb & 31); else if (lb < 64) sg1 |= (s >> 31) << (lb & 31);
                else if (lb < 96) sg2 |= (s >> 31) << (lb & 31); else sg3 |= (s >> 31) << (lb & 31);
            }
        }
        if (r == 0) record_initial(chain_ok, c, t, horizon, ar);                  // (k_begin = 1: the segment's window opens behind column 0)
        open_window(b, c, k_begin);
        rng.init_lane(p.seed, (uint64_t)(p.chain0 + cs));
        rng.restore(n0);
        if constexpr (TAPE) {
            const size_t gc = (size_t)(p.chain0 + (chain_ok ? cs : 0));
            tp_e = p.tape_exp + gc * (size_t)p.tape_n_exp;
            tp_u = p.tape_uni + gc * (size_t)p.tape_n_uni;
            tp_ie = 0; tp_iu = 0; tp_over = false;
        }
        double ts = 0.0, tp = 0.0, E = 0.0, lb = 0.0;
        int32_t errored = 0, rejected = 0, hits = 0;
        double ev[5] = {0, 0, 0, 0, 0};
        const bool adaptive = p.adaptive != 0;
        bool done = !chain_ok;
        int32_t k = (int32_t)k_begin, it = 0;
        const int32_t cap = (int32_t)p.cap, kend = (int32_t)k_end;
        bool fin_before = false, stopped = false;                                 // time-horizon variant: `while state.t < T`, sample.jl:372 (as GroupChain)
        if constexpr (DIAG) { fin_before = !(t < p.t_stop); if (fin_before) done = true; }
        if (k >= kend) done = true;
        enum : int { FRESH = 0, ERR = 1, REUSE = 2 };
        bool propose = false, brun = false;
        int mode = FRESH;
        double hB = horizon;
#ifdef PDMP_STAMP
        for (int q = 0; q < 16; ++q) st_acc[q] = 0;
        st_last = __builtin_amdgcn_s_memtime();
        st_cur = 8;
#endif
        while (__ballot(!done) != 0ull) {
            bool accepted = false, proposed = false;
            double tf = -0.0;
            int m = -1;
            LANE_MARK(8);
            if constexpr (TAPE) { if (tp_over) done = true; }                     // tape exhausted: stop this chain (status[3] below)
            const Prefetch pf = rng_prefetch();
            LANE_MARK(0);
            if (!done && !propose) {
                if (!brun) {
                    const double harg = (mode == ERR) ? horizon / 2 : ((mode == REUSE) ? hB : horizon);
                    hB = harg;
                    double e1;
                    if constexpr (TAPE) e1 = tape_exponential(); else e1 = -det_log(draw_uniform(pf));
                    E = (mode == REUSE) ? E + e1 : e1;
                    if (mode == REUSE) bound_resume(tp); else bound_begin(tp);
                    brun = true;
                }
                LANE_MARK(1);
                if (bound_advance(hB, E, tp, lb, budget)) {
                    LANE_MARK(2);
                    const double harg = hB;
                    brun = false;
                    if (mode == ERR) {
                        if (adaptive) horizon = harg;
                        errored++;
                        const int slot = errored % 5;
#pragma unroll
                        for (int q = 0; q < 5; ++q) if (q == slot) ev[q] = ar;
                        propose = tp < horizon;
                    } else if (mode == REUSE) {
                        if (adaptive) horizon = horizon / 1.04;
                        rejected++;
                        if (tp > horizon) { tf = horizon; ts += horizon; hits++; }
                        propose = tp < horizon;
                    } else if (tp > horizon) {
                        tf = horizon;
                        ts += horizon;
                        hits++;
                        if (adaptive) horizon = horizon * 1.01;
                    } else {
                        propose = tp < horizon;
                    }
                    mode = FRESH;
                    if (++it > cap) { done = true; if (head()) atomicAdd(&p.status[0], 1); }
                }
            }
            LANE_MARK(3);
            if (!done && propose) {
                const double lam = true_rate(tp);
                LANE_MARK(4);
                ar = lam / lb;
                propose = false;
                if (ar > 1.0) {
                    mode = ERR;
                } else {
                    const double u = draw_uniform(pf);
                    if (u < ar) {
                        tf = tp;
                        m = velocity_jump(lam, tp, draw_uniform(pf));
                        accepted = true;
                    } else {
                        mode = REUSE;
                    }
                }
                proposed = true;
            }
            LANE_MARK(5);
            flow_and_flip(tf, m);
            LANE_MARK(6);
            if (accepted) { t = t + tp + ts; ts = 0.0; tp = 0.0; }
            if (__ballot(accepted) != 0ull) record(accepted, c, k, t, horizon, ar, errored, ev, rejected, hits);
            if (accepted) {
                errored = 0; rejected = 0; hits = 0;
#pragma unroll
                for (int q = 0; q < 5; ++q) ev[q] = 0.0;
                it = 0;
                if (++k >= kend) done = true;
                if constexpr (DIAG) { if (!(t < p.t_stop)) { done = true; stopped = true; } }
            }
            LANE_MARK(7);
            if (proposed && ++it > cap && !done) { done = true; if (head()) atomicAdd(&p.status[0], 1); }
        }
        LANE_MARK(8);
#ifdef PDMP_STAMP
        // slots: 0 E draw, 1 bound cells, 2 bound epilogue, 3 -> true_rate, 4 ar / accept / walk, 5 flow + flip, 6 record, 7 loop tail, 8 prefetch
        if (lane == 0 && p.stamps) {
            atomicAdd(&p.stamps[0], st_acc[8] + st_acc[0]);                            // Philox prefetch + exponential
            atomicAdd(&p.stamps[1], st_acc[1] + st_acc[2]);                            // bound cells + epilogue
            atomicAdd(&p.stamps[2], st_acc[3] + st_acc[4]);                            // true rate, ar, accept, Categorical walk
            atomicAdd(&p.stamps[3], st_acc[5]);                                        // flow + flip
            atomicAdd(&p.stamps[4], st_acc[6] + st_acc[9]);                            // record: row and parameters to LDS
            atomicAdd(&p.stamps[5], st_acc[10] + st_acc[11] + st_acc[12] + st_acc[13]); // record: the four tiles (LDS reads, stores)
            atomicAdd(&p.stamps[6], st_acc[14]);                                       // record: carry, event times, diagnostics
            atomicAdd(&p.stamps[7], st_acc[15] + st_acc[7]);                           // loop tail
        }
#endif
        flush_carry(chain_ok, c, k);
        // hand-off state (write-through stores, as the group kernel's).  Real chains only: with a chain map the slot a padding lane
        // would address belongs to another chain of the original ensemble.
        if (chain_ok) {
#pragma unroll
            for (int i = 0; i < LD; ++i) if (in_dim(i)) publish(st + (size_t)gi(i) * WAVE, y[i]);
            publish(st + (size_t)w_sg * WAVE, __longlong_as_double((long long)(((uint64_t)sg1 << 32) | (uint64_t)sg0)));
            if constexpr (kWide) publish(st + (size_t)(w_sg + 1) * WAVE, __longlong_as_double((long long)(((uint64_t)sg3 << 32) | (uint64_t)sg2)));
            if (head()) {
                publish(st + (size_t)w_sc * WAVE, t); publish(st + (size_t)(w_sc + 1) * WAVE, horizon); publish(st + (size_t)(w_sc + 2) * WAVE, ar);
                publish(st + (size_t)(w_sc + 3) * WAVE, __longlong_as_double((long long)rng.n));
            }
        }
        if (guard) { if (head()) atomicAdd(&p.status[1], guard); guard = 0; }
        if constexpr (TAPE) { if (tp_over && chain_ok) atomicAdd(&p.status[3], 1); }
        if constexpr (DIAG) {
            if (p.count && chain_ok && head() && !fin_before && (stopped || r == nseg - 1)) p.count[c] = (int64_t)k;
            if (p.count && chain_ok && head() && fin_before && r == 0) p.count[c] = 1;
        }
    }
};

template <int TARGET, int LK, bool DIAG, bool TAPE = false, int GL = 1, int PG = 4>
__global__ __launch_bounds__(256, ((PG == 2 && !DIAG) ? 2 : LANE_WAVES)) void lane_kernel(const GroupParams gp, const int32_t *flag) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // not unit speeds: the group kernel runs instead
    extern __shared__ double smem[];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    double *tile = smem + (size_t)wave * lane_lds_per_wave(LK, GL, PG);
    LaneChain<TARGET, LK, DIAG, TAPE, GL, PG> chain(gp.s, lane, tile);
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
            const int64_t c = (int64_t)b * (WAVE / GL) + (lane >> (GL - 1));
            chain.run_segment(b, c, c < gp.s.C, r, gp.seg_len, gp.state, gp.cell_budget, gp.nseg);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&gp.progress[b], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lost && lane == 0) atomicAdd(&gp.s.status[2], 1);
}

#ifndef __HIPCC_RTC__
// segments per chain: ~`rounds` work units per resident wave slot (the units of a launch differ in length; 10^5 chains are 1 563
// batches for 2 048 slots, so one unit per batch would leave a quarter of the SIMDs with half the work of the others)
inline void lane_schedule(int64_t nb, int64_t slots, int64_t n_sk, int force, int rounds, int32_t &nseg, int64_t &seg_len) {
    const int64_t events = n_sk > 1 ? n_sk - 1 : 1;
    int64_t want = nb > slots ? (rounds * slots + nb - 1) / nb : 1;       // every batch resident at once: nothing to balance (9.6 -> 9.2 ms at 50 000 chains)
    if (force > 0) want = force;
    const int64_t max_seg = events / 16 > 0 ? events / 16 : 1;
    if (force <= 0 && want > max_seg) want = max_seg;
    if (want > events) want = events;
    if (want < 1) want = 1;
    seg_len = (events + want - 1) / want;
    nseg = (int32_t)((events + seg_len - 1) / seg_len);
}

template <int TARGET, int LK, bool DIAG, bool TAPE = false, int GL = 1, int PG = 4>
inline hipError_t launch_lane(const GroupParams &gp_in, const int32_t *flag, int rounds, hipStream_t stream) {
    GroupParams gp = gp_in;
    constexpr int wpb = 4;
    static_assert((size_t)wpb * lane_lds_per_wave(LK, GL, PG) * sizeof(double) <= 160u * 1024u, "a block of four waves must fit the CU's 160 KB of LDS (the widest rows: 15 carried + 52 coordinates, 157.9 KB)");
    const size_t lds = (size_t)wpb * lane_lds_per_wave(LK, GL, PG) * sizeof(double);
    void (*kernel)(const GroupParams, const int32_t *) = lane_kernel<TARGET, LK, DIAG, TAPE, GL, PG>;
    int per_cu = 0, dev = 0, cus = 0;
    hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE * wpb, lds);
    if (e != hipSuccess) return e;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (per_cu < 1) per_cu = 1;
    if (cus < 1) cus = 256;
    gp.nb = (gp.s.C + WAVE / GL - 1) / (WAVE / GL);
    const int64_t slots = (int64_t)cus * per_cu * wpb;
    lane_schedule(gp.nb, slots, gp.s.n_sk, gp.nseg, rounds, gp.nseg, gp.seg_len);
    int64_t blocks = (gp.nb * gp.nseg + wpb - 1) / wpb;
    if (blocks > (int64_t)cus * per_cu) blocks = (int64_t)cus * per_cu;
    if (gp.host_flags & 2) fprintf(stderr, "lane launch GL=%d PG=%d LK=%d: blocks per CU %d, CUs %d, LDS %zu B / block, batches %lld x segments %d -> %lld blocks\n", GL, PG, LK, per_cu, cus, lds, (long long)gp.nb, (int)gp.nseg, (long long)blocks);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(WAVE * wpb), lds, stream, gp, flag);
    return hipGetLastError();
}
// one translation unit per range of LK (inst_lane_zigzag*.hip); hipErrorNotSupported: not this unit's LK
template <int TARGET, int LK0, int LK1, int GL = 1, int PG = 4>
inline hipError_t launch_lane_by_slots(int lk, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    if constexpr (LK0 > LK1) return hipErrorNotSupported;
    else {
        const bool diag = gp.s.H || gp.s.AR || gp.s.EB || gp.s.RJ || gp.s.HH || gp.s.EV ||
                          gp.s.count || gp.s.chain_map || gp.s.resume || gp.s.t_stop < __builtin_huge_val();   // (the time-horizon features: DIAG kernels)
        return lk != LK0 ? launch_lane_by_slots<TARGET, LK0 + 1, LK1, GL, PG>(lk, gp, flag, rounds, stream)
                         : (diag ? launch_lane<TARGET, LK0, true, false, GL, PG>(gp, flag, rounds, stream) : launch_lane<TARGET, LK0, false, false, GL, PG>(gp, flag, rounds, stream));
    }
}
template <int LK0, int LK1, int GL = 1, int PG = 4>
inline hipError_t launch_lane_range(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream) {
    const int lk = (gp.s.d + PG * GL - 1) / (PG * GL);
    switch (target) {
    case T_GAUSS_ISO: return launch_lane_by_slots<T_GAUSS_ISO, LK0, LK1, GL, PG>(lk, gp, flag, rounds, stream);
    case T_BANANA: return launch_lane_by_slots<T_BANANA, LK0, LK1, GL, PG>(lk, gp, flag, rounds, stream);
    default: return hipErrorNotSupported;
    }
}
hipError_t launch_lane_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);       // LK = 13
hipError_t launch_lane_zigzag_a(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);     // LK = 7 .. 9
hipError_t launch_lane_zigzag_b(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);     // LK = 10 .. 12
hipError_t launch_lane_zigzag_c(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);     // LK = 3 .. 6 (d = 9 .. 24)
hipError_t launch_lane_zigzag_d(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);     // LK = 1, 2 (d = 1 .. 8)
hipError_t launch_lane_zigzag_tape(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);  // LK = 13, tape source
hipError_t launch_lane_check(const double *vinit, int64_t n, int32_t *flag, hipStream_t stream);
// a pair of lanes per chain (GL = 2): d = 53 .. 104, LK = ceil(d / 8)
hipError_t launch_lane2_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);      // LK = 13
hipError_t launch_lane2_zigzag_a(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);    // LK = 7 .. 9
hipError_t launch_lane2_zigzag_b(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);    // LK = 10 .. 12
// a pair of lanes per chain with two coordinates of a slot each (GL = 2, PG = 2): d = 25 .. 52, LK = ceil(d / 4), mid-sized launches
hipError_t launch_lane2q_zigzag(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);     // LK = 7 .. 13
hipError_t launch_lane2q_zigzag_c(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);   // LK = 3 .. 6
hipError_t launch_lane2q_zigzag_d(int target, const GroupParams &gp, const int32_t *flag, int rounds, hipStream_t stream);   // LK = 1, 2
#endif

PDMP_NS_END
