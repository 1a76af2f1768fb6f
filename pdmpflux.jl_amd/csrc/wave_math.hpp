// wave_math.hpp -- wave64 building blocks of the gfx950 PDMP engine: counter-based random stream,
// bit-reproducible elementary functions, fixed-order wave reductions/scans, forward-mode duals.
//
// Every floating-point expression here is part of the engine's numerical specification
// (DESIGN.md "Arithmetic contract"): compiled with -ffp-contract=off, no fast-math, so that results
// are bit-identical to the scalar CPU checker run in its WAVE64 summation order.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else   // runtime compilation of a user potential (user_jit.hip): hiprtc brings the HIP device API, not <stdint.h>
typedef unsigned char uint8_t;
typedef signed int int32_t;
typedef unsigned int uint32_t;
typedef signed long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
typedef unsigned long uintptr_t;
#endif

// ---------------------------------------------------------------- arithmetic contract version of a translation unit
// PDMP_RNG_VERSION (DESIGN.md "Arithmetic contract v2"; pdmp_config_t.rng_version): the random stream and the elementary functions the
// kernels of this translation unit are compiled with.  2 (default since round 5): Philox4x32-7, log / exp / sin / cos as Horner
// schemes on fma, octant Box-Muller.  1: Philox4x32-10 and the fdlibm forms without fma (rounds 1-4, the fixtures frozen then) --
// built for the LDS-form group kernels and the wave-per-chain kernel only (csrc/Makefile: v1/*.o).  Every name of the engine lives in
// an inline namespace per version so that the two builds of one kernel template are two symbols.
#ifndef PDMP_RNG_VERSION
#define PDMP_RNG_VERSION 2
#endif
#if PDMP_RNG_VERSION == 1
#define PDMP_NS_BEGIN namespace pdmp { inline namespace av1 {
#else
#define PDMP_NS_BEGIN namespace pdmp { inline namespace av2 {
#endif
#define PDMP_NS_END } }

PDMP_NS_BEGIN

constexpr int WAVE = 64;
constexpr int kRngVersion = PDMP_RNG_VERSION;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// ---------------------------------------------------------------- cross-lane helpers
__device__ __forceinline__ double readlane_d(double x, int l) {       // l wave-uniform
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readfirst_d(double x) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// DPP lane permutations inside a row of 16 lanes (no LDS traffic).  After the xor-1 and xor-2
// stages all four lanes of a quad hold the same value, so row_half_mirror (i <-> 7-i) is a valid
// xor-4 partner and row_mirror (i <-> 15-i) a valid xor-8 partner: same operands, same sums.
template <int CTRL>
__device__ __forceinline__ double dpp_d(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;     // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

// The xor-16 and xor-32 stages of a butterfly WITHOUT the LDS crossbar (ds_bpermute: two instructions per double, an address, a
// wait of ~120 cycles): gfx950's v_permlane16_swap / v_permlane32_swap exchange whole rows / halves between two registers.
// With x = y = a:  permlane32_swap leaves x = [a(0..31), a(0..31)], y = [a(32..63), a(32..63)]; permlane16_swap leaves
// x = [r0, r0, r2, r2], y = [r1, r1, r3, r3] (rows of 16 lanes).  x + y is then, in every lane, a[lane] + a[lane ^ 32] (^ 16): the same
// two operands as `a + __shfl_xor(a, 32)` -- IEEE addition is commutative, so the same bits.
__device__ __forceinline__ double xor32_sum(double a) {
    const int lo = __double2loint(a), hi = __double2hiint(a);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
}
__device__ __forceinline__ double xor16_sum(double a) {
    const int lo = __double2loint(a), hi = __double2hiint(a);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
}

// Sum over the 64 lanes, result in every lane.  Order (the "WAVE64" order of the checker):
// xor-butterfly with offsets 1,2,4,8,16,32; IEEE addition is commutative so both partners of a pair
// compute the same bits.
__device__ __forceinline__ double wave_sum(double a) {
    a = a + dpp_d<DPP_QUAD_XOR1>(a);
    a = a + dpp_d<DPP_QUAD_XOR2>(a);
    a = a + dpp_d<DPP_ROW_HALF_MIRROR>(a);
    a = a + dpp_d<DPP_ROW_MIRROR>(a);
    a = xor16_sum(a);
    a = xor32_sum(a);
    return a;
}

// Kogge-Stone inclusive prefix sum over the 64 lanes.
__device__ __forceinline__ double wave_scan_incl(double a, int lane) {
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        double t = __shfl_up(a, off, WAVE);
        if (lane >= off) a = a + t;
    }
    return a;
}

// ---------------------------------------------------------------- forward-mode dual numbers
// (val, eps): eps carries d/dt along the flow -- the engine's replacement for
// ForwardDiff.derivative in UpperBound.jl:102,210.  Product rule without fused multiply-add.
struct Dual {
    double v, e;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.e + b.e}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.e - b.e}; }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return {a.v - b, a.e}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.e * b.v + a.v * b.e}; }
__device__ __forceinline__ Dual operator*(double s, Dual a) { return {s * a.v, s * a.e}; }
__device__ __forceinline__ Dual operator*(Dual a, double s) { return {a.v * s, a.e * s}; }
__device__ __forceinline__ Dual operator+(Dual a, double b) { return {a.v + b, a.e}; }
__device__ __forceinline__ Dual operator+(double a, Dual b) { return {a + b.v, b.e}; }
__device__ __forceinline__ Dual operator-(double a, Dual b) { return {a - b.v, -b.e}; }
__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.e}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const double q = a.v / b.v; return {q, (a.e - q * b.e) / b.v}; }
__device__ __forceinline__ Dual operator/(Dual a, double b) { return {a.v / b, a.e / b}; }
__device__ __forceinline__ Dual operator/(double a, Dual b) { const double q = a / b.v; return {q, (-(q * b.e)) / b.v}; }
__device__ __forceinline__ double value(double a) { return a; }
__device__ __forceinline__ double value(Dual a) { return a.v; }
__device__ __forceinline__ double readlane_t(double x, int l) { return readlane_d(x, l); }
__device__ __forceinline__ Dual readlane_t(Dual x, int l) { return {readlane_d(x.v, l), readlane_d(x.e, l)}; }

__device__ __forceinline__ double pmax(double a, double b) { return a > b ? a : b; }

// ---------------------------------------------------------------- Philox4x32-10
template <int ROUNDS>
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t (&w)[4]) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;   // one v_mad_u64_u32 each
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&w)[4]) { philox4x32<10>(c0, c1, c2, c3, k0, k1, w); }

// the top 52 bits of a 64-bit half as the mantissa of a double in [1, 2)
__device__ __forceinline__ double mant_12(uint32_t lo, uint32_t hi) {
    return __hiloint2double((int)(0x3FF00000u | (hi >> 12)), (int)((lo >> 12) | (hi << 20)));
}
// u = (k + 0.5) 2^-52 in (0, 1), k = the top 52 bits.  v1 computes it by an integer conversion, v2 as m - (1 - 2^-53) from the
// mantissa form: the same double (both exact), three instructions instead of seven.
template <int RV>
__device__ __forceinline__ double bits_to_u_v(uint32_t lo, uint32_t hi) {
    if constexpr (RV == 1) {
        uint64_t bits = ((uint64_t)hi << 32) | lo;
        return ((double)(bits >> 12) + 0.5) * 0x1p-52;
    } else {
        return mant_12(lo, hi) - 0x1.fffffffffffffp-1;
    }
}
__device__ __forceinline__ double bits_to_u(uint32_t lo, uint32_t hi) { return bits_to_u_v<kRngVersion>(lo, hi); }

// ---- v1: log(x), x normal and positive: fdlibm e_log.c algorithm with +,-,*,/ only.
__device__ __forceinline__ double det_log_v1(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int32_t hx = __double2hiint(x);
    const int32_t lx = __double2loint(x);
    int32_t k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = __hiloint2double(hx | (i ^ 0x3ff00000), lx);
    k += (i >> 20);
    const double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) return (k == 0) ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        return (k == 0) ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f), z = s * s, w = z * z;
    i = hx - 0x6147a;
    const int32_t j = 0x6b851 - hx;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    const double R = t2 + t1;
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        return (k == 0) ? f - (hfsq - s * (hfsq + R))
                        : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return (k == 0) ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// exp(x): fdlibm e_exp.c algorithm with +,-,*,/ and integer operations on the exponent field only (same statements as the checker).
__device__ __forceinline__ double det_exp_v1(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return __builtin_huge_val();
    if (x < -7.45133219101941108420e+02) return 0.0;
    const uint32_t hx = (uint32_t)__double2hiint(x) & 0x7fffffffu;
    const int xsb = (int)((uint32_t)__double2hiint(x) >> 31);
    double hi = 0.0, lo = 0.0;
    int32_t k = 0;
    if (hx > 0x3fd62e42u) {
        if (hx < 0x3FF0A2B2u) {
            hi = xsb ? x + ln2HI : x - ln2HI;
            lo = xsb ? -ln2LO : ln2LO;
            k = 1 - xsb - xsb;
        } else {
            k = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
            const double t = (double)k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000u) {
        return 1.0 + x;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    long long iy = __double_as_longlong(y);
    if (k >= -1021) return __longlong_as_double(iy + ((long long)k << 52));
    return __longlong_as_double(iy + ((long long)(k + 1000) << 52)) * 9.33263618503218878990e-302;
}

__device__ __forceinline__ double ksin(double a) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = a * a, v = z * a;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return a + v * (S1 + z * r);
}
__device__ __forceinline__ double kcos(double a) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = a * a;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return (1.0 - 0.5 * z) + z * r;
}
__device__ __forceinline__ void quadrant(int q, double sa, double ca, double &s, double &c) {
    q &= 3;
    s = (q == 0) ? sa : (q == 1) ? ca : (q == 2) ? -sa : -ca;
    c = (q == 0) ? ca : (q == 1) ? -sa : (q == 2) ? -ca : sa;
}
__device__ __forceinline__ void det_sincos_2pi_v1(double u, double &s, double &c) {
    const int q = (int)(4.0 * u + 0.5);
    const double f = u - 0.25 * (double)q;
    const double a = f * 6.283185307179586476925;
    quadrant(q, ksin(a), kcos(a), s, c);
}
__device__ __forceinline__ void det_sincos_v1(double x, double &s, double &c) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_1t = 6.07710050650619224932e-11;
    const double fn = floor(x * invpio2 + 0.5);
    const double a = (x - fn * pio2_1) - fn * pio2_1t;
    quadrant((int)((long long)fn & 3), ksin(a), kcos(a), s, c);
}


// The coefficients of the v2 polynomials.  Default: literals -- the compiler emits v_fmac_f64 with the coefficient moved into the accumulator
// by two v_mov_b32 (37 of a Box-Muller pair's ~150 vector instructions).  PDMP_SCONST (A/B, scripts/probes/bm_probe.hip): the coefficient
// in a scalar register pair as the third operand of a hand-placed v_fma_f64 -- 40 % fewer vector instructions and SLOWER: 2.58 against
// 2.08 ms per 4 000 pairs per lane at two waves per SIMD (the scalar moves and the wait state in front of each fma serialise with the
// dependent chain), config 5 87.6 against 86.4 ms.  Measured, kept for the record, not the default.
#ifdef PDMP_SCONST
template <unsigned long long BITS>
__device__ __forceinline__ double sconst_bits() {
    unsigned lo, hi;
    asm("s_mov_b32 %0, %1" : "=s"(lo) : "i"((int)(unsigned)(BITS & 0xffffffffull)));
    asm("s_mov_b32 %0, %1" : "=s"(hi) : "i"((int)(unsigned)(BITS >> 32)));
    return __hiloint2double((int)hi, (int)lo);
}
#define PDMP_SC(x) (sconst_bits<__builtin_bit_cast(unsigned long long, (double)(x))>())
__device__ __forceinline__ double fma_k(double a, double b, double k_scalar) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k_scalar));
    return r;
}
#define PDMP_FMA_K(a, b, K) fma_k((a), (b), PDMP_SC(K))
#else
#define PDMP_SC(x) (x)
#define PDMP_FMA_K(a, b, K) fma((a), (b), (K))
#endif
// a / b and sqrt(x) for operands KNOWN to be finite, normal and far from the ends of the exponent range (the v2 functions below call them with
// b in [1.7, 2.5] resp. x in [2e-16, 75]): the compiler's own expansions of the IEEE operations (v_rcp_f64 / v_rsq_f64 + Newton / Goldschmidt steps on
// fma) WITHOUT their operand scaling (v_div_scale, v_div_fmas, v_div_fixup; the 2^256 pre-scale and the class test of sqrt), which is the identity
// on such operands -- the same correctly rounded results as `/` and __dsqrt_rn (= the CPU checker's `/` and sqrt()), eight resp. ten
// instructions instead of eleven resp. twenty.
__device__ __forceinline__ double div_inrange(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(r, fma(-b, r, 1.0), r);
    r = fma(r, fma(-b, r, 1.0), r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double sqrt_inrange(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double s0 = x * y, h0 = 0.5 * y;
    const double r0 = fma(-h0, s0, 0.5);
    const double s1 = fma(s0, r0, s0), h1 = fma(h0, r0, h0);
    const double s2 = fma(fma(-s1, s1, x), h1, s1);
    return fma(fma(-s2, s2, x), h1, s2);
}
// ---- v2 (mirrored statement by statement by the CPU checker: log_v2 .. box_muller there; fma = v_fma_f64 = C fma())
// A real function unless PDMP_INLINE_MATH is set (the lane kernels set it): inlined into the register-form group kernels -- built to the last
// register of their budget -- it costs 2 .. 14 VGPRs more than the branchy v1 code did (measured: 19 translation units spilled; with the
// call none does: the nine coefficients then live in scalar registers only while the function runs instead of across the event loop).
#ifdef PDMP_INLINE_MATH
#define PDMP_MATH_FN __device__ __forceinline__
#else
#define PDMP_MATH_FN __device__ __attribute__((noinline))
#endif
__device__ __forceinline__ double det_log_v2_inl(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(x);                 // [0.5, 1): v_frexp_mant_f64 == frexp()
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool c = m < 0.70710678118654752440;
    m = c ? m + m : m;
    e -= c ? 1 : 0;
    const double f = m - 1.0, dk = (double)e;
    const double s = div_inrange(f, 2.0 + f), z = s * s;
    const double P = PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_SC(Lg7), Lg6), Lg5), Lg4), Lg3), Lg2), Lg1);
    const double R = z * P;
    const double hfsq = (0.5 * f) * f;
    const double t = fma(s, hfsq + R, dk * PDMP_SC(ln2_lo));
    return fma(dk, PDMP_SC(ln2_hi), f - (hfsq - t));
}
PDMP_MATH_FN double det_log_v2(double x) { return det_log_v2_inl(x); }   // (a LEAF function: a call that calls again would need a stack)
PDMP_MATH_FN double det_exp_v2(double x) {
    constexpr double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return __builtin_huge_val();
    if (x < -7.45133219101941108420e+02) return 0.0;
    const double kf = __builtin_rint(x * invln2);
    const double hi = fma(-kf, ln2HI, x), lo = kf * ln2LO;
    const double r = hi - lo, t = r * r;
    const double P = PDMP_FMA_K(t, PDMP_FMA_K(t, PDMP_FMA_K(t, PDMP_FMA_K(t, PDMP_SC(P5), P4), P3), P2), P1);
    const double c = fma(-t, P, r);
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return __builtin_amdgcn_ldexp(y, (int)kf);
}
__device__ __forceinline__ double ksin_v2(double a) {
    constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = a * a;
    const double P = PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_SC(S6), S5), S4), S3), S2), S1);
    return fma(z * a, P, a);
}
__device__ __forceinline__ double kcos_v2(double a) {
    constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = a * a;
    const double P = PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_FMA_K(z, PDMP_SC(C6), C5), C4), C3), C2), C1);
    return fma(z * z, P, fma(z, -0.5, 1.0));
}
// quadrant q mod 4 as one swap and two sign flips (integer operations on the sign words)
__device__ __forceinline__ void quadrant_v2(int q, double sa, double ca, double &s, double &c) {
    const bool sw = (q & 1) != 0;
    const double ss = sw ? ca : sa, cc = sw ? sa : ca;
    s = __hiloint2double(__double2hiint(ss) ^ (int)(((uint32_t)q & 2u) << 30), __double2loint(ss));
    c = __hiloint2double(__double2hiint(cc) ^ (int)(((uint32_t)(q + 1) & 2u) << 30), __double2loint(cc));
}
struct Pair64 { double a, b; };                                  // two doubles returned in registers
PDMP_MATH_FN Pair64 sincos_core_v2(double x) {
    constexpr double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_1t = 6.07710050650619224932e-11;
    const double fn = __builtin_rint(x * invpio2);
    const double a = fma(-fn, PDMP_SC(pio2_1t), fma(-fn, PDMP_SC(pio2_1), x));
    Pair64 r;
    quadrant_v2((int)fn, ksin_v2(a), kcos_v2(a), r.a, r.b);
    return r;
}
__device__ __forceinline__ void det_sincos_v2(double x, double &s, double &c) { const Pair64 r = sincos_core_v2(x); s = r.a; c = r.b; }
PDMP_MATH_FN Pair64 sincos_2pi_core_v2(double u) {
    const double fn = __builtin_rint(4.0 * u);
    const double a = fma(-0.25, fn, u) * 6.283185307179586476925;
    Pair64 r;
    quadrant_v2((int)fn, ksin_v2(a), kcos_v2(a), r.a, r.b);
    return r;
}
__device__ __forceinline__ void det_sincos_2pi_v2(double u, double &s, double &c) { const Pair64 r = sincos_2pi_core_v2(u); s = r.a; c = r.b; }
template <int RV> __device__ __forceinline__ double det_log_v(double x) { if constexpr (RV == 1) return det_log_v1(x); else return det_log_v2(x); }
template <int RV> __device__ __forceinline__ double det_exp_v(double x) { if constexpr (RV == 1) return det_exp_v1(x); else return det_exp_v2(x); }
template <int RV> __device__ __forceinline__ void det_sincos_v(double x, double &s, double &c) { if constexpr (RV == 1) det_sincos_v1(x, s, c); else det_sincos_v2(x, s, c); }
template <int RV> __device__ __forceinline__ void det_sincos_2pi_v(double u, double &s, double &c) { if constexpr (RV == 1) det_sincos_2pi_v1(u, s, c); else det_sincos_2pi_v2(u, s, c); }
// the functions of this translation unit's contract
__device__ __forceinline__ double det_log(double x) { return det_log_v<kRngVersion>(x); }
__device__ __forceinline__ double det_exp(double x) { return det_exp_v<kRngVersion>(x); }
__device__ __forceinline__ void det_sincos(double x, double &s, double &c) { det_sincos_v<kRngVersion>(x, s, c); }
__device__ __forceinline__ void det_sincos_2pi(double u, double &s, double &c) { det_sincos_2pi_v<kRngVersion>(u, s, c); }

// One Box-Muller pair from the four words of a block.  v1: (z0, z1) = sqrt(-2 log u0) (cos, sin)(2 pi u1).  v2: the same radius; the
// angle is uniform on the circle as a in [-pi/4, pi/4) (the top 52 bits of the second half as a mantissa m in [1, 2): a = m pi/2 - 3 pi/4),
// one SWAP bit (sin <-> cos: the angle pi/2 - a covers (pi/4, 3 pi/4]) and one SIGN bit (the angle + pi) -- bits 0 and 1 of the same half:
// no quadrant reduction and no integer conversion, the same distribution.
PDMP_MATH_FN Pair64 box_muller_core_v2(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    const double u0 = bits_to_u_v<2>(w0, w1);
    double rad = sqrt_inrange(-2.0 * det_log_v2_inl(u0));
    const double a = fma(mant_12(w2, w3), PDMP_SC(1.57079632679489661923), PDMP_SC(-2.35619449019234492885));
    const double sa = ksin_v2(a), ca = kcos_v2(a);
    const bool sw = (w2 & 1u) != 0u;
    rad = __hiloint2double(__double2hiint(rad) ^ (int)((w2 & 2u) << 30), __double2loint(rad));
    Pair64 r;
    r.a = rad * (sw ? sa : ca);
    r.b = rad * (sw ? ca : sa);
    return r;
}
template <int RV>
__device__ __forceinline__ void box_muller_v(const uint32_t (&w)[4], double &z0, double &z1) {
    if constexpr (RV == 1) {
        const double u0 = bits_to_u_v<1>(w[0], w[1]), u1 = bits_to_u_v<1>(w[2], w[3]);
        const double rad = __dsqrt_rn(-2.0 * det_log_v1(u0));
        double sn, cs;
        det_sincos_2pi_v1(u1, sn, cs);
        z0 = rad * cs;
        z1 = rad * sn;
    } else {
        const Pair64 r = box_muller_core_v2(w[0], w[1], w[2], w[3]);
        z0 = r.a;
        z1 = r.b;
    }
}

// ---------------------------------------------------------------- per-chain random stream
// One stream per chain (= per wave); all state is wave-uniform so the Philox rounds run on the
// scalar unit.  Variate n uses block n>>1, half n&1 (spec: DESIGN.md "Random stream").
template <int RV>
struct ChainRngT {
    static constexpr int kRounds = RV == 1 ? 10 : 7;
    uint32_t k0, k1, c2, c3;   // key = seed, counter high words = chain id
    uint64_t n;                // variates consumed
    uint32_t w2, w3;           // second half of the current block

    __device__ __forceinline__ void init(uint64_t seed, uint64_t chain) {
        k0 = __builtin_amdgcn_readfirstlane((uint32_t)seed);
        k1 = __builtin_amdgcn_readfirstlane((uint32_t)(seed >> 32));
        c2 = __builtin_amdgcn_readfirstlane((uint32_t)chain);
        c3 = __builtin_amdgcn_readfirstlane((uint32_t)(chain >> 32));
        n = 0; w2 = 0; w3 = 0;
    }
    __device__ __forceinline__ void init_lane(uint64_t seed, uint64_t chain) {   // per-lane streams (group kernel)
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
        c2 = (uint32_t)chain; c3 = (uint32_t)(chain >> 32);
        n = 0; w2 = 0; w3 = 0;
    }
    // resume at variate counter n0: an odd counter needs the second half of its block again
    __device__ __forceinline__ void restore(uint64_t n0) {
        n = n0;
        if (n0 & 1) {
            uint32_t w[4];
            const uint64_t blk = n0 >> 1;
            philox4x32<kRounds>((uint32_t)blk, (uint32_t)(blk >> 32), c2, c3, k0, k1, w);
            w2 = w[2]; w3 = w[3];
        }
    }
    __device__ __forceinline__ double uniform() {
        uint32_t lo, hi;
        if ((n & 1) == 0) {
            uint32_t w[4];
            const uint64_t blk = n >> 1;
            philox4x32<kRounds>((uint32_t)blk, (uint32_t)(blk >> 32), c2, c3, k0, k1, w);
            lo = w[0]; hi = w[1]; w2 = w[2]; w3 = w[3];
        } else {
            lo = w2; hi = w3;
        }
        ++n;
        return bits_to_u_v<RV>(lo, hi);
    }
    __device__ __forceinline__ double exponential() { return -det_log_v<RV>(uniform()); }
    // the four words of block `blk` of this stream (variates 2 blk and 2 blk + 1), stateless
    __device__ __forceinline__ void block_at(uint64_t blk, uint32_t (&w)[4]) const {
        philox4x32<kRounds>((uint32_t)blk, (uint32_t)(blk >> 32), c2, c3, k0, k1, w);
    }
    // standard normal for coordinate i of a d-vector; call normals_done(d) once afterwards
    __device__ __forceinline__ double normal_coord(int i) const {
        const uint64_t blk = ((n + 1) >> 1) + (uint64_t)(i >> 1);
        uint32_t w[4];
        philox4x32<kRounds>((uint32_t)blk, (uint32_t)(blk >> 32), c2, c3, k0, k1, w);
        double z0, z1;
        box_muller_v<RV>(w, z0, z1);
        return (i & 1) ? z1 : z0;
    }
    // both normals of pair P (coordinates 2P and 2P+1) from one Philox block
    __device__ __forceinline__ void normal_pair(int P, double &z0, double &z1) const {
        const uint64_t blk = ((n + 1) >> 1) + (uint64_t)P;
        uint32_t w[4];
        philox4x32<kRounds>((uint32_t)blk, (uint32_t)(blk >> 32), c2, c3, k0, k1, w);
        box_muller_v<RV>(w, z0, z1);
    }
    __device__ __forceinline__ void normals_done(int d) {
        n = ((n + 1) & ~1ull) + 2 * (uint64_t)((d + 1) / 2);
    }
};
using ChainRng = ChainRngT<kRngVersion>;

PDMP_NS_END
