/*
 * pdmp_oracle.c -- CPU restatement (plain C, scalar, one chain at a time) of the PDMPFlux.jl
 * sample_skeleton()/sample_from_skeleton() hot path.
 *
 * TEST INFRASTRUCTURE: checker for the HIP engine and timed CPU baseline ("port").  Never on the
 * product path.   PARITY UNPINNED (see pdmp_oracle.h): no Julia toolchain, no golden vectors in
 * the reference; pinned by invariants, statistics, hand-derived answers and Philox/libm KATs.
 *
 * Build: see oracle/Makefile   (gcc -O2 -ffp-contract=off -fopenmp; no fast-math: the rounding of
 * every expression below is part of the specification the HIP engine is compared against).
 *
 * Citations are file:line under /root/reference/.
 */
#include "pdmp_oracle.h"
#include "julia_rng.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[512];
const char *orc_last_error(void) { return g_err; }
#define FAIL(code, ...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return (code); } while (0)

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Random stream (build-defined; replaces Julia's MersenneTwister/randexp/randn,
 * AbstractPDMP.jl:100, SamplingLoopInplace.jl:70,136,154,190 -- "third-party arithmetic").
 *   Philox4x32-10 (Salmon et al., SC'11; Random123 v1.14 known-answer vectors in tests/).
 *   key = (seed_lo, seed_hi); counter = (blk_lo, blk_hi, chain_lo, chain_hi).
 *   Variate n of a chain uses block n>>1, 64-bit half n&1: bits = w[2h] | w[2h+1]<<32;
 *   u = ((bits >> 12) + 0.5) * 2^-52  in (0,1);  Exp(1) = -log(u) with the log below.
 *   Normal pairs (Box-Muller) consume one whole block: (z0,z1) = sqrt(-2 log u0) (cos,sin)(2 pi u1).
 * ------------------------------------------------------------------------------------------ */
/* Arithmetic contract version of the calling thread (orc_config_t.rng_version; set per chain by the entry points below):
 *   1  Philox4x32-10, fdlibm-style log / exp / sin / cos written with + - * / only (rounds 1-4: every frozen fixture of that time);
 *   2  (default since round 5) Philox4x32-7 -- the smallest round count Salmon et al. report as passing BigCrush --, the same
 *      polynomials evaluated by Horner's rule on fma() (exact IEEE fusedMultiplyAdd on both sides: C fma() == v_fma_f64), argument
 *      reduction by frexp / rint, and a Box-Muller pair whose angle is drawn as (octant angle, swap bit, sign bit) instead of
 *      2 pi u with a quadrant reduction.  Same stream layout: variate n = block n>>1, half n&1; uniforms are the same function of
 *      a block's words.  DESIGN.md "Arithmetic contract v2". */
static __thread int g_ver = 2;
void orc_set_arith_version(int v) { g_ver = (v == 1) ? 1 : 2; }
int orc_get_arith_version(void) { return g_ver; }

void orc_philox4x32_r(int rounds, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32_r(10, ctr, key, out); }

/* Tape source: when te/tu/tn are set the chain replays pre-recorded Exp(1), U(0,1) and N(0,1)
 * variates (one tape per kind, each in consumption order -- enough to replay a run of ANY generator, e.g. a dump of
 * Julia's MersenneTwister randexp/rand/randn, because the control flow is a function of the variates) instead of
 * drawing from Philox.  An exhausted tape yields 1.0 / 0.5 / 0.0 and sets *overrun. */
typedef struct { uint64_t seed; uint64_t chain; uint64_t n;
                 const double *te, *tu, *tn; int64_t ne, nu, nn, ie, iu, in; int *overrun;
                 /* Julia source (julia_rng.c): the chain's MersenneTwister and, optionally, the log of what it hands out (le / lu / ln: this
                  * chain's output tapes of capacity ce / cu / cn; je / ju / jn: variates logged so far) */
                 jl_mt_t *jl; double *le, *lu, *ln; int64_t ce, cu, cn, je, ju, jn; int jl_scalar; } rng_t;
static inline double jl_log_put(double v, double *tape, int64_t cap, int64_t *n, int *overrun) {
    if (tape) { if (*n < cap) tape[*n] = v; else if (overrun) *overrun = 1; }
    ++*n;
    return v;
}

static inline double bits_to_u(uint64_t bits) {
    return ((double)(bits >> 12) + 0.5) * 0x1p-52;
}
static void rng_block(const rng_t *r, uint64_t blk, uint32_t w[4]) {
    uint32_t ctr[4] = { (uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)r->chain, (uint32_t)(r->chain >> 32) };
    uint32_t key[2] = { (uint32_t)r->seed, (uint32_t)(r->seed >> 32) };
    orc_philox4x32_r(g_ver == 1 ? 10 : 7, ctr, key, w);
}
static double rng_uniform(rng_t *r) {
    if (r->jl) return jl_log_put(jl_rand(r->jl), r->lu, r->cu, &r->ju, r->overrun);
    if (r->tu) { if (r->iu < r->nu) return r->tu[r->iu++]; if (r->overrun) *r->overrun = 1; return 0.5; }
    uint32_t w[4];
    rng_block(r, r->n >> 1, w);
    int h = (int)(r->n & 1);
    r->n++;
    return bits_to_u((uint64_t)w[2 * h] | ((uint64_t)w[2 * h + 1] << 32));
}

/* log(x) for normal x > 0: the fdlibm e_log.c algorithm (Sun Microsystems, freely distributable),
 * written with +,-,*,/ only so that CPU and GPU agree bit for bit. */
static double log_v1(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    uint64_t ix; memcpy(&ix, &x, 8);
    int32_t hx = (int32_t)(ix >> 32);
    int32_t k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    ix = (ix & 0xffffffffull) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
    memcpy(&x, &ix, 8);
    k += (i >> 20);
    double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {            /* |f| < 2^-20 */
        if (f == 0.0) return (k == 0) ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        return (k == 0) ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f), z = s * s, w = z * z;
    i = hx - 0x6147a;
    int32_t j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        return (k == 0) ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return (k == 0) ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* exp(x): the fdlibm e_exp.c algorithm (Sun Microsystems, freely distributable) with +,-,*,/ and integer operations on the
 * exponent field only, so that CPU and GPU agree bit for bit.  Overflow -> +Inf, underflow below -745.13 -> 0. */
static double exp_v1(double x) {
    static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
        invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    uint64_t ix; memcpy(&ix, &x, 8);
    const uint32_t hx = (uint32_t)(ix >> 32) & 0x7fffffffu;
    const int xsb = (int)(ix >> 63);
    double hi = 0.0, lo = 0.0;
    int32_t k = 0;
    if (hx > 0x3fd62e42u) {                         /* |x| > 0.5 ln2 */
        if (hx < 0x3FF0A2B2u) {                     /* and |x| < 1.5 ln2 */
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
    } else if (hx < 0x3e300000u) {                  /* |x| < 2^-28 */
        return 1.0 + x;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    uint64_t iy; memcpy(&iy, &y, 8);
    if (k >= -1021) {
        iy += (uint64_t)(int64_t)k << 52;           /* add k to y's exponent */
        memcpy(&y, &iy, 8);
        return y;
    }
    iy += (uint64_t)(int64_t)(k + 1000) << 52;
    memcpy(&y, &iy, 8);
    return y * 9.33263618503218878990e-302;         /* 2^-1000 */
}

/* fdlibm k_sin.c / k_cos.c polynomial kernels on |a| <= pi/4 (no tail term). */
static inline double ksin(double a) {
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = a * a, v = z * a;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return a + v * (S1 + z * r);
}
static inline double kcos(double a) {
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = a * a;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return (1.0 - 0.5 * z) + z * r;
}
static inline void quadrant(int q, double sa, double ca, double *s, double *c) {
    switch (q & 3) {
    case 0: *s = sa;  *c = ca;  break;
    case 1: *s = ca;  *c = -sa; break;
    case 2: *s = -sa; *c = -ca; break;
    default: *s = -ca; *c = sa; break;
    }
}
/* sin/cos(2 pi u), u in (0,1): exact quadrant reduction on u, then the kernels. */
static void sincos_2pi_v1(double u, double *s, double *c) {
    int q = (int)(4.0 * u + 0.5);
    double f = u - 0.25 * (double)q;               /* exact, |f| <= 1/8 */
    double a = f * 6.283185307179586476925;
    quadrant(q, ksin(a), kcos(a), s, c);
}
/* sin/cos(x) for 0 <= |x| < ~1e5: two-term Cody-Waite reduction by pi/2 (fdlibm constants). */
static void sincos_v1(double x, double *s, double *c) {
    static const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
        pio2_1t = 6.07710050650619224932e-11;
    double fn = floor(x * invpio2 + 0.5);
    double a = (x - fn * pio2_1) - fn * pio2_1t;
    quadrant((int)((int64_t)fn & 3), ksin(a), kcos(a), s, c);
}


/* ---- arithmetic contract v2: the same series, Horner on fma(); mirrored statement by statement in csrc/wave_math.hpp ---- */
/* log(x), x normal and positive.  x = 2^e m, m in [sqrt(1/2), sqrt(2)); f = m - 1; s = f / (2 + f);
 * log(1 + f) = f - (f^2/2 - s (f^2/2 + R(s^2)))  (the fdlibm series, its coefficients); + e ln2 in two pieces. */
static double log_v2(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    int e;
    double m = frexp(x, &e);                                  /* [0.5, 1) */
    const int c = m < 0.70710678118654752440;
    m = c ? m + m : m;                                        /* [sqrt(1/2), sqrt(2)) */
    e -= c;
    const double f = m - 1.0, dk = (double)e;
    const double s = f / (2.0 + f), z = s * s;
    const double P = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, Lg7, Lg6), Lg5), Lg4), Lg3), Lg2), Lg1);
    const double R = z * P;
    const double hfsq = (0.5 * f) * f;
    const double t = fma(s, hfsq + R, dk * ln2_lo);
    return fma(dk, ln2_hi, f - (hfsq - t));
}
/* exp(x): k = rint(x / ln2), r = (x - k ln2_hi) - k ln2_lo, exp(r) = 1 + (r c / (2 - c) ... ) with c = r - r^2 P(r^2) (fdlibm's series), 2^k by ldexp */
static double exp_v2(double x) {
    static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
        invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    const double kf = rint(x * invln2);
    const double hi = fma(-kf, ln2HI, x), lo = kf * ln2LO;
    const double r = hi - lo, t = r * r;
    const double P = fma(t, fma(t, fma(t, fma(t, P5, P4), P3), P2), P1);
    const double c = fma(-t, P, r);
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return ldexp(y, (int)kf);
}
static inline double ksin_v2(double a) {
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = a * a;
    const double P = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
    return fma(z * a, P, a);
}
static inline double kcos_v2(double a) {
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = a * a;
    const double P = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    return fma(z * z, P, fma(z, -0.5, 1.0));
}
/* sin / cos of x: n = rint(x 2/pi), a = (x - n pio2_1) - n pio2_1t by two fma, kernels on |a| <= pi/4, quadrant n mod 4 as one swap and two signs */
static void sincos_v2(double x, double *s, double *c) {
    static const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
        pio2_1t = 6.07710050650619224932e-11;
    const double fn = rint(x * invpio2);
    const double a = fma(-fn, pio2_1t, fma(-fn, pio2_1, x));
    const int q = (int)fn;
    const double sa = ksin_v2(a), ca = kcos_v2(a);
    double ss = (q & 1) ? ca : sa, cc = (q & 1) ? sa : ca;
    if (q & 2) ss = -ss;
    if ((q + 1) & 2) cc = -cc;
    *s = ss; *c = cc;
}
/* sin / cos (2 pi u), u in (0, 1): n = rint(4 u), a = 2 pi (u - n/4) (exact difference), as above */
static void sincos_2pi_v2(double u, double *s, double *c) {
    const double fn = rint(4.0 * u);
    const double a = fma(-0.25, fn, u) * 6.283185307179586476925;
    const int q = (int)fn;
    const double sa = ksin_v2(a), ca = kcos_v2(a);
    double ss = (q & 1) ? ca : sa, cc = (q & 1) ? sa : ca;
    if (q & 2) ss = -ss;
    if ((q + 1) & 2) cc = -cc;
    *s = ss; *c = cc;
}
double orc_log(double x) { return g_ver == 1 ? log_v1(x) : log_v2(x); }
double orc_exp(double x) { return g_ver == 1 ? exp_v1(x) : exp_v2(x); }
void orc_sincos(double x, double *s, double *c) { if (g_ver == 1) sincos_v1(x, s, c); else sincos_v2(x, s, c); }
void orc_sincos_2pi(double u, double *s, double *c) { if (g_ver == 1) sincos_2pi_v1(u, s, c); else sincos_2pi_v2(u, s, c); }

/* One Box-Muller pair from the four words of a block.  v1: (z0, z1) = sqrt(-2 log u0) (cos, sin)(2 pi u1).
 * v2: the radius as before; the angle is uniform on the circle as  a in [-pi/4, pi/4)  (52 bits of the second half as the mantissa
 * of m in [1, 2): a = m pi/2 - 3 pi/4),  one SWAP bit (sin <-> cos: the angle pi/2 - a covers (pi/4, 3 pi/4]) and one SIGN bit
 * (the angle + pi), bits 0 and 1 of the same 64-bit half -- no quadrant reduction, the same distribution. */
static inline double mant_12(uint64_t bits) {                  /* the top 52 bits as the mantissa of a double in [1, 2) */
    const uint64_t i = 0x3FF0000000000000ull | (bits >> 12);
    double m; memcpy(&m, &i, 8);
    return m;
}
static void box_muller(const uint32_t w[4], double *z0, double *z1) {
    const uint64_t b0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), b1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    if (g_ver == 1) {
        const double u0 = bits_to_u(b0), u1 = bits_to_u(b1);
        double rad = sqrt(-2.0 * log_v1(u0)), sn, cs;
        sincos_2pi_v1(u1, &sn, &cs);
        *z0 = rad * cs; *z1 = rad * sn;
        return;
    }
    const double u0 = mant_12(b0) - 0x1.fffffffffffffp-1;      /* (k + 0.5) 2^-52, exactly bits_to_u(b0) */
    double rad = sqrt(-2.0 * log_v2(u0));
    const double a = fma(mant_12(b1), 1.57079632679489661923, -2.35619449019234492885);
    double sa = ksin_v2(a), ca = kcos_v2(a);
    if (b1 & 1ull) { const double tmp = sa; sa = ca; ca = tmp; }
    if (b1 & 2ull) rad = -rad;
    *z0 = rad * ca; *z1 = rad * sa;
}

static double rng_exp(rng_t *r) {
    if (r->jl) return jl_log_put(jl_randexp(r->jl), r->le, r->ce, &r->je, r->overrun);
    if (r->te) { if (r->ie < r->ne) return r->te[r->ie++]; if (r->overrun) *r->overrun = 1; return 1.0; }
    return -orc_log(rng_uniform(r));
}

/* d standard normals; aligns the variate counter to a block boundary first. */
static void rng_normals(rng_t *r, int d, double *z) {
    if (r->jl) {            /* randn!(rng, v) on the chain's MersenneTwister: the bulk method from 13 elements on (julia_rng.c: jl_randn_fill), or -- jl_scalar,
                             * what an rng WRAPPED around it sees, julia/dump_tape.jl's TapeRNG -- d scalar draws in index order */
        jl_randn_fill(r->jl, d, z, r->jl_scalar);
        for (int i = 0; i < d; ++i) (void)jl_log_put(z[i], r->ln, r->cn, &r->jn, r->overrun);
        return;
    }
    if (r->tn) {
        for (int i = 0; i < d; ++i) { if (r->in < r->nn) z[i] = r->tn[r->in++]; else { z[i] = 0.0; if (r->overrun) *r->overrun = 1; } }
        return;
    }
    if (r->n & 1) r->n++;
    uint64_t blk0 = r->n >> 1;
    for (int p = 0; 2 * p < d; ++p) {
        uint32_t w[4];
        rng_block(r, blk0 + (uint64_t)p, w);
        double za, zb;
        box_muller(w, &za, &zb);
        z[2 * p] = za;
        if (2 * p + 1 < d) z[2 * p + 1] = zb;
    }
    r->n += 2 * (uint64_t)((d + 1) / 2);
}

/* First n draws of rand (0) / randexp (1) / randn (2) from the restated MersenneTwister(seed): known-answer tests (tests/test_julia_rng.py). */
void orc_julia_draws(int64_t seed, int seeding, int kind, int64_t n, double *out) {
    jl_mt_t m;
    jl_mt_seed(&m, seed, seeding);
    for (int64_t i = 0; i < n; ++i) out[i] = (kind == 0) ? jl_rand(&m) : (kind == 1) ? jl_randexp(&m) : jl_randn(&m);
}

/* randn!(MersenneTwister(seed), Vector{Float64}(undef, n)) as restated in julia_rng.c (bulk method from 13 elements on; scalar != 0: n scalar draws) */
void orc_julia_randn_fill(int64_t seed, int seeding, int n, int scalar, double *out) {
    jl_mt_t m;
    jl_mt_seed(&m, seed, seeding);
    jl_randn_fill(&m, n, out, scalar);
}

void orc_rng_dump(uint64_t seed, int64_t chain, int kind, int64_t n_draw, double *out) {
    rng_t r = { .seed = seed, .chain = (uint64_t)chain };
    if (kind == 2) { rng_normals(&r, (int)n_draw, out); return; }
    for (int64_t i = 0; i < n_draw; ++i) out[i] = (kind == 0) ? rng_uniform(&r) : rng_exp(&r);
}

/* Synthetic initial states of an ensemble ("xinit[:,c] ~ N(0,I), vinit[i,c] = +-1 from Philox key (2024, c)
 * stream 1; chain RNG = stream 0"): stream 1 of chain c is the chain's Philox stream with the top bit of the 64-bit chain word set.
 * Draw order: d normals for x (Box-Muller pairs, as rng_normals), then for v: ORC_V_RADEMACHER d uniforms (u < 0.5 -> -1, else +1),
 * ORC_V_GAUSSIAN d normals, ORC_V_SPHERE d normals divided by sqrt of their left-to-right sum of squares.
 * x_i = x_mean + x_scale z_i.  Mirrors pdmp_init_states_device (csrc/engine_api.hip) bit for bit. */
void orc_init_states(uint64_t seed, int64_t chain0, int64_t C, int d, double x_mean, double x_scale, int v_kind, double *x, double *v) {
    for (int64_t c = 0; c < C; ++c) {
        rng_t r = { .seed = seed, .chain = ((uint64_t)(chain0 + c)) | (1ull << 63) };
        double *xc = x + c * d, *vc = v + c * d;
        rng_normals(&r, d, xc);
        for (int i = 0; i < d; ++i) xc[i] = x_mean + x_scale * xc[i];
        if (v_kind == ORC_V_RADEMACHER) {
            for (int i = 0; i < d; ++i) vc[i] = (rng_uniform(&r) < 0.5) ? -1.0 : 1.0;
        } else {
            rng_normals(&r, d, vc);
            if (v_kind == ORC_V_SPHERE) {
                double ss = 0.0;
                for (int i = 0; i < d; ++i) ss += vc[i] * vc[i];
                const double nrm = sqrt(ss);
                for (int i = 0; i < d; ++i) vc[i] = vc[i] / nrm;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Summation orders (orc_config_t.arith)
 * ------------------------------------------------------------------------------------------ */
static double sum_seq(const double *a, int d) {
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += a[i];
    return s;
}
/* lane l accumulates a[l], a[l+64], ...; then xor-butterfly with offsets 1,2,4,...,32 */
static double sum_wave64(const double *a, int d) {
    double p[64], q[64];
    for (int l = 0; l < 64; ++l) {
        double s = 0.0;
        for (int i = l; i < d; i += 64) s += a[i];
        p[l] = s;
    }
    for (int off = 1; off < 64; off <<= 1) {
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof p);
    }
    return p[0];
}
/* GROUP(G): lane g of G accumulates a[g], a[g+G], ... left to right; then xor-butterfly 1,2,..,G/2.
 * GROUP(1) is the reference order, GROUP(64) the WAVE64 order. */
static double sum_group(const double *a, int d, int G) {
    double p[64], q[64];
    for (int g = 0; g < G; ++g) {
        double s = 0.0;
        for (int i = g; i < d; i += G) s += a[i];
        p[g] = s;
    }
    for (int off = 1; off < G; off <<= 1) {
        for (int g = 0; g < G; ++g) q[g] = p[g] + p[g ^ off];
        memcpy(p, q, sizeof(double) * (size_t)G);
    }
    return p[0];
}
static double vsum(const orc_config_t *cfg, const double *a, int d) {
    if (cfg->arith == ORC_ARITH_WAVE64) return sum_wave64(a, d);
    if (cfg->arith == ORC_ARITH_GROUP) return sum_group(a, d, cfg->group_lanes);
    return sum_seq(a, d);
}
/* first index i with inclusive-prefix(p)[i] > u, else d-1.
 * REFERENCE: Distributions.jl DiscreteNonParametric rand (compat 0.25, Project.toml:31; called at
 * ZigZagSamplers.jl:104): cp = p[1]; while cp <= draw && i < n: cp += p[i += 1].
 * WAVE64: chunks of 64, Kogge-Stone inclusive scan inside a chunk, running carry between chunks. */
static int categorical(const orc_config_t *cfg, const double *p, int d, double u) {
    if (cfg->arith == ORC_ARITH_REFERENCE) {
        double cp = p[0]; int i = 0;
        while (cp <= u && i < d - 1) { ++i; cp += p[i]; }
        return i;
    }
    const int W = (cfg->arith == ORC_ARITH_GROUP) ? cfg->group_lanes : 64;
    double carry = 0.0;
    for (int base = 0; base < d; base += W) {
        double y[64], z[64];
        for (int l = 0; l < W; ++l) y[l] = (base + l < d) ? p[base + l] : 0.0;
        for (int off = 1; off < W; off <<= 1) {
            for (int l = 0; l < W; ++l) z[l] = (l >= off) ? y[l] + y[l - off] : y[l];
            memcpy(y, z, sizeof(double) * (size_t)W);
        }
        for (int l = 0; l < W && base + l < d; ++l)
            if (carry + y[l] > u) return base + l;
        carry = carry + y[W - 1];
    }
    return d - 1;
}

/* GROUP order only: the engine's group kernel fuses the multiply-add of the linear flow x + v t, of the tangent
 * intersection numerator and of the intersection value (one rounding instead of two; the reference and the other
 * orders keep them separate).  fma() is exact IEEE fusedMultiplyAdd on both sides. */
#define FUSED(cfg) ((cfg)->arith == ORC_ARITH_GROUP)

/* max without NaN propagation -- both sides of the parity comparison use this form */
static inline double pmax(double a, double b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------
 * Built-in potentials U and their gradients, evaluated on dual numbers (val, eps) so that the
 * time-derivative of the rate is exact forward-mode AD (ForwardDiff.derivative, UpperBound.jl:102,210).
 * Dual product rule used everywhere: (a*b).eps = a.eps*b.val + a.val*b.eps, no fused multiply-add.
 * ------------------------------------------------------------------------------------------ */
static void grad_dual(const orc_config_t *cfg, const double *xv, const double *xe, double *gv, double *ge) {
    const int d = cfg->dim;
    switch (cfg->target_id) {
    case ORC_T_GAUSS_ISO:                       /* U = |x|^2/2  (README.md:36-38) */
        for (int i = 0; i < d; ++i) { gv[i] = xv[i]; ge[i] = xe[i]; }
        break;
    case ORC_T_GAUSS_DIAG: {                    /* U = sum prec_i (x_i-mu_i)^2 / 2 */
        const double *mu = cfg->target_params, *prec = cfg->target_params + d;
        for (int i = 0; i < d; ++i) { gv[i] = prec[i] * (xv[i] - mu[i]); ge[i] = prec[i] * xe[i]; }
        break;
    }
    case ORC_T_BANANA: {                        /* test/test_config.jl:33-36; SURVEY Q10 */
        if (d < 2) { gv[0] = xv[0]; ge[0] = xe[0]; break; }
        double sq_v = xv[0] * xv[0], sq_e = xe[0] * xv[0] + xv[0] * xe[0];
        double m_v = sq_v - 1.0, m_e = sq_e;                 /* mean_x2 = x1^2 - 1 */
        double r_v = xv[1] - m_v, r_e = xe[1] - m_e;         /* r = x2 - mean_x2 */
        double a_v = 2.0 * xv[0], a_e = 2.0 * xe[0];
        double pr_v = a_v * r_v, pr_e = a_e * r_v + a_v * r_e;
        gv[0] = xv[0] - pr_v; ge[0] = xe[0] - pr_e;          /* dU/dx1 = x1 - 2 x1 r */
        gv[1] = r_v;          ge[1] = r_e;                   /* dU/dx2 = r */
        for (int i = 2; i < d; ++i) { gv[i] = xv[i]; ge[i] = xe[i]; }
        break;
    }
    case ORC_T_GAUSS_DENSE: {                   /* U = x'Px/2, P column-major */
        const double *P = cfg->target_params;
        for (int i = 0; i < d; ++i) {
            double sv = 0.0, se = 0.0;
            for (int j = 0; j < d; ++j) { sv += P[i + (size_t)j * d] * xv[j]; se += P[i + (size_t)j * d] * xe[j]; }
            gv[i] = sv; ge[i] = se;
        }
        break;
    }
    case ORC_T_AD_BANANA: case ORC_T_AD_FUNNEL: case ORC_T_AD_RIDGED:      /* ForwardDiff.gradient(U, x), ADBackend.jl:107 */
        orc_ad_grad_dual(cfg->target_id, d, xv, xe, gv, ge);
        break;
    }
}
void orc_grad(const orc_config_t *cfg, const double *x, double *g) {
    const int d = cfg->dim;
    double *ze = calloc((size_t)d, 8), *ge = malloc((size_t)d * 8);
    grad_dual(cfg, x, ze, g, ge);
    free(ze); free(ge);
}
double orc_potential(const orc_config_t *cfg, const double *x) {
    const int d = cfg->dim; double s = 0.0;
    switch (cfg->target_id) {
    case ORC_T_AD_BANANA: case ORC_T_AD_FUNNEL: case ORC_T_AD_RIDGED: return orc_ad_potential(cfg->target_id, d, x);
    case ORC_T_GAUSS_ISO: for (int i = 0; i < d; ++i) s += x[i] * x[i]; return s / 2;
    case ORC_T_GAUSS_DIAG: { const double *mu = cfg->target_params, *pr = mu + d;
        for (int i = 0; i < d; ++i) s += pr[i] * (x[i] - mu[i]) * (x[i] - mu[i]); return s / 2; }
    case ORC_T_BANANA: { if (d < 2) return x[0] * x[0] / 2;
        double r = x[1] - (x[0] * x[0] - 1.0); s = x[0] * x[0] + r * r;
        for (int i = 2; i < d; ++i) s += x[i] * x[i]; return s / 2; }
    default: { const double *P = cfg->target_params;
        for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) s += x[i] * P[i + (size_t)j * d] * x[j];
        return s / 2; }
    }
}

/* ------------------------------------------------------------------------------------------
 * Per-chain workspace
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int d, n;
    double *x, *v;                         /* PDMPState.x / .v (Composites.jl:60-61) */
    double *xv, *xe, *vv, *ve, *gv, *ge;   /* dual scratch: x_t, v_t, grad */
    double *tmp, *tmp2;                    /* d-vectors */
    double *vals, *slps;                   /* d*n (vect) or n (scalar), column j = grid time j */
    double *fa, *fb;                       /* finite-difference scratch, d */
    double *grid, *box, *cum;              /* BoundBox (Composites.jl:15-20) */
    double *cell;                          /* d */
    double *fe_vo, *fe_e1, *fe_e2, *fe_z, *fe_g;   /* ForwardECMC velocity jump: d, d, d, d, 2d */
    double *vu, *actf;                     /* StickyZigZag: v .* is_active (PDMPState.v_active) and is_active as 0.0 / 1.0 */
    double step;
    double lr_raw;                         /* affine-tail order (BPS, GROUP): the accepted proposal's g.v, reused by the bounce */
} ws_t;

static int ws_init(ws_t *w, int d, int n) {
    memset(w, 0, sizeof *w);
    w->d = d; w->n = n;
    size_t D = (size_t)d, N = (size_t)n;
    double **ptrs[] = { &w->x, &w->v, &w->xv, &w->xe, &w->vv, &w->ve, &w->gv, &w->ge, &w->tmp, &w->tmp2,
                        &w->fa, &w->fb, &w->cell, &w->fe_vo, &w->fe_e1, &w->fe_e2, &w->fe_z, &w->vu, &w->actf };
    for (size_t i = 0; i < sizeof ptrs / sizeof *ptrs; ++i) if (!(*ptrs[i] = malloc(D * 8))) return -1;
    if (!(w->vals = malloc(D * N * 8)) || !(w->slps = malloc(D * N * 8)) || !(w->fe_g = malloc(2 * D * 8))) return -1;
    if (!(w->grid = malloc(N * 8)) || !(w->box = malloc(N * 8)) || !(w->cum = malloc(N * 8))) return -1;
    return 0;
}
static void ws_free(ws_t *w) {
    free(w->x); free(w->v); free(w->xv); free(w->xe); free(w->vv); free(w->ve); free(w->gv); free(w->ge);
    free(w->tmp); free(w->tmp2); free(w->fa); free(w->fb); free(w->cell);
    free(w->vals); free(w->slps); free(w->grid); free(w->box); free(w->cum);
    free(w->fe_vo); free(w->fe_e1); free(w->fe_e2); free(w->fe_z); free(w->fe_g); free(w->vu); free(w->actf);
}

static inline int is_zz(const orc_config_t *cfg) { return cfg->sampler == ORC_ZIGZAG || cfg->sampler == ORC_SUZZ || cfg->sampler == ORC_STICKY; }

/* SpeedUpZigZag flow (SpeedUpZigZagSamplers.jl:71-79) with a dual time: x_t = y + (v1 X1(t)) v where
 *   y = x - (v1 x1) v,  c = v1 (y.v),  a = (1 + y.y)/dim - c^2/dim^2,  Y0 = x1 + c/dim,
 *   b_t = (Y0 + sqrt(Y0^2 + a)) exp(sqrt(dim) v1 t),  X1 = (b_t^2 - a)/(2 b_t) - c/dim,
 * every operation in the order written there, the t-derivative by the forward-mode rules of the checker (no fma). */
static double vsum(const orc_config_t *cfg, const double *a, int d);
static void suzz_flow_dual(const orc_config_t *cfg, ws_t *w, const double *x, const double *v, double t) {
    const int d = w->d;
    double *y = w->fe_vo;                                                 /* scratch of its own (fa / fb hold finite-difference values) */
    const double av = v[0] * x[0];
    for (int i = 0; i < d; ++i) { y[i] = x[i] - av * v[i]; w->tmp[i] = y[i] * v[i]; }          /* y, y.v */
    const double yv = vsum(cfg, w->tmp, d);
    for (int i = 0; i < d; ++i) w->tmp[i] = y[i] * y[i];
    const double yy = vsum(cfg, w->tmp, d);
    const double dd = (double)d;
    const double c = v[0] * yv;
    const double a = (1.0 + yy) / dd - (c * c) / (dd * dd);
    const double Y0 = x[0] + c / dd;
    const double base = Y0 + sqrt(Y0 * Y0 + a);
    const double sd = sqrt(dd) * v[0];
    const double ev = orc_exp(sd * t), ee = sd * ev;                      /* exp(sd t), d/dt */
    const double bv = base * ev, be = base * ee;                          /* b_t */
    const double b2v = bv * bv, b2e = be * bv + bv * be;                  /* b_t^2 */
    const double nv = b2v - a, ne = b2e;
    const double dv = 2.0 * bv, de = 2.0 * be;
    const double qv = nv / dv, qe = (ne - qv * de) / dv;
    const double X1v = qv - c / dd, X1e = qe;
    const double kv = v[0] * X1v, ke = v[0] * X1e;                         /* v1 * X1 */
    for (int i = 0; i < d; ++i) {
        w->xv[i] = y[i] + kv * v[i];
        w->xe[i] = ke * v[i];
        w->vv[i] = v[i];
        w->ve[i] = 0.0;
    }
}
/* grad-U_effective(x) = speed(x) grad-U(x) - x / speed(x), speed = sqrt(1 + x.x) (SpeedUpZigZagSamplers.jl:81-83), on duals */
static void suzz_effective(const orc_config_t *cfg, ws_t *w, const double *xv, const double *xe, double *gv, double *ge) {
    const int d = w->d;
    for (int i = 0; i < d; ++i) { w->tmp[i] = xv[i] * xv[i]; w->tmp2[i] = xe[i] * xv[i] + xv[i] * xe[i]; }
    const double s2v = 1.0 + vsum(cfg, w->tmp, d), s2e = vsum(cfg, w->tmp2, d);
    const double spv = sqrt(s2v), spe = s2e / (2.0 * spv);
    for (int i = 0; i < d; ++i) {
        const double mv = spv * gv[i], me = spe * gv[i] + spv * ge[i];   /* speed * grad-U */
        const double qv = xv[i] / spv, qe = (xe[i] - qv * spe) / spv;    /* x / speed */
        gv[i] = mv - qv;
        ge[i] = me - qe;
    }
}

/* flow with a dual time (t, 1): ZigZagSamplers.jl:80, BouncyParticleSamplers.jl:32 (x + v t, v);
 * BoomerangSamplers.jl:31 (x cos t + v sin t, -x sin t + v cos t). */
static void flow_dual(const orc_config_t *cfg, ws_t *w, const double *x, const double *v, double t) {
    if (cfg->sampler == ORC_SUZZ) { suzz_flow_dual(cfg, w, x, v, t); return; }
    const int d = w->d;
    if (cfg->sampler == ORC_BOOMERANG) {
        double s, c; orc_sincos(t, &s, &c);
        /* d/dt: cos -> -sin, sin -> cos */
        for (int i = 0; i < d; ++i) {
            w->xv[i] = x[i] * c + v[i] * s;
            w->xe[i] = x[i] * (-s) + v[i] * c;
            w->vv[i] = (-x[i]) * s + v[i] * c;
            w->ve[i] = (-x[i]) * c + v[i] * (-s);
        }
    } else {
        for (int i = 0; i < d; ++i) {
            w->xv[i] = FUSED(cfg) ? fma(v[i], t, x[i]) : x[i] + v[i] * t;
            w->xe[i] = v[i];
            w->vv[i] = v[i];
            w->ve[i] = 0.0;
        }
    }
}

/* Vector-valued bounded function and its t-derivative at time t.
 * signed_rate_vect ZigZagSamplers.jl:95-98 / rate_vect :88-91. */
static void zz_func_dual(const orc_config_t *cfg, ws_t *w, const double *x, const double *v, double t,
                         double *val, double *slp) {
    flow_dual(cfg, w, x, v, t);
    grad_dual(cfg, w->xv, w->xe, w->gv, w->ge);
    if (cfg->sampler == ORC_SUZZ) suzz_effective(cfg, w, w->xv, w->xe, w->gv, w->ge);
    for (int i = 0; i < w->d; ++i) {
        double s = w->gv[i] * w->vv[i];
        double sd = w->ge[i] * w->vv[i] + w->gv[i] * w->ve[i];
        if (cfg->signed_bound) { val[i] = s; if (slp) slp[i] = sd; }
        else { val[i] = pmax(0.0, s); if (slp) slp[i] = (0.0 < s) ? sd : 0.0; }
    }
}
/* Scalar bounded function: signed_rate / rate of BPS (BouncyParticleSamplers.jl:39-47) and
 * Boomerang (BoomerangSamplers.jl:38-46).  Includes refresh_rate (Q2). */
static double sc_func_dual(const orc_config_t *cfg, ws_t *w, const double *x, const double *v, double t, double *slp) {
    flow_dual(cfg, w, x, v, t);
    grad_dual(cfg, w->xv, w->xe, w->gv, w->ge);
    if (cfg->sampler == ORC_SUZZ) suzz_effective(cfg, w, w->xv, w->xe, w->gv, w->ge);
    for (int i = 0; i < w->d; ++i) {
        w->tmp[i] = w->gv[i] * w->vv[i];
        w->tmp2[i] = w->ge[i] * w->vv[i] + w->gv[i] * w->ve[i];
    }
    if (is_zz(cfg)) {                            /* non-vectorized Zig-Zag: func = rate = sum(max.(0, .)), ZigZagSamplers.jl:83-86 */
        for (int i = 0; i < w->d; ++i) {
            double s = w->tmp[i];
            w->tmp[i] = pmax(0.0, s);
            w->tmp2[i] = (0.0 < s) ? w->tmp2[i] : 0.0;
        }
        if (slp) *slp = vsum(cfg, w->tmp2, w->d);
        return vsum(cfg, w->tmp, w->d);
    }
    double dv = vsum(cfg, w->tmp, w->d), de = vsum(cfg, w->tmp2, w->d);
    if (cfg->signed_bound) { if (slp) *slp = de; return dv + cfg->refresh_rate; }
    if (slp) *slp = (0.0 < dv) ? de : 0.0;
    return pmax(0.0, dv) + cfg->refresh_rate;
}

/* finite_difference_derivative, UpperBound.jl:50-76 (start = 0, horizon given) */
static void fd_points(double t, double horizon, double *tm, double *tp) {
    double h = 1.4901161193847656e-08 /* sqrt(eps(Float64)) */ * pmax(1.0, fabs(t));
    *tm = pmax(0.0, t - h);
    *tp = (horizon < t + h) ? horizon : t + h;      /* min(horizon, x+h) */
}

/* "Affine tail" (GROUP order only; engine-order choice): BPS with forward-mode slopes on a potential whose gradient
 * components i >= aff are affine in their own coordinate (isotropic / diagonal Gaussian: aff = 0; banana: aff = 2).  Along the
 * linear flow the bounded function g(x + v t).v of BouncyParticleSamplers.jl:39-47 is then  head(t) + A + B t  with
 *   A = sum_{i >= aff} g_i(x) v_i,   B = sum_{i >= aff} (dg_i/dx_i v_i) v_i   (one pass over the coordinates per state, summed in the
 *   GROUP order: per lane left to right over i = g, g+G, ..., then the butterfly), and
 *   head(t) = sum_{i < aff} g_i(x_t) v_i on forward-mode numbers -- two coordinates for the banana, evaluated in full at every time.
 * Value and t-derivative at a grid time: f = head.v + fma(B, t, A), f' = head.e + B.  The reference evaluates all d coordinates
 * at every grid time; the two agree up to the rounding of the sums (1e-5 tolerance test, tests/test_gpu_reference_order.py). */
static int lr_mode(const orc_config_t *cfg) {
    return cfg->arith == ORC_ARITH_GROUP && !cfg->no_identities && (cfg->sampler == ORC_BPS || cfg->sampler == ORC_FECMC) && cfg->deriv_mode == ORC_DERIV_FORWARD &&
           (cfg->target_id == ORC_T_GAUSS_ISO || cfg->target_id == ORC_T_GAUSS_DIAG || cfg->target_id == ORC_T_BANANA);
}
static int lr_aff(const orc_config_t *cfg) { return (cfg->target_id == ORC_T_BANANA && cfg->dim >= 2) ? 2 : 0; }
static void lr_coeffs(const orc_config_t *cfg, ws_t *w, const double *x, const double *v, double *A, double *B) {
    const int d = w->d, aff = lr_aff(cfg);
    grad_dual(cfg, x, v, w->gv, w->ge);                     /* g_i(x) and dg_i/dx_i v_i of the affine components (head entries unused) */
    for (int i = 0; i < d; ++i) {
        w->tmp[i] = (i >= aff) ? w->gv[i] * v[i] : 0.0;
        w->tmp2[i] = (i >= aff) ? w->ge[i] * v[i] : 0.0;
    }
    *A = sum_group(w->tmp, d, cfg->group_lanes);
    *B = sum_group(w->tmp2, d, cfg->group_lanes);
}
static double lr_func(const orc_config_t *cfg, const double *x, const double *v, double A, double B, double t, double *slp) {
    double fv = fma(B, t, A), fe = B;
    if (lr_aff(cfg)) {                                      /* banana head: coordinates 0 and 1 (test/test_config.jl:33-36) */
        const double x0v = fma(v[0], t, x[0]), x0e = v[0], x1v = fma(v[1], t, x[1]), x1e = v[1];
        double sq_v = x0v * x0v, sq_e = x0e * x0v + x0v * x0e;
        double m_v = sq_v - 1.0, m_e = sq_e;
        double r_v = x1v - m_v, r_e = x1e - m_e;
        double a_v = 2.0 * x0v, a_e = 2.0 * x0e;
        double pr_v = a_v * r_v, pr_e = a_e * r_v + a_v * r_e;
        const double g0v = x0v - pr_v, g0e = x0e - pr_e;
        const double hv = (0.0 + g0v * v[0]) + r_v * v[1], he = (0.0 + g0e * v[0]) + r_e * v[1];
        fv = hv + fv; fe = he + fe;
    }
    if (slp) *slp = fe;
    return fv;
}

/* "Quadratic form" (GROUP order only; engine-order choice): Boomerang on the isotropic Gaussian.  Along the rotation
 * x_t = x cos t + v sin t, v_t = -x sin t + v cos t (BoomerangSamplers.jl:31) the bounded function grad-U(x_t).v_t = x_t.v_t of
 * :38-46 is  XV (c^2 - s^2) + (VV - XX) s c  with the three dot products XX = x.x, XV = x.v, VV = v.v of the state (GROUP order:
 * per lane left to right, then the butterfly), its t-derivative  XV (-4 s c) + (VV - XX) (c^2 - s^2).  One pass over the
 * coordinates per state instead of one per grid time (and per finite-difference point: 30 per bound with AD_backend="FiniteDiff"). */
static int lrb_mode(const orc_config_t *cfg) {
    return cfg->arith == ORC_ARITH_GROUP && !cfg->no_identities && cfg->sampler == ORC_BOOMERANG && cfg->target_id == ORC_T_GAUSS_ISO;
}
static void lrb_coeffs(const orc_config_t *cfg, ws_t *w, const double *x, const double *v, double *XX, double *XV, double *VV) {
    const int d = w->d, G = cfg->group_lanes;
    for (int i = 0; i < d; ++i) w->tmp[i] = x[i] * x[i];
    *XX = sum_group(w->tmp, d, G);
    for (int i = 0; i < d; ++i) w->tmp[i] = x[i] * v[i];
    *XV = sum_group(w->tmp, d, G);
    for (int i = 0; i < d; ++i) w->tmp[i] = v[i] * v[i];
    *VV = sum_group(w->tmp, d, G);
}
static double lrb_form(double XX, double XV, double VV, double sn, double cs, double *de);
static double lrb_raw(double XX, double XV, double VV, double t, double *de) {
    double sn, cs; orc_sincos(t, &sn, &cs);
    return lrb_form(XX, XV, VV, sn, cs, de);
}
/* finite-difference neighbours t + delta of a grid time (|delta| <= 1.5e-8 max(1, t), UpperBound.jl:58-61): sin / cos by the
 * angle-addition formulas from those of t, with sin(delta) = delta, cos(delta) = 1 - delta^2/2 (exact to the last bit at that size) */
static double lrb_raw_shifted(double XX, double XV, double VV, double t, double tt) {
    double sn, cs; orc_sincos(t, &sn, &cs);
    const double dl = tt - t, cd = 1.0 - 0.5 * (dl * dl);
    return lrb_form(XX, XV, VV, sn * cd + cs * dl, cs * cd - sn * dl, NULL);
}
static double lrb_form(double XX, double XV, double VV, double sn, double cs, double *de) {
    const double cc = cs * cs, ss = sn * sn, sc = sn * cs, dq = VV - XX;
    if (de) *de = XV * (-4.0 * sc) + dq * (cc - ss);
    return XV * (cc - ss) + dq * sc;
}
/* the scalar bounded function of the bound (signed_rate / rate incl. refresh_rate, BoomerangSamplers.jl:38-46; Q2) */
static double lrb_sc(const orc_config_t *cfg, double XX, double XV, double VV, double t, double *slp) {
    double de, dv = lrb_raw(XX, XV, VV, t, &de);
    if (cfg->signed_bound) { if (slp) *slp = de; return dv + cfg->refresh_rate; }
    if (slp) *slp = (0.0 < dv) ? de : 0.0;
    return pmax(0.0, dv) + cfg->refresh_rate;
}

/* The unsigned total rate used by the accept test (always `sampler.rate`, SamplingLoopInplace.jl:116; Q3) */
static double true_rate(const orc_config_t *cfg, ws_t *w, double t) {
    if (cfg->sampler == ORC_RHMC) return cfg->refresh_rate;          /* rate = (_x0, _v0, _t) -> refresh_rate, :133 */
    if (lrb_mode(cfg)) {
        double XX, XV, VV;
        lrb_coeffs(cfg, w, w->x, w->v, &XX, &XV, &VV);
        return pmax(0.0, lrb_raw(XX, XV, VV, t, NULL)) + cfg->refresh_rate;
    }
    if (lr_mode(cfg)) {
        double A, B;
        lr_coeffs(cfg, w, w->x, w->v, &A, &B);
        w->lr_raw = lr_func(cfg, w->x, w->v, A, B, t, NULL);
        return pmax(0.0, w->lr_raw) + cfg->refresh_rate;
    }
    flow_dual(cfg, w, w->x, w->v, t);
    grad_dual(cfg, w->xv, w->xe, w->gv, w->ge);
    if (cfg->sampler == ORC_SUZZ) suzz_effective(cfg, w, w->xv, w->xe, w->gv, w->ge);
    if (is_zz(cfg)) {                                 /* ZigZagSamplers.jl:83-86 */
        for (int i = 0; i < w->d; ++i) w->tmp[i] = pmax(0.0, w->gv[i] * w->vv[i]);
        return vsum(cfg, w->tmp, w->d);
    }
    for (int i = 0; i < w->d; ++i) w->tmp[i] = w->gv[i] * w->vv[i];
    return pmax(0.0, vsum(cfg, w->tmp, w->d)) + cfg->refresh_rate;   /* BPS :39-42, Boomerang :38-41 */
}

static double brent_max_rate(const orc_config_t *cfg, ws_t *w, double x_lower, double x_upper);

/* upper_bound_grid_vect (UpperBound.jl:203-247) / upper_bound_grid (:92-137); chosen as in
 * init_state (AbstractPDMP.jl:121-136).  Fills w->grid/box/cum/step. */
static void upper_bound(const orc_config_t *cfg, ws_t *w, const double *x, const double *v, double horizon) {
    const int d = w->d, n = w->n;
    if (cfg->grid_size == 0) {                      /* constant bound, AbstractPDMP.jl:121-125 (x, v are the state: w->x, w->v) */
        /* RHMC: its own init_state builds the 2-point BoundBox of the constant rate (RandomizedHamiltonianMonteCarlo.jl:208-218) */
        const double m = (cfg->sampler == ORC_RHMC) ? cfg->refresh_rate : brent_max_rate(cfg, w, 0.0, horizon);
        w->grid[0] = 0.0; w->grid[1] = horizon;     /* :27 */
        w->box[0] = m;                              /* refresh_rate = 0.0 default: nothing added, :29 */
        w->cum[0] = 0.0; w->cum[1] = m * (horizon - 0.0);   /* :32-33 */
        w->step = horizon - 0.0;                    /* :35 */
        return;
    }
    /* t = range(0, stop=horizon, length=n); step_size = t[2]-t[1]  (:204-205 / :94-95).
     * Build-defined rounding: step = horizon/(n-1), t_j = j*step, t_{n-1} = horizon
     * (Julia's TwicePrecision range may differ in the last bit -- unpinned). */
    const double h = horizon / (double)(n - 1);
    for (int j = 0; j < n; ++j) w->grid[j] = (double)j * h;
    w->grid[n - 1] = horizon;
    w->step = h;
    const double *t = w->grid;

    if (cfg->vectorized_bound) {
        for (int j = 0; j < n; ++j) {
            double *val = w->vals + (size_t)j * d, *slp = w->slps + (size_t)j * d;
            if (cfg->deriv_mode == ORC_DERIV_FORWARD) {
                zz_func_dual(cfg, w, x, v, t[j], val, slp);                   /* :208-210 */
            } else {                                                            /* :223-224, :50-76 */
                zz_func_dual(cfg, w, x, v, t[j], val, NULL);                    /* fx, :56 */
                double tm, tp; fd_points(t[j], horizon, &tm, &tp);
                if (tp == tm) { for (int i = 0; i < d; ++i) slp[i] = val[i] - val[i]; }
                else if (tm == t[j]) { zz_func_dual(cfg, w, x, v, tp, w->fa, NULL);
                    for (int i = 0; i < d; ++i) slp[i] = (w->fa[i] - val[i]) / (tp - t[j]); }
                else if (tp == t[j]) { zz_func_dual(cfg, w, x, v, tm, w->fb, NULL);
                    for (int i = 0; i < d; ++i) slp[i] = (val[i] - w->fb[i]) / (t[j] - tm); }
                else { zz_func_dual(cfg, w, x, v, tp, w->fa, NULL); zz_func_dual(cfg, w, x, v, tm, w->fb, NULL);
                    for (int i = 0; i < d; ++i) slp[i] = (w->fa[i] - w->fb[i]) / (tp - tm); }
            }
        }
        if (cfg->arith == ORC_ARITH_REFERENCE) memset(w->cum, 0, (size_t)n * 8);
        /* coordinates i >= affine_from have d grad_i / dx_j = const * delta_ij (isotropic / diagonal Gaussian: all; banana: i >= 2);
         * the engine decides per slot of G coordinates, hence the rounding to a multiple of G below */
        const int affine_cells = cfg->arith == ORC_ARITH_GROUP && !cfg->no_identities && cfg->sampler == ORC_ZIGZAG && cfg->signed_bound &&
                                 cfg->deriv_mode == ORC_DERIV_FORWARD;
        const int affine_from = (cfg->target_id == ORC_T_GAUSS_ISO || cfg->target_id == ORC_T_GAUSS_DIAG) ? 0
                              : (cfg->target_id == ORC_T_BANANA ? 2 : (1 << 30));
        double *run = w->fa;                       /* per-dimension running cumsum (:243-244) */
        for (int i = 0; i < d; ++i) run[i] = 0.0;
        for (int j = 0; j + 1 < n; ++j) {
            const double *v0 = w->vals + (size_t)j * d, *v1 = v0 + d, *g0 = w->slps + (size_t)j * d, *g1 = g0 + d;
            for (int i = 0; i < d; ++i) {
                /* :229  absolute-time tangent intersection (Q1) */
                if (affine_cells && (i / cfg->group_lanes) * cfg->group_lanes >= affine_from) {
                    /* GROUP order, engine-order choice "affine cell" (DESIGN.md): forward-mode Zig-Zag under the signed bound on a
                     * coordinate whose gradient component is affine in that coordinate -- the slope is the same double at both
                     * ends, so the reference's tangent intersection degenerates (num / 0 -> clamp -> p = h or 0) and `inter` is
                     * v0 or v0 + g h = v1 up to rounding: the cell maximum is max(v0, v1, 0), taken without the detour. */
                    w->cell[i] = pmax(pmax(v0[i], v1[i]), 0.0);
                    continue;
                }
                double num = FUSED(cfg) ? fma(-g0[i], t[j], fma(g1[i], t[j + 1], v0[i] - v1[i]))
                                        : ((v0[i] - v1[i]) + g1[i] * t[j + 1]) - g0[i] * t[j];
                double p = num / (g1[i] - g0[i]);
                if (p != p) p = 0.0;                                   /* :234 */
                p = (p > h) ? h : ((p < 0.0) ? 0.0 : p);               /* :235 clamp */
                double inter = FUSED(cfg) ? fma(g0[i], p, v0[i]) : v0[i] + g0[i] * p;   /* :237 */
                w->cell[i] = pmax(pmax(pmax(v0[i], v1[i]), inter), 0.0); /* :239-241 */
            }
            w->box[j] = vsum(cfg, w->cell, d);                         /* :246 sum(box_max, dims=1) */
            if (cfg->arith == ORC_ARITH_REFERENCE) {
                for (int i = 0; i < d; ++i) { run[i] += w->cell[i]; w->tmp[i] = run[i] * h; }
                w->cum[j + 1] = sum_seq(w->tmp, d);                    /* :244,:246 */
            }
        }
        if (cfg->arith != ORC_ARITH_REFERENCE) {                       /* engine orders: cumsum after the sum */
            double acc = 0.0; w->cum[0] = 0.0;
            for (int j = 0; j + 1 < n; ++j) { acc += w->box[j]; w->cum[j + 1] = acc * h; }
        }
        return;
    }

    /* scalar grid bound, :92-137 */
    double *val = w->vals, *slp = w->slps;
    const double refresh = cfg->signed_bound ? cfg->refresh_rate : 0.0;    /* AbstractPDMP.jl:104-112 */
    double lrA = 0.0, lrB = 0.0, qXX = 0.0, qXV = 0.0, qVV = 0.0;
    if (lr_mode(cfg)) lr_coeffs(cfg, w, x, v, &lrA, &lrB);
    if (lrb_mode(cfg)) lrb_coeffs(cfg, w, x, v, &qXX, &qXV, &qVV);
    for (int j = 0; j < n; ++j) {
        if (lrb_mode(cfg)) {                                                  /* quadratic-form order: both slope modes */
            if (cfg->deriv_mode == ORC_DERIV_FORWARD) { val[j] = lrb_sc(cfg, qXX, qXV, qVV, t[j], &slp[j]); continue; }
            val[j] = lrb_sc(cfg, qXX, qXV, qVV, t[j], NULL);
            double tm, tp; fd_points(t[j], horizon, &tm, &tp);
#define LRB_NB(tt) (cfg->signed_bound ? lrb_raw_shifted(qXX, qXV, qVV, t[j], (tt)) + cfg->refresh_rate \
                                      : pmax(0.0, lrb_raw_shifted(qXX, qXV, qVV, t[j], (tt))) + cfg->refresh_rate)
            if (tp == tm) slp[j] = val[j] - val[j];
            else if (tm == t[j]) slp[j] = (LRB_NB(tp) - val[j]) / (tp - t[j]);
            else if (tp == t[j]) slp[j] = (val[j] - LRB_NB(tm)) / (t[j] - tm);
            else { double fp = LRB_NB(tp), fm = LRB_NB(tm);
                slp[j] = (fp - fm) / (tp - tm); }
#undef LRB_NB
            continue;
        }
        if (lr_mode(cfg)) {                                                   /* affine-tail order: same function, O(1) per grid time */
            double de, dv = lr_func(cfg, x, v, lrA, lrB, t[j], &de);
            if (cfg->signed_bound) { val[j] = dv + cfg->refresh_rate; slp[j] = de; }
            else { val[j] = pmax(0.0, dv) + cfg->refresh_rate; slp[j] = (0.0 < dv) ? de : 0.0; }
        } else if (cfg->deriv_mode == ORC_DERIV_FORWARD) {
            val[j] = sc_func_dual(cfg, w, x, v, t[j], &slp[j]);               /* :97,:102 */
        } else {
            val[j] = sc_func_dual(cfg, w, x, v, t[j], NULL);
            double tm, tp; fd_points(t[j], horizon, &tm, &tp);
            if (tp == tm) slp[j] = val[j] - val[j];
            else if (tm == t[j]) slp[j] = (sc_func_dual(cfg, w, x, v, tp, NULL) - val[j]) / (tp - t[j]);
            else if (tp == t[j]) slp[j] = (val[j] - sc_func_dual(cfg, w, x, v, tm, NULL)) / (t[j] - tm);
            else { double fp = sc_func_dual(cfg, w, x, v, tp, NULL), fm = sc_func_dual(cfg, w, x, v, tm, NULL);
                slp[j] = (fp - fm) / (tp - tm); }
        }
    }
    double acc = 0.0; w->cum[0] = 0.0;
    for (int j = 0; j + 1 < n; ++j) {
        double num = FUSED(cfg) ? fma(slp[j + 1], h, val[j] - val[j + 1]) : (val[j] - val[j + 1]) + slp[j + 1] * h;
        double p = num / (slp[j + 1] - slp[j]);                                      /* :123 relative form */
        if (p != p) p = 0.0;                                                         /* :124 */
        p = (p > h) ? h : ((p < 0.0) ? 0.0 : p);                                     /* :125 */
        double inter = FUSED(cfg) ? fma(slp[j], p, val[j]) : val[j] + slp[j] * p;    /* :127 */
        double b = pmax(pmax(pmax(val[j], val[j + 1]), inter), 0.0);                 /* :128-130 */
        b += refresh;                                                                /* :131 (Q2) */
        w->box[j] = b;
        acc += b; w->cum[j + 1] = acc * h;                                           /* :133-134 */
    }
}

/* upper_bound_constant, UpperBound.jl:18-36: maximise the UNSIGNED rate (AbstractPDMP.jl:121-125) on
 * [0, horizon] with Brent's method and bound it by that one constant (2-point grid, cum_sum = [0, max*horizon]).
 * The minimiser is Optim.jl's univariate Brent() (compat "1.9.4, 2", Project.toml:35 -- third-party, not in the
 * reference tree, version unpinned): Brent (1973) golden-section + parabolic interpolation with Optim's default
 * rel_tol = sqrt(eps), abs_tol = eps, iterations = 1000, restated from the published algorithm. */
static double brent_max_rate(const orc_config_t *cfg, ws_t *w, double x_lower, double x_upper) {
    const double rel_tol = 1.4901161193847656e-08, abs_tol = 2.220446049250313e-16;
    const double golden = 0.5 * (3.0 - 2.23606797749979);      /* (3 - sqrt(5)) / 2 */
    double new_x = x_lower + golden * (x_upper - x_lower);
    double new_f = -true_rate(cfg, w, new_x);                   /* func_min = x -> -func(x), :21 */
    double step = 0.0, old_step = 0.0;
    double old_x = new_x, old_old_x = new_x, old_f = new_f, old_old_f = new_f;
    for (int iteration = 0; iteration < 1000;) {
        double p = 0.0, q = 0.0;
        const double x_tol = rel_tol * fabs(new_x) + abs_tol;
        const double x_mid = (x_upper + x_lower) / 2;
        if (fabs(new_x - x_mid) <= 2 * x_tol - (x_upper - x_lower) / 2) break;        /* converged */
        ++iteration;
        if (fabs(old_step) > x_tol) {                           /* parabolic interpolation through the three best points */
            double r = (new_x - old_x) * (new_f - old_old_f);
            q = (new_x - old_old_x) * (new_f - old_f);
            p = (new_x - old_old_x) * q - (new_x - old_x) * r;
            q = 2 * (q - r);
            if (q > 0) p = -p; else q = -q;
        }
        if (fabs(p) < fabs(q * old_step / 2) && p < q * (x_upper - new_x) && p < q * (new_x - x_lower)) {
            old_step = step;
            step = p / q;
            const double x_temp = new_x + step;                 /* not too close to the ends of the bracket */
            if ((x_temp - x_lower) < 2 * x_tol || (x_upper - x_temp) < 2 * x_tol)
                step = (new_x < x_mid) ? x_tol : -x_tol;
        } else {                                                /* golden section */
            old_step = (new_x < x_mid) ? x_upper - new_x : x_lower - new_x;
            step = golden * old_step;
        }
        const double xt = (fabs(step) >= x_tol) ? new_x + step : new_x + ((step > 0) ? x_tol : -x_tol);
        const double ft = -true_rate(cfg, w, xt);
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
    return -new_f;                                              /* box_max = [-Optim.minimum(result)], :28 */
}

/* next_event, UpperBound.jl:264-273 */
void orc_next_event(int n, const double *grid, const double *box, const double *cum, double step,
                    double E, double *tp, double *lb) {
    int idx = 0;                                   /* searchsortedfirst: first cum[idx] >= E (0-based) */
    while (idx < n && cum[idx] < E) ++idx;
    if (idx >= n) { *tp = INFINITY; *lb = box[n - 2]; return; }
    if (idx == 0) idx = 1;                         /* E <= 0 cannot occur (u < 1); guard */
    *tp = grid[idx - 1] + (E - cum[idx - 1]) / (cum[idx] - cum[idx - 1]) * step;
    *lb = box[idx - 1];
}

double orc_upper_bound(const orc_config_t *cfg, const double *x, const double *v, double horizon,
                       double *grid, double *box_max, double *cum_sum) {
    g_ver = (cfg->rng_version == 1) ? 1 : 2;
    ws_t w; ws_init(&w, cfg->dim, cfg->grid_size == 0 ? 2 : cfg->grid_size);
    memcpy(w.x, x, (size_t)cfg->dim * 8); memcpy(w.v, v, (size_t)cfg->dim * 8);
    upper_bound(cfg, &w, x, v, horizon);
    memcpy(grid, w.grid, (size_t)w.n * 8); memcpy(box_max, w.box, (size_t)(w.n - 1) * 8);
    memcpy(cum_sum, w.cum, (size_t)w.n * 8);
    double s = w.step; ws_free(&w); return s;
}

/* in-place flow of the state (no dual) */
/* velocity-Verlet flow of RHMC (RandomizedHamiltonianMonteCarlo.jl:97-130) on (x, v) in place: n = floor(t/h) steps of size h and
 * one of size r = t - n h; g, xe, ge are d-vector scratch. */
static void rhmc_step(const orc_config_t *cfg, double *x, double *v, int d, double h, double *xe, double *g, double *ge) {
    grad_dual(cfg, x, xe, g, ge);
    for (int i = 0; i < d; ++i) { v[i] = v[i] - (0.5 * h) * g[i]; x[i] = x[i] + h * v[i]; }
    grad_dual(cfg, x, xe, g, ge);
    for (int i = 0; i < d; ++i) v[i] = v[i] - (0.5 * h) * g[i];
}
static void rhmc_flow(const orc_config_t *cfg, double *x, double *v, int d, double t, double *xe, double *g, double *ge) {
    if (t == 0.0) return;
    for (int i = 0; i < d; ++i) xe[i] = 0.0;
    const double h = cfg->rh_step_size;
    const int64_t n = (int64_t)floor(t / h);
    const double r = t - (double)n * h;
    for (int64_t k = 0; k < n; ++k) rhmc_step(cfg, x, v, d, h, xe, g, ge);
    if (r > 0.0) rhmc_step(cfg, x, v, d, r, xe, g, ge);
}

static void flow_state(const orc_config_t *cfg, ws_t *w, double t) {
    if (cfg->sampler == ORC_RHMC) { rhmc_flow(cfg, w->x, w->v, w->d, t, w->xe, w->gv, w->ge); return; }
    if (cfg->sampler == ORC_SUZZ) {                      /* SpeedUpZigZagSamplers.jl:71-79 */
        suzz_flow_dual(cfg, w, w->x, w->v, t);
        for (int i = 0; i < w->d; ++i) w->x[i] = w->xv[i];
        return;
    }
    if (cfg->sampler == ORC_BOOMERANG) {
        double s, c; orc_sincos(t, &s, &c);
        for (int i = 0; i < w->d; ++i) {
            double xn = w->x[i] * c + w->v[i] * s, vn = (-w->x[i]) * s + w->v[i] * c;
            w->x[i] = xn; w->v[i] = vn;
        }
    } else {
        for (int i = 0; i < w->d; ++i) w->x[i] = FUSED(cfg) ? fma(w->v[i], t, w->x[i]) : w->x[i] + w->v[i] * t;
    }
}

/* dot product / norm in the configured summation order (LinearAlgebra's dot and norm are BLAS calls of unspecified order) */
static double vdot(const orc_config_t *cfg, ws_t *w, const double *a, const double *b) {
    for (int i = 0; i < w->d; ++i) w->tmp[i] = a[i] * b[i];
    return vsum(cfg, w->tmp, w->d);
}
static double vnorm(const orc_config_t *cfg, ws_t *w, const double *a) { return sqrt(vdot(cfg, w, a, a)); }

/* _velocity_jump_event_chain, ForwardEventChainMonteCarlo.jl:127-175 (the default variant: normal = false, speed_factor = 1),
 * with _orthogonal_switch (:60-91) / _full_refresh (:108-117).  grad-U(x) is in w->gv on entry.  u^(2/(dim-1)) is restated as
 * exp(log(u) * (2/(dim-1))) with the checker's log / exp; randn(key, 2, dim) is column-major: g1[j], g2[j] = normals 2j, 2j+1. */
static int fecmc_jump(const orc_config_t *cfg, ws_t *w, rng_t *rng) {
    const int d = w->d;
    double *n = w->gv, *vo = w->fe_vo, *e1 = w->fe_e1, *e2 = w->fe_e2, *z = w->fe_z;
    const double u = rng_uniform(rng);                                       /* :134 */
    const double pw = orc_exp(orc_log(u) * (2.0 / (double)(d - 1)));
    const double sf = (cfg->fe_speed_factor == 0.0) ? 1.0 : cfg->fe_speed_factor;   /* _velocity_jump_event_chain_speed_up, :177-218 (1.0: :127-175, same bits) */
    const double rho = sf * (-sqrt(1.0 - pw));                               /* :135 / :185 */
    const double ng = vnorm(cfg, w, n);                                      /* :143-149 */
    if (ng == 0.0) { for (int i = 0; i < d; ++i) n[i] = 0.0; }
    else for (int i = 0; i < d; ++i) n[i] = n[i] / ng;
    const double vn = vdot(cfg, w, w->v, n);                                 /* :152-153 */
    for (int i = 0; i < d; ++i) { double vp = vn * n[i]; vo[i] = w->v[i] - vp; }
    if (vnorm(cfg, w, vo) < 1e-10) {                                         /* :156-159 */
        rng_normals(rng, d, vo);
        const double dn = vdot(cfg, w, vo, n);
        for (int i = 0; i < d; ++i) vo[i] = vo[i] - dn * n[i];
    }
    const double u2 = rng_uniform(rng);                                      /* :161 */
    const double sq = sqrt(sf * sf - rho * rho);                             /* :164 / :207 */
    if (u2 >= cfg->fe_mix_p) {                                               /* :162-166 */
        const double nv = vnorm(cfg, w, vo);
        for (int i = 0; i < d; ++i) w->v[i] = vo[i] / nv * sq + rho * n[i];
        return 0;
    }
    double *prop = w->fe_z;
    if (cfg->fe_switch) {                                                    /* _orthogonal_switch, :60-91 */
        rng_normals(rng, 2 * d, w->fe_g);
        for (int i = 0; i < d; ++i) { e1[i] = w->fe_g[2 * i]; e2[i] = w->fe_g[2 * i + 1]; }
        const double d1 = vdot(cfg, w, e1, n);
        for (int i = 0; i < d; ++i) e1[i] = e1[i] - d1 * n[i];               /* :67 */
        const double d2 = vdot(cfg, w, e2, n);
        for (int i = 0; i < d; ++i) e2[i] = e2[i] - d2 * n[i];               /* :68 */
        const double n1 = vnorm(cfg, w, e1);
        for (int i = 0; i < d; ++i) e1[i] = e1[i] / n1;                      /* :71 */
        const double d21 = vdot(cfg, w, e2, e1);
        for (int i = 0; i < d; ++i) e2[i] = e2[i] - d21 * e1[i];             /* :72 */
        const double n2 = vnorm(cfg, w, e2);
        for (int i = 0; i < d; ++i) e2[i] = e2[i] / n2;                      /* :73 */
        const double a = vdot(cfg, w, vo, e1), b = vdot(cfg, w, vo, e2);
        double sn = 0.0, cs = 0.0;
        if (cfg->fe_ran_p) { const double ut = rng_uniform(rng); orc_sincos_2pi(ut, &sn, &cs); }   /* theta = rand * 2pi, :79 */
        for (int i = 0; i < d; ++i) {
            const double vr = (vo[i] - a * e1[i]) - b * e2[i];               /* :76 */
            if (cfg->fe_ran_p) z[i] = (vr + (cs * e1[i] + sn * e2[i]) * a) + (sn * e1[i] - cs * e2[i]) * b;   /* :80 */
            else z[i] = (vr + e2[i] * a) + e1[i] * b;                        /* :77 */
        }
        if (cfg->fe_positive) {                                              /* :83-85 */
            const double sdot = vdot(cfg, w, vo, z);
            const double sg = (sdot > 0.0) ? 1.0 : ((sdot < 0.0) ? -1.0 : sdot);
            for (int i = 0; i < d; ++i) z[i] = z[i] * sg;
        }
    } else {                                                                 /* _full_refresh, :108-117 */
        rng_normals(rng, d, z);
        const double nz = vnorm(cfg, w, z);
        for (int i = 0; i < d; ++i) z[i] = z[i] / nz;
        const double dn = vdot(cfg, w, z, n);
        for (int i = 0; i < d; ++i) z[i] = z[i] - dn * n[i];
    }
    const double np = vnorm(cfg, w, prop);                                   /* :170-172 */
    for (int i = 0; i < d; ++i) w->v[i] = prop[i] / np * sq + rho * n[i];
    return 0;
}

/* velocity_jump: ZigZagSamplers.jl:101-107, BouncyParticleSamplers.jl:50-74, BoomerangSamplers.jl:49-67 */
static int velocity_jump(const orc_config_t *cfg, ws_t *w, rng_t *rng) {
    const int d = w->d;
    if (cfg->sampler == ORC_RHMC) {                /* Horowitz refreshment v <- cos(phi) v + sin(phi) xi, :144-149 */
        double sphi, cphi;
        orc_sincos(cfg->rh_phi, &sphi, &cphi);
        rng_normals(rng, d, w->tmp);
        for (int i = 0; i < d; ++i) w->v[i] = cphi * w->v[i] + sphi * w->tmp[i];
        return 0;
    }
    for (int i = 0; i < d; ++i) { w->xe[i] = 0.0; }
    grad_dual(cfg, w->x, w->xe, w->gv, w->ge);
    if (cfg->sampler == ORC_SUZZ) suzz_effective(cfg, w, w->x, w->xe, w->gv, w->ge);     /* SpeedUpZigZagSamplers.jl:104 */
    if (is_zz(cfg)) {
        for (int i = 0; i < d; ++i) w->tmp[i] = pmax(0.0, w->gv[i] * w->v[i]);
        double S = vsum(cfg, w->tmp, d);
        double u = rng_uniform(rng);                          /* Categorical draw (1 uniform) */
        if (!(S > 0.0)) return 1;                             /* Q11: Categorical would throw */
        double thr = u;
        if (cfg->arith == ORC_ARITH_GROUP) thr = u * S;       /* engine order: un-normalised walk against u*S */
        else for (int i = 0; i < d; ++i) w->tmp[i] = w->tmp[i] / S;
        int m = categorical(cfg, w->tmp, d, thr);
        w->v[m] = -w->v[m];                                   /* v[m] *= -1 */
        return 0;
    }
    if (cfg->sampler == ORC_FECMC) return fecmc_jump(cfg, w, rng);
    double *g = w->gv;
    if (cfg->sampler == ORC_BOOMERANG) for (int i = 0; i < d; ++i) g[i] = g[i] - w->x[i];   /* :51-52 (Q8) */
    for (int i = 0; i < d; ++i) w->tmp[i] = g[i] * w->v[i];
    double gv_dot = vsum(cfg, w->tmp, d);
    if (lr_mode(cfg)) gv_dot = w->lr_raw;                     /* the engine reuses the accepted proposal's g.v (same point, same state) */
    double bounce = pmax(0.0, gv_dot);
    double prob = bounce / (bounce + cfg->refresh_rate);      /* NaN when both are 0 (Q9) */
    double u = rng_uniform(rng);
    if (u < prob) {
        for (int i = 0; i < d; ++i) w->tmp[i] = g[i] * g[i];
        double gg = vsum(cfg, w->tmp, d);
        if (cfg->sampler == ORC_BPS) {
            if (gg == 0.0) return 0;
            double scale = 2.0 * gv_dot / gg;                 /* :60 */
            for (int i = 0; i < d; ++i) w->v[i] = w->v[i] - scale * g[i];
        } else {
            double nrm = sqrt(gg);                            /* e = g/|g|, :58 */
            for (int i = 0; i < d; ++i) { w->ge[i] = g[i] / nrm; w->tmp[i] = w->v[i] * w->ge[i]; }
            double ve = 2.0 * vsum(cfg, w->tmp, d);
            for (int i = 0; i < d; ++i) w->v[i] = w->v[i] - ve * w->ge[i];   /* :59 */
        }
    } else {
        rng_normals(rng, d, w->v);                            /* randn!, :67 / randn(dim), :65 (chain stream, Q8) */
        if (cfg->sampler == ORC_BPS && !cfg->gaussian_velocity) {
            for (int i = 0; i < d; ++i) w->tmp[i] = w->v[i] * w->v[i];
            double inv = 1.0 / sqrt(vsum(cfg, w->tmp, d));    /* normalize!: rmul!(v, inv(norm(v))) */
            for (int i = 0; i < d; ++i) w->v[i] = w->v[i] * inv;
        }
    }
    return 0;
}

int orc_normalize_config(orc_config_t *c) {
    if (c->dim <= 0) FAIL(-1, "dimension dim must be positive. Current value: %d", c->dim);           /* ZigZagSamplers.jl:62-64 */
    if (c->grid_size < 0) FAIL(-1, "grid_size must be non-negative. Current value: %d", c->grid_size); /* :66-68 */
    if (c->grid_size == 1) FAIL(-1, "grid_size == 1: range(...)[2] is a BoundsError in the reference");
    if (c->sampler < 0 || c->sampler > 6) FAIL(-1, "unknown sampler id %d", c->sampler);
    if (c->sampler == ORC_STICKY && (c->n_target_params < c->dim || !c->target_params)) FAIL(-1, "StickyZigZag needs kappa[dim] as the last dim target_params");
    if (c->sampler == ORC_RHMC) {                                                                     /* RandomizedHamiltonianMonteCarlo.jl:59-95 */
        if (!(c->refresh_rate > 0.0) || !isfinite(c->refresh_rate)) FAIL(-1, "refresh_rate must be finite and positive. Current value: %g", c->refresh_rate);
        if (!(0.0 < c->rh_phi && c->rh_phi <= 1.5707963267948966)) FAIL(-1, "phi must satisfy 0 < phi <= pi/2. Current value: %g", c->rh_phi);
        if (!(c->rh_step_size > 0.0) || !isfinite(c->rh_step_size)) FAIL(-1, "step_size must be finite and positive. Current value: %g", c->rh_step_size);
        c->grid_size = 0; c->vectorized_bound = 0; c->signed_bound = 0;
    }
    if (c->sampler == ORC_FECMC) {                                                                    /* ForwardEventChainMonteCarlo.jl:298-310 */
        if (c->dim < 2) FAIL(-1, "The dimension must be at least 2 to use the ForwardEventChain. Got dimension %d", c->dim);
        if (c->dim == 2) c->fe_mix_p = 0.0;
        c->refresh_rate = 0.0;
    }
    if (c->target_id < 0 || (c->target_id > 3 && (c->target_id < ORC_T_AD_BANANA || c->target_id > ORC_T_AD_RIDGED))) FAIL(-1, "unknown target id %d", c->target_id);
    if (c->tmax == 0.0) { c->tmax = 1.0; c->adaptive = 1; }                                           /* :73 */
    if (c->sampler == ORC_ZIGZAG || c->sampler == ORC_SUZZ || c->sampler == ORC_STICKY) {
        if (c->signed_bound && !c->vectorized_bound) c->signed_bound = 0;                             /* :75-78 */
    } else c->vectorized_bound = 0;                                                                  /* BPS :37, Boomerang :36 */
    if (c->arith == ORC_ARITH_GROUP) {
        int G = c->group_lanes;
        if (G < 1 || G > 64 || (G & (G - 1))) FAIL(-1, "group_lanes must be a power of two in 1..64");
    }
    if (c->target_id == ORC_T_GAUSS_DIAG && c->n_target_params != (2 + (c->sampler == ORC_STICKY)) * (int64_t)c->dim) FAIL(-1, "GAUSS_DIAG needs 2*dim params");
    if (c->target_id == ORC_T_GAUSS_DENSE && c->n_target_params != (int64_t)c->dim * c->dim) FAIL(-1, "GAUSS_DENSE needs dim^2 params");
    return 0;
}

/* ---- StickyZigZag (src/StickySamplingLoop.jl).  Sums over the frozen coordinates in the configured order; the thawing draw walks
 * the frozen coordinates in index order (`acc >= u`, :150-158), in GROUP order by slots of G with a scan inside the slot. */
static double sticky_thaw_rate(const orc_config_t *cfg, ws_t *w, const double *kappa) {
    for (int i = 0; i < w->d; ++i) w->tmp[i] = (w->actf[i] != 0.0) ? 0.0 : kappa[i];
    return vsum(cfg, w->tmp, w->d);
}
static int sticky_thaw_pick(const orc_config_t *cfg, ws_t *w, const double *kappa, double u) {
    const int d = w->d;
    int last = -1;
    if (cfg->arith == ORC_ARITH_REFERENCE) {
        double acc = 0.0;
        for (int j = 0; j < d; ++j) if (w->actf[j] == 0.0) { last = j; acc += kappa[j]; if (acc >= u) return j; }
        return last;
    }
    const int W = (cfg->arith == ORC_ARITH_GROUP) ? cfg->group_lanes : 64;
    double carry = 0.0;
    for (int base = 0; base < d; base += W) {
        double y[64], z[64];
        for (int l = 0; l < W; ++l) y[l] = (base + l < d && w->actf[base + l] == 0.0) ? kappa[base + l] : 0.0;
        for (int off = 1; off < W; off <<= 1) {
            for (int l = 0; l < W; ++l) z[l] = (l >= off) ? y[l] + y[l - off] : y[l];
            memcpy(y, z, sizeof(double) * (size_t)W);
        }
        for (int l = 0; l < W && base + l < d; ++l)
            if (w->actf[base + l] == 0.0) { last = base + l; if (carry + y[l] >= u) return base + l; }
        carry = carry + y[W - 1];
    }
    return last;
}

/* One chain: sample_skeleton (sample.jl:253-284) = init_state (AbstractPDMP.jl:93-153, the unused
 * first bound of :140 skipped, Q14) + (n_sk-1) x get_event_state! (SamplingLoopInplace.jl:27-39). */
static void flow_plain(const orc_config_t *cfg, int d, const double *x0, const double *v0, double tau, double *xo, double *vo);

/* T_until < 0: the fixed-count loop of sample.jl:253-284.  T_until >= 0: the time-horizon variant sample_skeleton(sampler,
 * T::Float64, ...), sample.jl:323-439 -- events are recorded while state.t <= T (:372-378); the first event beyond T is replaced
 * by ONE closing column at exactly t = T, obtained by the sampler's flow from the previous state (:379-420; StickyPDMP: with
 * v_used = active_prev ? v_prev : 0 and is_active = active_prev, :386-401) with the statistics of a non-event point (zeros,
 * :402-408); an event at exactly t == T ends the loop without a closing column (`while state.t < T`, :362).  n_sk is the
 * capacity: *count_out = number of columns, or -(columns so far) - 1 when the capacity ran out before T (the caller doubles it,
 * _grow_history, Composites.jl:172-191: same columns, same stream). */
static int run_chain(const orc_config_t *cfg, ws_t *w, int64_t n_sk, uint64_t seed, uint64_t chain,
                     const double *xinit, const double *vinit,
                     double *X, double *V, double *T, double *H, double *AR, int32_t *EB, double *EV,
                     int32_t *RJ, int32_t *HH, uint8_t *IA, double T_until, int64_t *count_out) {
    const int d = w->d, n = w->n;
    const int64_t cap = cfg->max_iter_per_event > 0 ? cfg->max_iter_per_event : 1000000;
    int overrun = 0;
    g_ver = (cfg->rng_version == 1) ? 1 : 2;        /* (thread-local: this chain's arithmetic contract) */
    rng_t rng = { .seed = seed, .chain = chain, .overrun = &overrun };
    jl_mt_t jl_state;
#define JL_COUNTS() do { if (cfg->julia_rng && cfg->jl_count) { int64_t *q_ = cfg->jl_count + 3 * (int64_t)chain; q_[0] = rng.je; q_[1] = rng.ju; q_[2] = rng.jn; } } while (0)
    if (cfg->julia_rng) {                           /* chain c <- MersenneTwister(seed + c) */
        jl_mt_seed(&jl_state, (int64_t)(seed + chain), cfg->julia_rng & 3);
        rng.jl = &jl_state;
        rng.jl_scalar = (cfg->julia_rng & 4) != 0;  /* ORC_JL_SCALAR_RANDN: vectors of normals as scalar draws (an rng wrapped around the MersenneTwister) */
        if (cfg->jl_tape_exp) {
            const int64_t c = (int64_t)chain;
            rng.le = cfg->jl_tape_exp + c * cfg->jl_cap_exp; rng.ce = cfg->jl_cap_exp;
            rng.lu = cfg->jl_tape_uni + c * cfg->jl_cap_uni; rng.cu = cfg->jl_cap_uni;
            rng.ln = cfg->jl_tape_nrm + c * cfg->jl_cap_nrm; rng.cn = cfg->jl_cap_nrm;
        }
    } else
    if (cfg->tape_exp) {
        const int64_t c = (int64_t)chain;          /* tapes are indexed by the global chain id */
        rng.te = cfg->tape_exp + c * cfg->tape_n_exp; rng.ne = cfg->tape_n_exp;
        rng.tu = cfg->tape_uni + c * cfg->tape_n_uni; rng.nu = cfg->tape_n_uni;
        rng.tn = cfg->tape_nrm + c * cfg->tape_n_nrm; rng.nn = cfg->tape_n_nrm;
    }
    memcpy(w->x, xinit, (size_t)d * 8); memcpy(w->v, vinit, (size_t)d * 8);   /* copy: fixes aliasing Q7 */
    for (int i = 0; i < d; ++i) w->actf[i] = 1.0;                              /* is_active = trues(d), Composites.jl:95 */
    double t = 0.0, horizon = cfg->tmax, ts = 0.0, tp = 0.0, E = 0.0, lb = 0.0, ar = 0.0;
    int status = 0;
    const int until = T_until >= 0.0;
    double prev_buf[until ? 3 * d : 1];             /* x_prev, v_prev, active_prev (:356-360) */
    double *x_prev = prev_buf, *v_prev = prev_buf + (until ? d : 0), *a_prev = prev_buf + (until ? 2 * d : 0), t_prev = 0.0;
    if (until && count_out) *count_out = 0;
    for (int64_t k = 0; k < n_sk; ++k) {
        int32_t errored = 0, rejected = 0, hits = 0; double ev[5] = { 0, 0, 0, 0, 0 };   /* :28-31 */
        if (until && k > 0) {
            if (!(t < T_until)) { if (count_out) *count_out = k; JL_COUNTS(); return overrun ? -6 : 0; }   /* while state.t < T, :362 (T == 0: :350-353) */
            memcpy(x_prev, w->x, (size_t)d * 8); memcpy(v_prev, w->v, (size_t)d * 8); memcpy(a_prev, w->actf, (size_t)d * 8);   /* :363-368 */
            t_prev = t;
        }
        if (k > 0 && cfg->sampler == ORC_STICKY) {
            /* get_event_state(state, sampler::StickyPDMP), StickySamplingLoop.jl:13-29: an accepted flip, the crossing of an axis
             * (the coordinate sticks) and the thawing of a coordinate are all events. */
            const double *kappa = cfg->target_params + (cfg->n_target_params - d);
            int accept = 0, event = 0; int64_t it = 0;
#define SET_VU() do { for (int q_ = 0; q_ < d; ++q_) w->vu[q_] = (w->actf[q_] != 0.0) ? w->v[q_] : 0.0; } while (0)
#define WITH_VU(stmt) do { double *vs_ = w->v; SET_VU(); w->v = w->vu; stmt; w->v = vs_; } while (0)
            while (!accept && !event) {                                               /* :23-25 */
                if (++it > cap) return -4;
                if (overrun) return -6;
                /* one_step_of_thinning_or_sticking_or_thawing, :31-67 */
                WITH_VU(upper_bound(cfg, w, w->x, w->vu, horizon));                   /* (the Brent bound reads the state's velocity: masked too) */
                E = rng_exp(&rng);
                orc_next_event(n, w->grid, w->box, w->cum, w->step, E, &tp, &lb);
                const double rate_thaw = sticky_thaw_rate(cfg, w, kappa);             /* :40-45 */
                const double tt = (rate_thaw == 0.0) ? INFINITY : rng_exp(&rng) / rate_thaw;   /* :46 */
                double event_time = tp < horizon ? tp : horizon;                      /* min(tp, horizon, tt), :50 */
                if (tt < event_time) event_time = tt;
                int crossed = 0;
                for (int i = 0; i < d && !crossed; ++i) {                             /* :51-58 */
                    const double xe = FUSED(cfg) ? fma(w->vu[i], event_time, w->x[i]) : w->x[i] + w->vu[i] * event_time;
                    if (w->x[i] * xe < 0.0) crossed = 1;
                }
                if (crossed) {                                                        /* move_to_axes_and_stick, :73-107 */
                    double t_togo = INFINITY; int ia = -1;
                    for (int j = 0; j < d; ++j) if (w->actf[j] != 0.0) {
                        const double dj = w->x[j] * w->v[j];
                        if (dj < 0.0) { const double tj = -dj; if (tj < t_togo) { t_togo = tj; ia = j; } }
                    }
                    if (ia < 0) FAIL(-7, "erronous t_togo, although no axis is crossed");
                    WITH_VU(flow_state(cfg, w, t_togo));
                    w->actf[ia] = 0.0;
                    t = t + (t_togo + ts); ts = 0.0;                                  /* state.t += t_togo + state.ts, :101 */
                    event = 1;
                } else if ((tp < tt ? tp : tt) > horizon) {                           /* move_to_horizon, :61-62 */
                    WITH_VU(flow_state(cfg, w, horizon));
                    ts += horizon; hits++;
                    if (cfg->adaptive) horizon = horizon * 1.01;
                } else {                                                              /* moves_until_horizon_or_axes, :120-131 */
                    while ((tp < tt ? tp : tt) < horizon && !accept && !event) {
                        if (++it > cap) return -4;
                        if (overrun) return -6;
                        if (tp < tt) {                                                /* ac_step (SamplingLoopInplace.jl:113-168 with v .* is_active) */
                            double lam; WITH_VU(lam = true_rate(cfg, w, tp));
                            ar = lam / lb;
                            if (ar > 1.0) {
                                double h2 = horizon / 2;
                                WITH_VU(upper_bound(cfg, w, w->x, w->vu, h2));
                                E = rng_exp(&rng);
                                orc_next_event(n, w->grid, w->box, w->cum, w->step, E, &tp, &lb);
                                if (cfg->adaptive) horizon = h2;
                                errored++;
                                ev[errored % 5] = ar;
                            } else {
                                double u = rng_uniform(&rng);
                                accept = u < ar;
                                if (accept) {
                                    WITH_VU(flow_state(cfg, w, tp));
                                    status |= velocity_jump(cfg, w, &rng);            /* on the full v, :180 */
                                    t = t + tp + ts; ts = 0.0; tp = 0.0;
                                } else {
                                    E = E + rng_exp(&rng);
                                    orc_next_event(n, w->grid, w->box, w->cum, w->step, E, &tp, &lb);
                                    if (cfg->adaptive) horizon = horizon / 1.04;
                                    rejected++;
                                }
                                if (!accept && (tp < tt ? tp : tt) > horizon) {       /* move_to_horizon2!, :163-165 */
                                    WITH_VU(flow_state(cfg, w, horizon));
                                    ts += horizon; hits++;
                                }
                            }
                        } else {                                                      /* thaw_one_coordinate, :138-164 */
                            WITH_VU(flow_state(cfg, w, tt));
                            const double total = sticky_thaw_rate(cfg, w, kappa);
                            const double u = rng_uniform(&rng) * total;
                            const int ia = sticky_thaw_pick(cfg, w, kappa, u);
                            if (ia >= 0) w->actf[ia] = 1.0;
                            t = t + tt; ts = 0.0;                                     /* state.t += state.tt (ts is dropped, :160-161) */
                            event = 1;
                        }
                    }
                }
            }
#undef WITH_VU
#undef SET_VU
        } else if (k > 0) {
            int accept = 0; int64_t it = 0;
            while (!accept) {                                                         /* :33-35 */
                if (++it > cap) return -4;
                if (overrun) return -6;
                /* one_step_of_thinning!, :65-85 */
                upper_bound(cfg, w, w->x, w->v, horizon);
                E = rng_exp(&rng);
                orc_next_event(n, w->grid, w->box, w->cum, w->step, E, &tp, &lb);
                if (tp > horizon) {                                                   /* move_to_horizon!, :87-101 */
                    flow_state(cfg, w, horizon);
                    ts += horizon; hits++;
                    if (cfg->adaptive) horizon = horizon * 1.01;
                } else {                                                              /* moves_until_horizon!, :103-111 */
                    while (tp < horizon && !accept) {
                        if (++it > cap) return -4;
                        if (overrun) return -6;
                        double lam = true_rate(cfg, w, tp);                           /* ac_step!, :113-129 */
                        ar = lam / lb;
                        if (ar > 1.0) {                                               /* erroneous_acceptance_rate!, :131-151 */
                            double h2 = horizon / 2;
                            upper_bound(cfg, w, w->x, w->v, h2);
                            E = rng_exp(&rng);
                            orc_next_event(n, w->grid, w->box, w->cum, w->step, E, &tp, &lb);
                            if (cfg->adaptive) horizon = h2;
                            errored++;
                            ev[errored % 5] = ar;                                     /* :148, 1-based errored%5+1 */
                        } else {                                                      /* ac_step_with_proxy!, :153-168 */
                            double u = rng_uniform(&rng);
                            accept = u < ar;
                            if (accept) {                                             /* if_accept!, :170-186 */
                                flow_state(cfg, w, tp);
                                status |= velocity_jump(cfg, w, &rng);
                                t = t + tp + ts; ts = 0.0; tp = 0.0;
                            } else {                                                  /* if_reject!, :188-203 */
                                E = E + rng_exp(&rng);
                                orc_next_event(n, w->grid, w->box, w->cum, w->step, E, &tp, &lb);
                                if (cfg->adaptive) horizon = horizon / 1.04;
                                rejected++;
                            }
                            if (!accept && tp > horizon) {                            /* move_to_horizon2!, :205-217 (tt = Inf) */
                                flow_state(cfg, w, horizon);
                                ts += horizon; hits++;
                            }
                        }
                    }
                }
            }
        }
        if (until && k > 0 && t > T_until) {                                          /* overshoot, :379-420 */
            const double tau = T_until - t_prev;
            for (int i = 0; i < d; ++i) {
                w->actf[i] = a_prev[i];                                               /* :399-401 */
                v_prev[i] = (cfg->sampler == ORC_STICKY && a_prev[i] == 0.0) ? 0.0 : v_prev[i];   /* v_used, :386-392 */
            }
            flow_plain(cfg, d, x_prev, v_prev, tau, w->x, w->v);                      /* xT, vT = sampler.flow(x_prev, v_used, tau), :394-396 */
            t = T_until;
            ar = 0.0; errored = 0; rejected = 0; hits = 0;                            /* :402-408 (the local ar is not read again) */
            for (int q = 0; q < 5; ++q) ev[q] = 0.0;
            memcpy(X + (size_t)k * d, w->x, (size_t)d * 8);
            memcpy(V + (size_t)k * d, w->v, (size_t)d * 8);
            T[k] = t;
            if (H) H[k] = horizon;
            if (AR) AR[k] = 0.0;
            if (EB) EB[k] = 0;
            if (EV) memcpy(EV + 5 * (size_t)k, ev, 40);
            if (RJ) RJ[k] = 0;
            if (HH) HH[k] = 0;
            if (IA) for (int i = 0; i < d; ++i) IA[(size_t)k * d + i] = (uint8_t)(w->actf[i] != 0.0);
            if (count_out) *count_out = k + 1;
            JL_COUNTS();
            return overrun ? -6 : 0;
        }
        /* record!, Composites.jl:239-260 */
        memcpy(X + (size_t)k * d, w->x, (size_t)d * 8);
        memcpy(V + (size_t)k * d, w->v, (size_t)d * 8);
        T[k] = t;
        if (H) H[k] = horizon;
        if (AR) AR[k] = ar;
        if (EB) EB[k] = errored;
        if (EV) memcpy(EV + 5 * (size_t)k, ev, 40);
        if (RJ) RJ[k] = rejected;
        if (HH) HH[k] = hits;
        if (IA) for (int i = 0; i < d; ++i) IA[(size_t)k * d + i] = (uint8_t)(w->actf[i] != 0.0);   /* Composites.jl:256 */
    }
    (void)status;
    if (until) {                                    /* the capacity ran out (or the last column sits at t >= T exactly) */
        if (count_out) *count_out = (t < T_until) ? -n_sk - 1 : n_sk;
    }
    JL_COUNTS();
    return overrun ? -6 : 0;
#undef JL_COUNTS
}

static int sample_skeleton_impl(const orc_config_t *cfg_in, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                        const double *xinit, const double *vinit, uint64_t seed,
                        double *X, double *V, double *t, double *horizon, double *ar,
                        int32_t *errored_bound, double *error_value_ar, int32_t *rejected,
                        int32_t *hitting_horizon, int n_threads, double T_until, int64_t *count);

int orc_sample_skeleton(const orc_config_t *cfg_in, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                        const double *xinit, const double *vinit, uint64_t seed,
                        double *X, double *V, double *t, double *horizon, double *ar,
                        int32_t *errored_bound, double *error_value_ar, int32_t *rejected,
                        int32_t *hitting_horizon, int n_threads) {
    return sample_skeleton_impl(cfg_in, C, n_sk, chain_id_offset, xinit, vinit, seed, X, V, t, horizon, ar, errored_bound,
                                error_value_ar, rejected, hitting_horizon, n_threads, -1.0, NULL);
}

/* sample_skeleton(sampler, T::Float64, xinit, vinit), sample.jl:323-439, with `capacity` history columns per chain */
int orc_sample_skeleton_until(const orc_config_t *cfg_in, int64_t C, int64_t capacity, int64_t chain_id_offset,
                              const double *xinit, const double *vinit, uint64_t seed, double T_end,
                              double *X, double *V, double *t, double *horizon, double *ar,
                              int32_t *errored_bound, double *error_value_ar, int32_t *rejected,
                              int32_t *hitting_horizon, int64_t *count, int n_threads) {
    if (!(T_end >= 0.0) || !isfinite(T_end)) FAIL(-1, "T must be finite and non-negative. Current value: %g", T_end);   /* :335-337 */
    if (!count) FAIL(-1, "null count");
    return sample_skeleton_impl(cfg_in, C, capacity, chain_id_offset, xinit, vinit, seed, X, V, t, horizon, ar, errored_bound,
                                error_value_ar, rejected, hitting_horizon, n_threads, T_end, count);
}

static int sample_skeleton_impl(const orc_config_t *cfg_in, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                        const double *xinit, const double *vinit, uint64_t seed,
                        double *X, double *V, double *t, double *horizon, double *ar,
                        int32_t *errored_bound, double *error_value_ar, int32_t *rejected,
                        int32_t *hitting_horizon, int n_threads, double T_until, int64_t *count) {
    orc_config_t cfg = *cfg_in;
    if (orc_normalize_config(&cfg)) return -1;
    if (n_sk <= 0) FAIL(-1, "n_sk must be positive. Current value: %lld", (long long)n_sk);   /* sample.jl:262-264 */
    if (C < 0) FAIL(-1, "number of chains must be non-negative");
    const int d = cfg.dim;
    int rc = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel num_threads(n_threads)
    {
        ws_t w; int ok = ws_init(&w, d, cfg.grid_size == 0 ? 2 : cfg.grid_size) == 0;
#pragma omp for schedule(static)        /* contiguous slabs: a thread first-touches and fills its own part of the outputs */
        for (int64_t c = 0; c < C; ++c) {
            if (!ok) { rc = -1; continue; }
            size_t o = (size_t)c * (size_t)n_sk;
            int r = run_chain(&cfg, &w, n_sk, seed, (uint64_t)(chain_id_offset + c),
                              xinit + (size_t)c * d, vinit + (size_t)c * d,
                              X + o * d, V + o * d, t + o,
                              horizon ? horizon + o : NULL, ar ? ar + o : NULL,
                              errored_bound ? errored_bound + o : NULL,
                              error_value_ar ? error_value_ar + 5 * o : NULL,
                              rejected ? rejected + o : NULL, hitting_horizon ? hitting_horizon + o : NULL,
                              cfg.is_active_out ? cfg.is_active_out + o * d : NULL, T_until, count ? count + c : NULL);
            if (r) rc = r;
        }
        ws_free(&w);
    }
    if (rc == -4) FAIL(-4, "a chain hit the per-event iteration cap");
    if (rc == -6) FAIL(-6, "a random-variate tape was exhausted");
    if (rc == -7) return -7;
    if (rc) FAIL(rc, "allocation failure");
    return 0;
}

/* The sampler's deterministic flow as sample_from_skeleton and the closing point of the time-horizon variant call it
 * (sampler.flow(x0, v0, tau), sample.jl:499 / :394): plain `x + v * tau` (two roundings, as Julia evaluates it), the rotation for
 * the Boomerang, the closed form for SpeedUpZigZag (dot products in the engine's wave order), velocity-Verlet for RHMC. */
static void flow_plain(const orc_config_t *cfg, int d, const double *x0, const double *v0, double tau, double *xo, double *vo) {
    if (cfg->sampler == ORC_RHMC) {
        double *xb = malloc((size_t)d * 8 * 5), *vb = xb + d, *xe = vb + d, *g = xe + d, *ge = g + d;
        for (int q = 0; q < d; ++q) { xb[q] = x0[q]; vb[q] = v0[q]; }
        rhmc_flow(cfg, xb, vb, d, tau, xe, g, ge);
        for (int q = 0; q < d; ++q) { xo[q] = xb[q]; if (vo) vo[q] = vb[q]; }
        free(xb);
    } else if (cfg->sampler == ORC_SUZZ) {          /* SpeedUpZigZagSamplers.jl:71-79 */
        double *yb = malloc((size_t)d * 8), *pb = malloc((size_t)d * 8);
        const double av = v0[0] * x0[0], dd = (double)d;
        for (int q = 0; q < d; ++q) { yb[q] = x0[q] - av * v0[q]; pb[q] = yb[q] * v0[q]; }
        const double yv = sum_wave64(pb, d);
        for (int q = 0; q < d; ++q) pb[q] = yb[q] * yb[q];
        const double yy = sum_wave64(pb, d);
        const double c = v0[0] * yv, a = (1.0 + yy) / dd - (c * c) / (dd * dd), Y0 = x0[0] + c / dd;
        const double bt = (Y0 + sqrt(Y0 * Y0 + a)) * orc_exp(sqrt(dd) * v0[0] * tau);
        const double X1 = (bt * bt - a) / (2.0 * bt) - c / dd;
        const double kv = v0[0] * X1;
        for (int q = 0; q < d; ++q) { const double vq = v0[q]; xo[q] = yb[q] + kv * vq; if (vo) vo[q] = vq; }
        free(yb); free(pb);
    } else if (cfg->sampler == ORC_BOOMERANG) {
        double s, cs; orc_sincos(tau, &s, &cs);
        for (int q = 0; q < d; ++q) {
            const double xq = x0[q], vq = v0[q];
            xo[q] = xq * cs + vq * s;
            if (vo) vo[q] = (-xq) * s + vq * cs;
        }
    } else {
        for (int q = 0; q < d; ++q) { const double xq = x0[q], vq = v0[q]; xo[q] = xq + vq * tau; if (vo) vo[q] = vq; }
    }
}

/* The three sample_from_skeleton methods of the reference, per chain:
 *   method 0  N::Int                (sample.jl:475-513; StickyPDMP :516-561): N samples, dt = t_end / N
 *   method 1  dt::Float64           (:573-646; StickyPDMP: use_active):        samples at j dt, j = 1..floor(t_end / dt)
 *   method 2  (N::Int, dt::Float64) (:649-682): history columns 1..n_prefix, tm = dt:dt:t[n_prefix], previous index
 *             searchsortedfirst(t, tm) - 1 clamped to 1, flow with v0 .* active for EVERY sampler
 * is_active: d x n_sk x C bytes or NULL (all active); consulted by the StickyPDMP methods and by method 2.
 * lens: per-chain number of history columns (ragged histories of the time-horizon variant) or NULL (n_sk).
 * N: samples per chain (methods 1/2 with counts: the capacity; counts[c] = min(N, floor(t_end_c / dt)) are written). */
int orc_sample_from_skeleton_ex(const orc_config_t *cfg, int method, int64_t C, int64_t n_sk, const double *X, const double *V,
                                const double *t, const uint8_t *is_active, const int64_t *lens, int64_t n_prefix, int64_t N,
                                double dt_in, int discard_vt, double *out, int64_t *counts, int n_threads) {
    if (N <= 0) FAIL(-1, "N must be positive. Current value: %lld", (long long)N);           /* :476-478 */
    if (n_sk <= 0) FAIL(-1, "empty history");
    if (method != 0 && !(dt_in > 0.0)) FAIL(-1, "dt must be positive");
    if (method == 2 && (n_prefix <= 0 || n_prefix > n_sk)) FAIL(-1, "N out of range of the history");
    if (cfg->sampler == ORC_STICKY && !is_active) FAIL(-1, "StickyPDMP methods need history.is_active");
    const int d = cfg->dim;
    const size_t rows = discard_vt ? (size_t)d : 2 * (size_t)d + 1;                           /* :487 */
    const int ia_mode = !is_active ? 0 : (method == 2 ? 2 : (cfg->sampler == ORC_STICKY ? 1 : 0));
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#endif
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int64_t c = 0; c < C; ++c) {
        g_ver = (cfg->rng_version == 1) ? 1 : 2;
        const double *Xc = X + (size_t)c * n_sk * d, *Vc = V + (size_t)c * n_sk * d, *tc = t + (size_t)c * n_sk;
        const uint8_t *Ac = is_active ? is_active + (size_t)c * n_sk * d : NULL;
        double *oc = out + (size_t)c * N * rows;
        double *vu = malloc((size_t)d * 8 * 2), *vn = vu + d;
        int64_t nh = (method == 2) ? n_prefix : n_sk;                                        /* views 1:N, :650-653 */
        if (lens && lens[c] > 0 && lens[c] < nh) nh = lens[c];
        const double dt = method != 0 ? dt_in : tc[nh - 1] / (double)N;                      /* :483-484 */
        int64_t nmax = N;
        if (method != 0 && counts) {
            const double q = floor(tc[nh - 1] / dt_in);                                      /* n_skel = Int(floor(T_end / dt)), :608 */
            nmax = q < (double)N ? (int64_t)q : N;
            counts[c] = nmax;
        }
        int64_t i = 0;
        for (int64_t j = 1; j <= nmax; ++j) {
            const double tm = (double)j * dt;                                                /* :492 */
            if (method == 2) { while (i < nh - 1 && tc[i + 1] < tm) ++i; }                   /* searchsortedfirst(t, tm) - 1, :657 (i >= 1, :663) */
            else { while (i < nh - 1 && tc[i + 1] <= tm) ++i; }                              /* :493-495 */
            const double tau = tm - tc[i];                                                   /* :497 */
            const double *x0 = Xc + (size_t)i * d, *v0 = Vc + (size_t)i * d;
            double *o = oc + (size_t)(j - 1) * rows;
            if (ia_mode) {
                const uint8_t *a0 = Ac + (size_t)i * d;
                for (int q = 0; q < d; ++q)
                    vu[q] = (ia_mode == 2) ? v0[q] * (a0[q] ? 1.0 : 0.0) : (a0[q] ? v0[q] : 0.0);   /* :667 / :547-549 */
                v0 = vu;
            }
            flow_plain(cfg, d, x0, v0, tau, o, discard_vt ? vn : o + d);
            if (!discard_vt) o[2 * d] = tm;                                                  /* :505-508 */
        }
        free(vu);
    }
    return 0;
}

/* sample_from_skeleton(sampler, N, history), sample.jl:475-513 */
int orc_sample_from_skeleton(const orc_config_t *cfg, int64_t C, int64_t n_sk, const double *X,
                             const double *V, const double *t, int64_t N, int discard_vt, double *out,
                             int n_threads) {
    return orc_sample_from_skeleton_ex(cfg, 0, C, n_sk, X, V, t, NULL, NULL, 0, N, 0.0, discard_vt, out, NULL, n_threads);
}

/* dt_in > 0: the `dt::Float64` method (sample.jl:573-646): samples at j*dt_in, j = 1..N (the caller passes
 * N = floor(t[end]/dt), :608); dt_in <= 0: the `N::Int` method (:475-513), dt = t[end]/N. */
int orc_sample_from_skeleton_dt(const orc_config_t *cfg, int64_t C, int64_t n_sk, const double *X,
                                const double *V, const double *t, int64_t N, double dt_in, int discard_vt, double *out,
                                int n_threads) {
    return orc_sample_from_skeleton_ex(cfg, dt_in > 0.0 ? 1 : 0, C, n_sk, X, V, t, NULL, NULL, 0, N, dt_in > 0.0 ? dt_in : 0.0, discard_vt,
                                       out, NULL, n_threads);
}
