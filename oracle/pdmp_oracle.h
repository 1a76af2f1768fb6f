/*
 * pdmp_oracle.h -- CPU restatement of PDMPFlux.jl's sample_skeleton()/sample_from_skeleton()
 * hot path.  TEST INFRASTRUCTURE ONLY: this is the checker, never the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference (pure Julia, /root/reference) cannot be executed in this
 * pipeline (no Julia toolchain) and holds no golden vectors for this path.  The restatement is
 * pinned by the reference's invariant / statistical tests, by hand-derived known answers, and by
 * Philox/libm known-answer vectors for the (build-defined) random stream.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef PDMP_ORACLE_H
#define PDMP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sampler ids -- src/Samplers/{ZigZag,BouncyParticle,Boomerang}Samplers.jl */
enum { ORC_ZIGZAG = 0, ORC_BPS = 1, ORC_BOOMERANG = 2,
       ORC_FECMC = 3, /* ForwardECMC, src/Samplers/ForwardEventChainMonteCarlo.jl (default velocity jump, :127-175) */
       ORC_SUZZ = 4,  /* SpeedUpZigZag, src/Samplers/SpeedUpZigZagSamplers.jl: closed-form non-linear flow (:71-79), effective gradient (:81-83) */
       ORC_RHMC = 5,  /* RHMC, src/Samplers/RandomizedHamiltonianMonteCarlo.jl: velocity-Verlet flow (:97-130), constant refresh clock, Horowitz refreshment (:144-149) */
       ORC_STICKY = 6 /* StickyZigZag, src/Samplers/StickyZigZagSamplers.jl + src/StickySamplingLoop.jl: coordinates stick at 0 and thaw at rates kappa;
                       * kappa[d] = the LAST d doubles of target_params */ };
/* built-in potentials (closures cannot cross a C ABI) */
enum { ORC_T_GAUSS_ISO = 0, ORC_T_GAUSS_DIAG = 1, ORC_T_BANANA = 2, ORC_T_GAUSS_DENSE = 3,
       /* grad-U by forward-mode differentiation of a scalar potential (ADBackend.jl:95-109), ad_potentials.c:
        * the reference's test potentials U_banana, U_funnel, U_ridged_gauss (test/test_config.jl:33-46) */
       ORC_T_AD_BANANA = 16, ORC_T_AD_FUNNEL = 17, ORC_T_AD_RIDGED = 18 };
/* time-derivative of the bounded rate: UpperBound.jl:209-210 (ForwardDiff) / :223-224 (FiniteDiff) */
enum { ORC_DERIV_FORWARD = 0, ORC_DERIV_FINITE_DIFF = 1 };
/* floating-point summation order:
 *   REFERENCE: plain left-to-right sums over coordinates, per-dimension cumsum before the sum over
 *              dimensions (UpperBound.jl:243-246), sequential Categorical walk.
 *   WAVE64   : the order the MI355X engine uses (64-lane ascending xor-butterfly, Kogge-Stone
 *              prefix scan, cumsum after the sum over dimensions).  Bitwise target for the HIP path. */
/*   GROUP    : G lanes per chain (orc_config_t.group_lanes): per-lane left-to-right partial sums over the
 *              coordinates i = g, g+G, ..., ascending xor-butterfly over the G lanes; cum_sum after the sum
 *              over dimensions (as WAVE64); Categorical walk in slots of G over the un-normalised rates
 *              against u*sum (no division per coordinate).  GROUP(1) sums like REFERENCE. */
enum { ORC_ARITH_REFERENCE = 0, ORC_ARITH_WAVE64 = 1, ORC_ARITH_GROUP = 2 };

typedef struct orc_config_t {
    int32_t sampler;
    int32_t dim;
    int32_t grid_size;          /* n >= 2; 0 (Brent constant bound, UpperBound.jl:18-36) is not restated */
    int32_t vectorized_bound;
    int32_t signed_bound;
    int32_t adaptive;
    int32_t gaussian_velocity;  /* BPS only, BouncyParticleSamplers.jl:24,68-72 */
    int32_t deriv_mode;
    int32_t target_id;
    int32_t arith;
    double  tmax;
    double  refresh_rate;
    const double *target_params; /* DIAG: mu[d], prec[d];  DENSE: P[d*d] column-major */
    int64_t n_target_params;
    int64_t max_iter_per_event;  /* 0 = default 1e6; safety cap (status -4) */
    int32_t group_lanes;         /* G for ORC_ARITH_GROUP */
    /* tape source (all three or none): per chain tape_n_* variates, chain-major (chain c at c*tape_n_*) */
    const double *tape_exp, *tape_uni, *tape_nrm;
    int64_t tape_n_exp, tape_n_uni, tape_n_nrm;
    /* ForwardECMC keywords (ForwardEventChainMonteCarlo.jl:290-292): mix_p = 0.5, ran_p = false, switch = true, positive = true */
    double  fe_mix_p;
    int32_t fe_ran_p, fe_switch, fe_positive;
    int32_t no_identities;       /* GROUP order only, checker-side switch (tests/test_oracle_identities.py): 1 = evaluate the reference's full
                                  * formulas in the GROUP summation order, WITHOUT the three algebraic identities the engine order takes
                                  * (affine cell, affine tail, quadratic form) -- what those identities are compared against */
    /* RHMC keywords (RandomizedHamiltonianMonteCarlo.jl:48-57): step_size = 0.05, phi = pi/2 (refresh_rate = 1, tmax = 10, adaptive = false) */
    double  rh_step_size, rh_phi;
    double  fe_speed_factor;     /* ForwardECMC speed_factor (ForwardEventChainMonteCarlo.jl:177-218); 0 -> 1.0 */
    uint8_t *is_active_out;      /* StickyZigZag: PDMPHistory.is_active (Composites.jl:142,256), d x n_sk x C bytes, or NULL */
    /* Julia random source (oracle/julia_rng.c): 0 = off (Philox or tapes); JL_SEED_LIMBS (1) / JL_SEED_SHA256 (2): chain c draws
     * from MersenneTwister(seed + c) -- the reference's own idiom for many chains, `seed = 42 + i`, test/test_property_based.jl:138 -- seeded as
     * Julia 1.0 - 1.10 / as Julia 1.11, through randexp / rand / randn in the reference's program order: one stream for the
     * three kinds, as in the reference.  Vectors of normals (`randn!(rng, v)`, BouncyParticleSamplers.jl:67; `randn(key, dim)`, ForwardECMC) follow the
     * MersenneTwister method of randn! -- from 13 elements on: bulk fill first, rejections drawn afterwards (julia_rng.c: jl_randn_fill) -- unless
     * + 4 (ORC_JL_SCALAR_RANDN) asks for scalar draws in index order, what an rng WRAPPED around the MersenneTwister takes (julia/dump_tape.jl's TapeRNG).
     * The Boomerang's refresh `randn(dim)` (BoomerangSamplers.jl:65) draws from Julia's GLOBAL rng, which the reference never seeds: not
     * reproducible in the reference itself, drawn from the chain's stream here (Q8).  Unverified against a running Julia. */
    int32_t julia_rng;
    int32_t pad0_;
    /* ... and, when jl_tape_exp is set, the variates every chain consumed, per kind in consumption order: chain c (global id) at c * jl_cap_*,
     * its counts at jl_count[3 c + kind] -- the tapes pdmp_set_tape replays on the engine.  A chain that overruns a capacity returns -6. */
    double *jl_tape_exp, *jl_tape_uni, *jl_tape_nrm;
    int64_t jl_cap_exp, jl_cap_uni, jl_cap_nrm;
    int64_t *jl_count;
    /* Arithmetic contract of the random stream and the elementary functions: 0 / 2 = v2 (default since round 5: Philox4x32-7,
     * fma-Horner log / exp / sin / cos, octant Box-Muller), 1 = v1 (Philox4x32-10, fdlibm forms without fma: rounds 1-4 and their
     * frozen fixtures).  pdmp_oracle.c "Arithmetic contract version"; mirrors pdmp_config_t.rng_version. */
    int32_t rng_version;
    int32_t pad1_;
} orc_config_t;

const char *orc_last_error(void);

/* Normalises cfg the way the reference constructors do (ZigZagSamplers.jl:62-78,
 * BouncyParticleSamplers.jl:26-37, BoomerangSamplers.jl:25-36).  Returns 0 or -1. */
int orc_normalize_config(orc_config_t *cfg);

/* Ensemble of C independent chains; chain c uses the Philox key (seed, chain_id_offset + c).
 * Layouts (column-major, as PDMPHistory, Composites.jl:138-149,239-260):
 *   xinit,vinit: d x C;  X,V: d x n_sk x C;  t,horizon,ar: n_sk x C;  error_value_ar: 5 x n_sk x C;
 *   errored_bound,rejected,hitting_horizon: int32 n_sk x C.   Diagnostic pointers may be NULL.
 * n_threads <= 0 -> all OpenMP threads.   Returns 0, -1 (argument), -4 (iteration cap). */
int orc_sample_skeleton(const orc_config_t *cfg, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                        const double *xinit, const double *vinit, uint64_t seed,
                        double *X, double *V, double *t,
                        double *horizon, double *ar, int32_t *errored_bound, double *error_value_ar,
                        int32_t *rejected, int32_t *hitting_horizon, int n_threads);

/* sample.jl:475-513, per chain; out: d x N x C (or (2d+1) x N x C if !discard_vt). */
int orc_sample_from_skeleton(const orc_config_t *cfg, int64_t C, int64_t n_sk,
                             const double *X, const double *V, const double *t,
                             int64_t N, int discard_vt, double *out, int n_threads);

int orc_sample_from_skeleton_dt(const orc_config_t *cfg, int64_t C, int64_t n_sk,
                                const double *X, const double *V, const double *t,
                                int64_t N, double dt, int discard_vt, double *out, int n_threads);

/* All three sample_from_skeleton methods (sample.jl:475-513 / :516-561, :573-646, :649-682) with PDMPHistory.is_active and
 * ragged history lengths; see pdmp_oracle.c. */
int orc_sample_from_skeleton_ex(const orc_config_t *cfg, int method, int64_t C, int64_t n_sk, const double *X, const double *V,
                                const double *t, const uint8_t *is_active, const int64_t *lens, int64_t n_prefix, int64_t N,
                                double dt, int discard_vt, double *out, int64_t *counts, int n_threads);

/* sample_skeleton(sampler, T::Float64, xinit, vinit) (sample.jl:323-439) with `capacity` history columns per chain:
 * count[c] = columns of chain c (events with t <= T + the closing column at t = T), or -(columns so far) - 1 when the capacity
 * ran out before T (the caller doubles it, as _grow_history does, Composites.jl:172-191). */
int orc_sample_skeleton_until(const orc_config_t *cfg, int64_t C, int64_t capacity, int64_t chain_id_offset,
                              const double *xinit, const double *vinit, uint64_t seed, double T_end,
                              double *X, double *V, double *t, double *horizon, double *ar,
                              int32_t *errored_bound, double *error_value_ar, int32_t *rejected,
                              int32_t *hitting_horizon, int64_t *count, int n_threads);

/* --- pieces exposed for known-answer tests --- */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void   orc_philox4x32_r(int rounds, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* the arithmetic contract (1 / 2) the cfg-less functions below use on the calling thread: orc_log .. orc_init_states */
void   orc_set_arith_version(int v);
int    orc_get_arith_version(void);
double orc_log(double x);
double orc_exp(double x);
void   orc_sincos(double x, double *s, double *c);
void   orc_sincos_2pi(double u, double *s, double *c);
/* first n_draw variates of chain stream (seed, chain): kind 0 uniform, 1 exp, 2 normal-vector (d=n_draw) */
void   orc_rng_dump(uint64_t seed, int64_t chain, int kind, int64_t n_draw, double *out);
void   orc_julia_draws(int64_t seed, int seeding, int kind, int64_t n, double *out);
void   orc_julia_randn_fill(int64_t seed, int seeding, int n, int scalar, double *out);   /* randn!(MersenneTwister(seed), zeros(n)): bulk method (n >= 13) or scalar draws */   /* rand / randexp / randn of the restated MersenneTwister(seed) */
/* Synthetic initial states from stream 1 of each chain's key; mirrors pdmp_init_states_device. */
enum { ORC_V_RADEMACHER = 0, ORC_V_GAUSSIAN = 1, ORC_V_SPHERE = 2 };
void   orc_init_states(uint64_t seed, int64_t chain0, int64_t C, int d, double x_mean, double x_scale, int v_kind, double *x, double *v);
/* one bound + inversion: fills grid[n], box_max[n-1], cum_sum[n]; returns step_size (UpperBound.jl) */
double orc_upper_bound(const orc_config_t *cfg, const double *x, const double *v, double horizon,
                       double *grid, double *box_max, double *cum_sum);
void   orc_next_event(int n, const double *grid, const double *box_max, const double *cum_sum,
                      double step_size, double exp_rv, double *tp, double *lambda_bar);
/* gradient of the built-in potential (the user's grad-U closure) */
void   orc_grad(const orc_config_t *cfg, const double *x, double *g);
double orc_potential(const orc_config_t *cfg, const double *x);
/* ad_potentials.c: grad-U of ORC_T_AD_* by forward mode, value gv and directional derivative ge along xe */
void   orc_ad_grad_dual(int target, int d, const double *xv, const double *xe, double *gv, double *ge);
double orc_ad_potential(int target, int d, const double *x);
int    orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
