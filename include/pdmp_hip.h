/*
 * pdmp_hip.h -- C ABI of libpdmphip.so, the MI355X (gfx950) engine behind PDMPFlux.jl's
 * sample_skeleton() / sample_from_skeleton() / sample() for ensembles of independent chains.
 *
 * PDMPFlux.jl has no FFI seam of its own (pure Julia, multiple dispatch on AbstractPDMP,
 * src/Samplers/AbstractPDMP.jl:6); these entry points are what a `ccall` shim bound to the exported
 * functions of src/PDMPFlux.jl:26 needs.  Each declaration names the reference interface it replaces
 * (file:line under the reference tree).  Plain pointers and sizes only; no torch / HIP types.
 *
 * Conventions
 *   - All arrays are column-major exactly as the reference's PDMPHistory (src/Composites.jl:138-149):
 *       xinit, vinit : d x C                       (chain c = column c)
 *       X, V         : d x n_sk x C                (event k of chain c at offset ((c*n_sk)+k)*d)
 *       t, horizon, ar            : n_sk x C       (double)
 *       errored_bound, rejected, hitting_horizon : n_sk x C (int32, Composites.jl:145-148)
 *       error_value_ar            : 5 x n_sk x C
 *   - Chain c draws from the Philox4x32 stream (7 rounds; 10 with rng_version = 1) keyed (seed, chain_id_offset + c): results do not
 *     depend on how chains are sharded over GPUs (mirrors `seed = 42 + i`, test/test_property_based.jl:138).
 *   - Return 0 on success; <0 on error with a message in pdmp_last_error():
 *       PDMP_ERR_ARGUMENT  (-1)  -> Julia ArgumentError   (sample.jl:262-264,476-478; ZigZagSamplers.jl:62-68)
 *       PDMP_ERR_DIMENSION (-2)  -> Julia DimensionMismatch (AbstractPDMP.jl:96-98)
 *       PDMP_ERR_RUNTIME   (-3)  HIP runtime failure / no device
 *       PDMP_ERR_ITERCAP   (-4)  a chain exceeded max_iter_per_event thinning steps for one event
 *       PDMP_ERR_UNSUPPORTED (-5) configuration outside what the engine implements
 *   - Inputs are never mutated (deliberate fix of the aliasing quirk, AbstractPDMP.jl:141-149).
 *   - A handle is not re-entrant; distinct handles may be used from distinct host threads.
 */
#ifndef PDMP_HIP_H
#define PDMP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDMP_ABI_VERSION 11   /* 11: + pdmp_config_t.rng_version (arithmetic contract v2 the default), grid_size 2..64 on the group kernel for every sampler / potential; 10: + pdmp_device_alloc_streams (the streamed output arrays drawn several times, the fastest set kept); 9: + pdmp_device_alloc / pdmp_device_free (output arrays from independent physical chunks), pdmp_init_states_device, pdmp_multi_sample_skeleton_device / pdmp_multi_init_states, pdmp_reload_policy (launch policy fixed per handle); 8: + the time horizon as a kernel feature (pdmp_sample_skeleton_until_device, pdmp_close_chunk_device, pdmp_until_advance_device, pdmp_compact_chunk_device); tape replay on every kernel form; 7: + pdmp_last_launch_form; 6: is_active / ragged lengths in sample_from_skeleton*, moments and close_at; + the (N, dt) method, pdmp_compact_*, pdmp_multi_*; 5: + PDMP_STICKY, pdmp_diag_t.is_active; 2: + pdmp_create_from_source, PDMP_T_USER; 3: + PDMP_FECMC and its keywords at the end of pdmp_config_t; 4: + PDMP_SUZZ, PDMP_RHMC and its keywords */

enum { PDMP_OK = 0, PDMP_ERR_ARGUMENT = -1, PDMP_ERR_DIMENSION = -2, PDMP_ERR_RUNTIME = -3,
       PDMP_ERR_ITERCAP = -4, PDMP_ERR_UNSUPPORTED = -5 };

/* src/Samplers/ZigZagSamplers.jl:35-116, BouncyParticleSamplers.jl:1-84, BoomerangSamplers.jl:1-77,
 * ForwardEventChainMonteCarlo.jl:262-344 (ForwardECMC with its default velocity jump, :127-175) */
enum { PDMP_ZIGZAG = 0, PDMP_BPS = 1, PDMP_BOOMERANG = 2, PDMP_FECMC = 3,
       PDMP_SUZZ = 4, /* SpeedUpZigZag, src/Samplers/SpeedUpZigZagSamplers.jl */
       PDMP_RHMC = 5, /* RHMC, src/Samplers/RandomizedHamiltonianMonteCarlo.jl (velocity-Verlet flow, constant refresh clock) */
       PDMP_STICKY = 6 /* StickyZigZag, src/Samplers/StickyZigZagSamplers.jl + src/StickySamplingLoop.jl; kappa[dim] = the LAST dim doubles of target_params */ };
/* Potential descriptors replacing the Julia closure field `grad-U` (ZigZagSamplers.jl:37): */
enum { PDMP_T_GAUSS_ISO = 0,   /* U = |x|^2/2                         (README.md:36-38)            */
       PDMP_T_GAUSS_DIAG = 1,  /* U = sum prec_i (x_i-mu_i)^2/2 ; params = mu[d], prec[d]            */
       PDMP_T_BANANA = 2,      /* U_banana                           (test/test_config.jl:33-36)   */
       PDMP_T_GAUSS_DENSE = 3, /* U = x'Px/2 ; params = P[d*d] column-major (Zig-Zag, BPS; Boomerang at dim <= 1024: batched form only) */
       PDMP_T_USER = 4         /* potential given as source: pdmp_create_from_source only              */ };
/* AD_backend for the time-derivative of the rate (UpperBound.jl:209-224, AbstractPDMP.jl:18-28) */
enum { PDMP_DERIV_FORWARD = 0 /* "ForwardDiff": exact forward-mode duals */,
       PDMP_DERIV_FINITE_DIFF = 1 /* "FiniteDiff": UpperBound.jl:50-76 */ };

/* Keyword arguments of the reference constructors as a plain struct. */
typedef struct pdmp_config_t {
    int32_t sampler;            /* PDMP_ZIGZAG | PDMP_BPS | PDMP_BOOMERANG | PDMP_FECMC | PDMP_SUZZ | PDMP_RHMC | PDMP_STICKY */
    int32_t dim;                /* > 0 */
    int32_t grid_size;          /* 2..64 grid bound; 0 = Brent constant bound (UpperBound.jl:18-36) */
    int32_t vectorized_bound;   /* ZigZag default 1; forced 0 for BPS/Boomerang */
    int32_t signed_bound;       /* default 1 */
    int32_t adaptive;           /* default 1 */
    int32_t gaussian_velocity;  /* BPS(Gaussian_velocity=...) */
    int32_t deriv_mode;         /* PDMP_DERIV_* */
    int32_t target_id;          /* PDMP_T_* */
    int32_t device;             /* HIP device ordinal; -1 = current device */
    double  tmax;               /* 0 -> 1.0 and adaptive=1 (ZigZagSamplers.jl:73) */
    double  refresh_rate;
    const double *target_params;/* host pointer, copied by pdmp_create */
    int64_t n_target_params;
    int64_t max_iter_per_event; /* 0 -> 1e6 */
    /* ForwardECMC(dim, grad-U; ran_p = false, mix_p = 0.5, switch = true, positive = true) (ForwardEventChainMonteCarlo.jl:290-292);
     * ignored by the other samplers.  speed_factor: fe_speed_factor below; normal = true is not implemented (the reference's own
     * variant takes sqrt of a vector, :228, and cannot run). */
    double  fe_mix_p;
    int32_t fe_ran_p, fe_switch, fe_positive, fe_reserved;
    /* RHMC(dim, grad-U; refresh_rate = 1.0 (or 1 / mean_duration), phi = pi/2, step_size = 0.05, tmax = 10.0, adaptive = false)
     * (RandomizedHamiltonianMonteCarlo.jl:48-57); ignored by the other samplers. */
    double  rh_step_size, rh_phi;
    double  fe_speed_factor;    /* ForwardECMC(...; speed_factor = 1.0) (:292, jump variant :177-218); 0 -> 1.0 */
    /* Arithmetic contract of the build-defined random stream (the engine's stand-in for MersenneTwister / randexp / randn, AbstractPDMP.jl:100)
     * and of the elementary functions log / exp / sin / cos.  0 = the default = 2 (round 5: Philox4x32-7, Horner-on-fma polynomials, octant
     * Box-Muller -- every kernel form); 1 = the contract of rounds 1-4 (Philox4x32-10, fdlibm forms without fma), kept for the fixtures frozen
     * then: served by the LDS-form group kernels and the wave-per-chain kernel only (no register / lane / batched dense form, no tape). */
    int32_t rng_version;
    int32_t reserved0;
} pdmp_config_t;

typedef struct pdmp_engine *pdmp_handle;

/* Optional per-event diagnostics of PDMPHistory (Composites.jl:143-148); any pointer may be NULL. */
typedef struct pdmp_diag_t {
    double  *horizon;
    double  *ar;
    int32_t *errored_bound;
    double  *error_value_ar;
    int32_t *rejected;
    int32_t *hitting_horizon;
    uint8_t *is_active;         /* StickyZigZag: PDMPHistory.is_active (Composites.jl:142,256), d x n_sk x C bytes; NULL otherwise */
} pdmp_diag_t;

typedef struct pdmp_timing_t {
    double  kernel_ms_total;    /* sum of hipEvent-timed kernel durations since the last reset */
    double  kernel_ms_last;
    int64_t launches;
} pdmp_timing_t;

int         pdmp_abi_version(void);
const char *pdmp_last_error(void);       /* thread-local */
int         pdmp_device_count(void);     /* hipGetDeviceCount; 0 when no GPU is visible */

/* Sampler construction: ZigZag/ZigZagAD (ZigZagSamplers.jl:58-126), BPS/BPSAD
 * (BouncyParticleSamplers.jl:21-94), Boomerang/BoomerangAD (BoomerangSamplers.jl:21-87).
 * Applies the constructors' normalisations; the normalised values can be read back. */
int pdmp_create(const pdmp_config_t *cfg, pdmp_handle *out);
/* The `U::Function` argument of the *AD constructors and create_gradient_function (src/ADBackend.jl:30-142), which
 * accepts a scalar potential (differentiated by ForwardDiff, :95-109) or an already-differentiated gradient (:55-66).
 * A Julia closure cannot cross a C ABI; the engine takes the function as a few lines of HIP source instead, compiled at
 * run time (hiprtc) into the same kernel as the built-in potentials:
 *   PDMP_SRC_POTENTIAL:  template <class X> __device__ auto U(const X &x, int d)              x[j] = coordinate j (0-based)
 *   PDMP_SRC_GRADIENT:   template <class X> __device__ auto grad_i(int i, const X &x, int d)   coordinate i of grad-U
 * written with + - * /, sq (= x^2), sqrt_, log_, exp_, sin_, cos_ so that it runs on doubles, on forward-mode numbers in time
 * (the slope of the rate along the flow, UpperBound.jl:102,210) and -- for a potential -- on the nested forward-mode
 * numbers that yield grad-U (one partial per pass; pdmpflux.jl_amd/csrc/user_ad.hpp states every rule).  cfg->target_id
 * and target_params are ignored.  A source that does not compile returns PDMP_ERR_ARGUMENT with the compiler's log in
 * pdmp_last_error().  grid_size: 2..64, or 0 (the Brent bound). */
enum { PDMP_SRC_POTENTIAL = 0, PDMP_SRC_GRADIENT = 1 };
int pdmp_create_from_source(const pdmp_config_t *cfg, const char *source, int32_t kind, pdmp_handle *out);
int pdmp_get_config(pdmp_handle h, pdmp_config_t *out);
int pdmp_destroy(pdmp_handle h);
/* Arithmetic layout the engine chose for this configuration: lanes per chain G of the group kernel
 * (sums run in the "GROUP(G)" order of DESIGN.md; G = 1 is the reference's left-to-right order), or
 * 0 when the wave-per-chain kernel ("WAVE64" order) serves it.  Lets a checker pick the matching order. */
int pdmp_get_group_lanes(pdmp_handle h);
/* Lanes per chain of the following launches.  pdmp_suggest_group_lanes(h, C) is the layout the engine recommends for launches of
 * C chains: the default G (the smallest with ceil(dim/G) <= 13 coordinate slots per lane) for every C -- measured on MI355X, more
 * lanes per chain do not speed up launches that are too small to fill the GPU (such a launch is bound by the latency
 * of one wave's sequential event loop, which is per-chain scalar work).  pdmp_set_group_lanes pins another power of two (0 = back to
 * the default).  G fixes the order of the sums over coordinates ("GROUP(G)"): results are bit-identical across shardings for equal
 * G, which is why a sharded run (pdmp_multi_*, the Python driver) uses ONE G for all shards. */
int pdmp_suggest_group_lanes(pdmp_handle h, int64_t C);
int pdmp_set_group_lanes(pdmp_handle h, int32_t G);
/* Which kernel form serves a launch (results never depend on it) and the forms' tuning numbers are a per-handle POLICY, fixed when the
 * handle is created: defaults measured on MI355X, overridden by the PDMP_* environment variables of README.md (test hooks, tuning
 * scripts).  No launch reads the environment; a caller that changes it between launches of one handle calls this to re-read it.
 * (Four process-wide tuning knobs are outside the policy and read ONCE per process, on first use: PDMP_DGEMM_POLITE, PDMP_DENSE_STEP_BLOCKS,
 * PDMP_DGEMM_BM -- A/B knobs of the batched dense forms -- and PDMP_ALLOC_POOL.) */
int pdmp_reload_policy(pdmp_handle h);
/* Which kernel form served the handle's last pdmp_sample_skeleton* launch (waits for it to finish): a diagnostic for benchmarks and
 * tests, results do not depend on it.  PDMP_FORM_LANE: one lane per chain (csrc/lane_kernel.hpp; large Zig-Zag ensembles at
 * dim 27..52 with unit speeds on the isotropic Gaussian / banana -- bit-identical to the group kernel with G = 4). */
enum { PDMP_FORM_NONE = 0, PDMP_FORM_WAVE = 1, PDMP_FORM_GROUP = 2, PDMP_FORM_GROUP_REG = 3, PDMP_FORM_LANE = 4, PDMP_FORM_DENSE = 5, PDMP_FORM_USER = 6,
       PDMP_FORM_LANE_PAIR = 7 /* the lane kernel with a PAIR of lanes per chain: dim 53..104 (bit-identical to the group kernel with G = 8), and mid-sized launches at dim 25..52 (G = 4) */ };
int pdmp_last_launch_form(pdmp_handle h);

/* Tape source: replace the built-in Philox stream (the engine's stand-in for Julia's
 * MersenneTwister, AbstractPDMP.jl:100) by pre-recorded variates, one tape per kind in consumption order --
 * randexp (SamplingLoopInplace.jl:70,136,190), rand (:154 and the samplers' jumps), randn (BouncyParticleSamplers.jl:67)
 * -- so that a run of ANY generator (e.g. a dump from Julia) can be replayed draw for draw.  HOST pointers, C x n_*
 * doubles, chain-major, indexed by the global chain id; copied.  While a tape is set, seeds are ignored.  The replay runs on the
 * kernel form the same launch uses with the Philox stream -- LDS-form / register-form group kernel, one lane per chain (dim 49..52),
 * every sampler on the built-in separable / banana targets, as separate instantiations of the same code with the tapes as random
 * source (csrc/inst_tape_*.hip), in ONE segment per chain; pdmp_get_group_lanes() keeps returning the layout, i.e. the order of
 * the sums.  Grid sizes other than 10 / 0 (Zig-Zag, BPS, Boomerang): the wave-per-chain kernel ("WAVE64" order,
 * pdmp_get_group_lanes() = 0).  An exhausted tape fails the call (ForwardECMC's randn(key, 2, dim) is logged column-major). */
int pdmp_set_tape(pdmp_handle h, int64_t C, int64_t n_exp, int64_t n_uni, int64_t n_nrm,
                  const double *exp_tape, const double *uni_tape, const double *nrm_tape);
int pdmp_clear_tape(pdmp_handle h);

/* sample_skeleton(sampler, n_sk, xinit, vinit; seed)  (src/sample.jl:253-284) for C chains:
 * init_state (AbstractPDMP.jl:93-153) + (n_sk-1) x get_event_state! (SamplingLoopInplace.jl:27-39)
 * + record! (Composites.jl:239-260).  HOST pointers; blocking. */
int pdmp_sample_skeleton(pdmp_handle h, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                         const double *xinit, const double *vinit, uint64_t seed,
                         double *X, double *V, double *t, const pdmp_diag_t *diag);

/* Same with DEVICE pointers (HBM-resident inputs and outputs), enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = the null stream).  Asynchronous: chain status is checked by
 * pdmp_check_status().  This is the entry the multi-GPU driver and bench.py use. */
int pdmp_sample_skeleton_device(pdmp_handle h, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                                const double *xinit, const double *vinit, uint64_t seed,
                                double *X, double *V, double *t, const pdmp_diag_t *diag, void *stream);

/* sample_from_skeleton(sampler, N::Int, history; discard_vt)  (src/sample.jl:475-513; the StickyPDMP method :516-561) for C chains.
 * out: rows x N x C with rows = d (discard_vt) or 2d+1.
 * is_active: PDMPHistory.is_active, d x n_sk x C bytes (pdmp_diag_t.is_active of the run).  Required for PDMP_STICKY handles,
 *   whose method flows with v_used = active ? v : 0 (:547-549) -- frozen coordinates stay exactly 0; ignored (may be NULL)
 *   for the other samplers, as in the reference's generic method.
 * lens (device forms; may be NULL): lens[c] = number of history columns of chain c (<= n_sk) for the ragged histories of the
 *   time-horizon variant (count[] of pdmp_close_at_time_device); t_end = t[lens[c]-1]. */
int pdmp_sample_from_skeleton(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                              const double *t, const uint8_t *is_active, int64_t N, int32_t discard_vt, double *out);
int pdmp_sample_from_skeleton_device(pdmp_handle h, int64_t C, int64_t n_sk, const double *X,
                                     const double *V, const double *t, const uint8_t *is_active, const int64_t *lens,
                                     int64_t N, int32_t discard_vt, double *out, void *stream);

/* sample_from_skeleton(sampler, dt::Float64, history; discard_vt) (src/sample.jl:573-646; StickyPDMP: use_active, :588-596):
 * samples at j*dt, j = 1..floor(t_end/dt) (:608).  out is rows x N x C; with counts == NULL the caller passes N = floor(t_end/dt)
 * (one chain, or chains known to share it); with counts != NULL, N is the capacity per chain, counts[c] = min(N, floor(t_end_c/dt))
 * columns of chain c are written and the rest is left untouched (host form: zero). */
int pdmp_sample_from_skeleton_dt(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                                 const double *t, const uint8_t *is_active, int64_t N, double dt, int32_t discard_vt,
                                 double *out, int64_t *counts);
int pdmp_sample_from_skeleton_dt_device(pdmp_handle h, int64_t C, int64_t n_sk, const double *X,
                                        const double *V, const double *t, const uint8_t *is_active, const int64_t *lens,
                                        int64_t N, double dt, int32_t discard_vt, double *out, int64_t *counts, void *stream);

/* sample_from_skeleton(sampler, N::Int, dt::Float64, history; discard_vt) (src/sample.jl:649-682): the first n_prefix columns
 * of the history (:650-653), sample times tm = dt:dt:t[n_prefix], previous index searchsortedfirst(t, tm) - 1 (the last column
 * with t < tm, :657, clamped to the first), flow with v .* is_active for EVERY sampler (:667; is_active NULL = all active).
 * out: rows x M x C; counts[c] (may be NULL) = min(M, floor(t[n_prefix-1]/dt)) columns written for chain c.  tm[j] = j*dt
 * (Julia's range arithmetic is TwicePrecision: same values, the length may differ when t_end/dt is within rounding of an integer). */
int pdmp_sample_from_skeleton_prefix(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                                     const double *t, const uint8_t *is_active, int64_t n_prefix, double dt, int64_t M,
                                     int32_t discard_vt, double *out, int64_t *counts);
int pdmp_sample_from_skeleton_prefix_device(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                                            const double *t, const uint8_t *is_active, int64_t n_prefix, double dt, int64_t M,
                                            int32_t discard_vt, double *out, int64_t *counts, void *stream);

/* The step right after the path: per-chain sums of x and x^2 over the N samples that
 * sample_from_skeleton(sampler, N, history) (src/sample.jl:475-513 / :516-561) would return, without writing them:
 * sum_x, sum_x2 are d x C (DEVICE pointers).  mean = sum_x / N, var = sum_x2 / N - mean^2 per chain, or pooled.
 * is_active / lens as above (is_active required for PDMP_STICKY). */
int pdmp_sample_moments_device(pdmp_handle h, int64_t C, int64_t n_sk, const double *X, const double *V,
                               const double *t, const uint8_t *is_active, const int64_t *lens, int64_t N,
                               double *sum_x, double *sum_x2, void *stream);

/* The cut of the time-horizon variant sample_skeleton(sampler, T::Float64, xinit, vinit) (src/sample.jl:323-439) on a
 * fixed-count skeleton held in HBM (DEVICE pointers; the first events of both variants coincide for one seed): count[c] =
 * number of history columns of chain c -- the events with t <= T (:372) plus, unless the last of them sits at exactly t = T,
 * one closing column at t = T written in place over the overshooting event by flow from the last event before T (:385-420;
 * StickyPDMP: v_used = active_prev ? v_prev : 0 and is_active = active_prev, :386-401, which needs diag->is_active).  With
 * diag != NULL the closing column's statistics are those of a non-event point: ar, errored_bound, error_value_ar, rejected,
 * hitting_horizon = 0 (:402-408); horizon keeps the overshooting state's value (the reference records the same state object).
 * count[c] < 0: the chain has not passed T within n_sk events (-count[c]-1 events so far); rerun with a larger n_sk
 * (_grow_history doubles, Composites.jl:172-191). */
int pdmp_close_at_time_device(pdmp_handle h, int64_t C, int64_t n_sk, double *X, double *V, double *t, const pdmp_diag_t *diag,
                              double T, int64_t *count, void *stream);

/* ---- sample_skeleton(sampler, T::Float64, xinit, vinit; seed, init_capacity) (src/sample.jl:323-439) as a kernel feature.
 * The reference grows ONE chain's history in place (`while state.t < T`, :372; _grow_history doubles the capacity, :375-377,
 * Composites.jl:172-191).  For an ensemble the history is produced in CHUNKS: a chunk is one launch over the chains that have not
 * reached T yet, `capacity` columns each; a chain stops inside the kernel after recording its first event with t >= T, the others
 * use the whole capacity and RESUME in the next chunk from their hand-off state (x, v, t, horizon, ar, stream position -- the last
 * recorded event), which the next chunk records again as its column 0.  Finished chains cost nothing; memory is the sum of the
 * chunks, O(sum of the history lengths), never chains x longest history.
 *
 * pdmp_sample_skeleton_until_device: one chunk.  DEVICE pointers, asynchronous on `stream`.
 *   resume = 0: the first chunk, C chains from xinit / vinit (d x C), streams (seed, chain_id_offset + c); chain_map must be NULL.
 *   resume = 1: C = the number of chains still running; chain_map[c] (int64, DEVICE) = the index of launch chain c in the FIRST
 *     chunk's ensemble -- its state slot and its random stream; xinit / vinit are ignored (may be NULL).  seed and
 *     chain_id_offset must be the first chunk's.  X, V, t (and diag) are this chunk's own buffers, d x capacity x C etc.
 *   n_recorded[c] (int64, DEVICE) = columns chain c recorded in this chunk (capacity, or fewer when it reached T).
 *   Runs on the group / lane kernels: every sampler (StickyZigZag: the hand-off state carries the active masks; pass diag->is_active),
 *   built-in and source-given targets, grid_size = 10 or 0, no tape; and (round 4) on the batched dense-precision forms -- Zig-Zag / BPS /
 *   Boomerang, first chunk of >= 64 chains, d <= 1024: the handle's workspace holds the chains' states between chunks.
 *   PDMP_ERR_UNSUPPORTED otherwise (use the fixed-count path + pdmp_close_at_time_device with a doubled capacity).
 * pdmp_close_chunk_device: pdmp_close_at_time_device restricted to the recorded columns of a chunk (stride = capacity):
 *   count[c] >= 0: the chain's history ends in this chunk with count[c] columns, the last one at exactly t = T (the closing
 *   column written in place over the overshooting event, :385-420; none when an event sits at exactly T);
 *   count[c] < 0: not finished, -count[c]-1 = n_recorded[c] columns, all of them events.
 * pdmp_until_advance_device: the bookkeeping between chunks, on the device (one block; ballot + prefix sums):
 *   for launch chain c with original index o = map_in ? map_in[c] : c:  start[c] = total[o];  total[o] += |columns of c| - skip
 *   (skip = 0 for the first chunk, 1 for a resumed one: its column 0 repeats the previous chunk's last event); the chains with
 *   count[c] < 0 are appended, in order, to map_out (their original indices) and *n_active (DEVICE int64) receives their number.
 *   total is C_first int64, zeroed by the caller before the first chunk; after the last chunk it holds every chain's history
 *   length, and pdmp_compact_offsets_device(total) the dense offsets.
 * pdmp_compact_chunk_device: columns [skip, |count[c]|) of every chain of one chunk into dense buffers (_trim_history for an
 *   ensemble): column k of launch chain c goes to dense column offsets[o] + start[c] + (k - skip). */
int pdmp_sample_skeleton_until_device(pdmp_handle h, int64_t C, int64_t capacity, int64_t chain_id_offset, const double *xinit,
                                      const double *vinit, uint64_t seed, double T, int32_t resume, const int64_t *chain_map,
                                      double *X, double *V, double *t, const pdmp_diag_t *diag, int64_t *n_recorded, void *stream);
int pdmp_close_chunk_device(pdmp_handle h, int64_t C, int64_t capacity, double *X, double *V, double *t, const pdmp_diag_t *diag,
                            double T, const int64_t *n_recorded, int64_t *count, void *stream);
int pdmp_until_advance_device(pdmp_handle h, int64_t C, const int64_t *count, int64_t skip, const int64_t *map_in, int64_t *total,
                              int64_t *start, int64_t *map_out, int64_t *n_active, void *stream);
int pdmp_compact_chunk_device(pdmp_handle h, int64_t C, int64_t capacity, const int64_t *count, int64_t skip, const int64_t *chain_map,
                              const int64_t *start, const int64_t *offsets, const double *X, const double *V, const double *t,
                              const pdmp_diag_t *diag, double *X_out, double *V_out, double *t_out, const pdmp_diag_t *diag_out, void *stream);

/* _trim_history (src/Composites.jl:198-214) for an ensemble: the ragged histories packed back to back into dense buffers.
 * pdmp_compact_offsets_device: offsets[c] = sum_{c' < c} max(count[c'], 0), offsets[C] = total columns (C+1 int64, DEVICE;
 *   wave shuffle scan + cross-wave carry, one block).
 * pdmp_compact_history_device: column k < count[c] of chain c moves to dense column offsets[c] + k of X_out, V_out (d x total),
 *   t_out (total) and of every diagnostic array present in both diag and diag_out. */
int pdmp_compact_offsets_device(pdmp_handle h, int64_t C, const int64_t *count, int64_t *offsets, void *stream);
int pdmp_compact_history_device(pdmp_handle h, int64_t C, int64_t n_sk, const int64_t *count, const int64_t *offsets,
                                const double *X, const double *V, const double *t, const pdmp_diag_t *diag,
                                double *X_out, double *V_out, double *t_out, const pdmp_diag_t *diag_out, void *stream);

/* Multi-device ensembles from ONE host process (SURVEY.md 5 "distributed communication backend", 8(e); the reference is
 * single-chain, single-thread: its stand-in is the `seed = 42 + i` loop of test/test_property_based.jl:138).  The C chains
 * shard into contiguous ranges over n_devices GPUs (pdmp_multi_shard_range: sizes differ by at most one); device r runs
 * pdmp_sample_skeleton_device on its range, on its own stream, with the GLOBAL chain ids (chain_id_offset + lo_r + c), so the
 * result does not depend on n_devices; nothing is exchanged while sampling.
 * gather = 0: X[r], V[r], t[r] (DEVICE pointers on device r) receive shard r only (d x n_sk x len_r ...).
 * gather = 1: X[r], V[r], t[r] are full-size (d x n_sk x C ...) buffers on every device; shard r is written at its chain
 *   offset and ONE collective per array over RCCL / xGMI (ncclAllGather in place when the shards are equal, else one grouped
 *   ncclBroadcast per shard = all-gather-v) leaves the whole ensemble on every device -- the north star's final exchange.
 * xinit, vinit: HOST pointers, d x C (16 d bytes per chain; each shard is copied to its device).  Blocking.
 * RCCL is loaded on first use (librccl.so.1); n_devices = 1 needs no RCCL. */
typedef struct pdmp_multi *pdmp_multi_handle;
int pdmp_multi_shard_range(int64_t C, int32_t rank, int32_t n_ranks, int64_t *lo, int64_t *hi);
int pdmp_multi_create(const pdmp_config_t *cfg, int32_t n_devices, const int32_t *devices /* NULL: 0..n-1 */, pdmp_multi_handle *out);
int pdmp_multi_destroy(pdmp_multi_handle m);
int pdmp_multi_sample_skeleton(pdmp_multi_handle m, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                               const double *xinit, const double *vinit, uint64_t seed, int32_t gather,
                               double *const *X, double *const *V, double *const *t);
/* kernel / exchange time of the last pdmp_multi_sample_skeleton (ms, max over devices; hipEvents on each device's stream) */
int pdmp_multi_get_timing(pdmp_multi_handle m, double *kernel_ms, double *exchange_ms);
/* The same step with the shards' initial states already ON their devices: xinit[r], vinit[r] = DEVICE pointers on device r holding
 * shard r only (d x len_r) -- nothing is staged over PCIe (the host-pointer entry copies 16 d bytes per chain per call).
 * pdmp_multi_init_states fills such buffers with the synthetic states of pdmp_init_states_device, keyed by the global chain id. */
int pdmp_multi_sample_skeleton_device(pdmp_multi_handle m, int64_t C, int64_t n_sk, int64_t chain_id_offset,
                                      const double *const *xinit, const double *const *vinit, uint64_t seed, int32_t gather,
                                      double *const *X, double *const *V, double *const *t);
int pdmp_multi_init_states(pdmp_multi_handle m, int64_t C, int64_t chain_id_offset, uint64_t seed, double x_mean, double x_scale,
                           int32_t v_kind, double *const *xinit, double *const *vinit);

/* Synchronises the handle's last stream and returns PDMP_ERR_ITERCAP if any chain of the launches
 * since the previous check hit the iteration cap (Q11 guard events are counted in *n_guard if non-NULL). */
int pdmp_check_status(pdmp_handle h, int64_t *n_guard);

/* hipEvent timing of the kernels launched through this handle (on the launch stream). */
int pdmp_get_timing(pdmp_handle h, pdmp_timing_t *out);
int pdmp_reset_timing(pdmp_handle h);

/* Diagnostics: number of GEMM phases (one d x d by d x 2C fp64 GEMM each) of the last dense-precision batched run
 * on this process (pdmpflux.jl_amd/csrc/dense_batched.hip); 0 if that path has not run. */
long long pdmp_dense_batched_last_phases(void);
/* ... and the number of right-hand-side columns those GEMMs multiplied (C per phase, 2C when P v is re-synchronised). */
long long pdmp_dense_batched_last_columns(void);
/* ... and which form served it: 0 none yet, 1 one GEMM per phase (d > 1024 or PDMP_DENSE_MODE=gemm), 2 incremental
 * (P x and P v kept current by O(d) updates, one GEMM per `PDMP_DENSE_RESYNC` = 256 events; last_phases() is 0 then). */
int pdmp_dense_batched_last_mode(void);

/* Device memory for the output arrays of a large ensemble, built from independent physical chunks (HIP virtual memory management;
 * chunk_bytes = 0: 1 GiB).  Plain pointers like any device allocation -- the caller still owns and frees them (pdmp_device_free, which
 * waits for the device first) -- but the launch time no longer depends on how the driver happened to place two 40 GB blocks relative
 * to each other: measured on MI355X at config-2 size, hipMalloc'ed arrays run 14.4 ms or ~18 ms per launch for as long as the allocation
 * lives (about one placement in five), chunked arrays 14.4 - 14.75 ms in every allocation (csrc/device_alloc.hip).
 * device = -1: the current device.  Errors: PDMP_ERR_ARGUMENT / PDMP_ERR_RUNTIME with pdmp_device_alloc_error(). */
int pdmp_device_alloc(int32_t device, size_t bytes, size_t chunk_bytes, void **out);
int pdmp_device_free(void *ptr);

/* The arrays one launch streams side by side (X, V of an ensemble first, then t and anything else), allocated like pdmp_device_alloc --
 * and drawn up to `draws` times: how fast the event rows of X and V can be written depends on where their physical chunks lie relative
 * to each other (14.3 ms or 17 - 19 ms for the same launch), a fresh set of chunks is a fresh draw, and a set can be
 * judged in 20 ms by a kernel with the launch's store pattern (one 64-byte line of each chain's X row and V row per step, chains
 * `stride_bytes` apart, 1 024 waves).  Every draw allocates all n arrays (as long as the free memory holds another set beside the best
 * one so far), times that writer on arrays 0 and 1, and only the fastest set survives; draw_ms[k] (may be NULL, room for `draws`
 * values) receives the times, *n_drawn how many sets were drawn.  draws <= 1 or n == 1: no measurement.  The arrays' contents are
 * undefined afterwards.  Each returned pointer is released with pdmp_device_free.  stride_bytes = 8 n_sk dim for X, V (C, n_sk, dim). */
int pdmp_device_alloc_streams(int32_t device, int32_t n, const size_t *bytes, size_t chunk_bytes, size_t stride_bytes, int32_t draws,
                              void **out_ptrs, float *draw_ms, int32_t *n_drawn);
/* The pool behind these allocators (round 5): pdmp_device_free PARKS an allocation -- address range and mapping intact -- and the next
 * pdmp_device_alloc of the same size on the same device, or the next pdmp_device_alloc_streams of the same set of sizes, returns it
 * (for a set judged before: *n_drawn = 0, nothing is drawn or timed again; microseconds instead of seconds).  Parked memory is released
 * when an allocation would otherwise fail, or by pdmp_device_pool_trim().  PDMP_ALLOC_POOL=0 in the environment disables the pool. */
int pdmp_device_pool_trim(void);
const char *pdmp_device_alloc_error(void);

/* Synthetic initial states of an ensemble, generated on the device (the ensemble API is new -- the reference starts its one chain
 * from user arrays, README.md:40-44 -- and BASELINE.md section 3 fixes the synthetic workloads as "xinit[:,c] ~ N(0,I), vinit[i,c] = +-1
 * from Philox key (seed, c) stream 1; chain RNG = stream 0"): stream 1 of chain c is its Philox stream with the top bit of the 64-bit
 * chain word set, so the initial states, like the trajectories, depend on the GLOBAL chain id only (sharding-invariant).
 * xinit[:, c] = x_mean + x_scale * N(0, I); vinit by v_kind: +-1 coordinates, N(0, I), or uniform on the unit sphere.
 * DEVICE pointers, d x C each, enqueued on `stream`.  Checker: orc_init_states (bit for bit). */
enum { PDMP_V_RADEMACHER = 0, PDMP_V_GAUSSIAN = 1, PDMP_V_SPHERE = 2 };
int pdmp_init_states_device(pdmp_handle h, int64_t C, int64_t chain_id_offset, uint64_t seed, double x_mean, double x_scale,
                            int32_t v_kind, double *xinit, double *vinit, void *stream);

/* Test hook: first n variates of stream (seed, chain) produced ON THE DEVICE with the engine's
 * generator; kind 0 uniform, 1 Exp(1), 2 normal vector.  out is a HOST pointer. */
int pdmp_debug_rng(pdmp_handle h, uint64_t seed, int64_t chain, int32_t kind, int64_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif
