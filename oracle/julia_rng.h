/* julia_rng.h -- Julia's MersenneTwister / rand / randexp / randn restated from the published algorithms (see julia_rng.c).
 * TEST INFRASTRUCTURE: a tape producer for the checker; unverified against a running Julia. */
#ifndef JULIA_RNG_H
#define JULIA_RNG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define JL_DSFMT_N 191                       /* (19937 - 128) / 104 + 1 */
enum { JL_SEED_LIMBS = 1,                    /* Julia 1.0 - 1.10: dsfmt_init_by_array(make_seed(n)) */
       JL_SEED_SHA256 = 2 };                 /* Julia 1.11: dsfmt_init_by_array(hash_seed(n)), SHA-256 of the limbs */

typedef union jl_w128_t { uint64_t u[2]; uint32_t u32[4]; } jl_w128_t;   /* (dSFMT's w128_t: the initialisation works on 32-bit words) */
typedef struct jl_mt_t {
    union { jl_w128_t w[JL_DSFMT_N + 1]; uint64_t flat[2 * (JL_DSFMT_N + 1)]; uint32_t flat32[4 * (JL_DSFMT_N + 1)]; } st;   /* dSFMT state: N 128-bit
                                             * words of two [1, 2) doubles each + the lung (w[N]) */
    int idx;                                 /* next unread 64-bit word of w[0 .. N-1] */
} jl_mt_t;

void   jl_init_tables(void);                 /* builds the ziggurat tables (not thread-safe: call once before threads start) */
void   jl_mt_seed(jl_mt_t *m, int64_t seed, int seeding);
void   jl_dsfmt_close1_open2(uint32_t seed, int n, double *out);   /* dSFMT's own init_gen_rand(seed), then n values in [1, 2): the published test vector */
double jl_rand(jl_mt_t *m);                  /* rand(rng)     in [0, 1) */
double jl_randexp(jl_mt_t *m);               /* randexp(rng)  */
double jl_randn(jl_mt_t *m);                 /* randn(rng)    */
void   jl_randn_fill(jl_mt_t *m, int n, double *out, int scalar);   /* randn!(rng, A::Array{Float64}) of a plain MersenneTwister (bulk fill from 13 elements on), or n scalar draws */
void   jl_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
void   jl_zig_tables(const uint64_t **ki, const double **wi, const double **fi, const uint64_t **ke, const double **we, const double **fe);

#ifdef __cplusplus
}
#endif
#endif
