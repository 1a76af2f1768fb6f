/*
 * julia_rng.c -- the random source of the reference, restated: Julia's `MersenneTwister(seed)` (AbstractPDMP.jl:100) with the
 * variate functions the sampling loop calls on it -- `randexp` (SamplingLoopInplace.jl:70,136,190), `rand` (:154; the samplers'
 * jumps), `randn` (BouncyParticleSamplers.jl:67, BoomerangSamplers.jl:65).
 *
 * TEST INFRASTRUCTURE: a tape PRODUCER.  The checker can draw from it instead of from Philox
 * (orc_config_t.julia_rng) and logs every variate it consumes, per kind, in consumption order; the engine then replays those
 * tapes through pdmp_set_tape on whichever kernel form the launch uses.  None of this is in /root/reference -- it is third-party
 * arithmetic (Julia's stdlib Random, itself a wrapper of dSFMT 2.2 by M. Saito and M. Matsumoto, and the ziggurat tables of
 * randmtzig.c as printed into Random/src/normal.jl) restated from the PUBLISHED algorithms:
 *   - dSFMT-19937: recursion, init_by_array, initial mask and period certification of dSFMT.c / dSFMT-params19937.h;
 *   - Julia's seeding: 1.0 - 1.10 `make_seed(n)` = the 32-bit limbs of n, little end first (JL_SEED_LIMBS); 1.11 `hash_seed(n)` = SHA-256
 *     over the same limbs, the digest's eight 32-bit words as the key (JL_SEED_SHA256);
 *   - `rand(Float64)` = the next [1, 2) double of the dSFMT sequence minus 1; `UInt52Raw` = its 52 fraction bits;
 *   - `randexp` / `randn`: Marsaglia & Tsang's ziggurat with 256 strips as in Random/src/normal.jl; its six tables regenerated in exact
 *     arithmetic by oracle/make_julia_zig_tables.py (the construction of randmtzig.c with Julia's constants: julia_zig_tables.inc).
 * VERIFIED HERE against values remembered from Julia's own documentation and forum posts, which pin the generator core, the limb
 * seeding, rand and randn (tests/test_julia_rng.py): rand(MersenneTwister(1234)) = 0.5908446386657102, rand(MersenneTwister(0)) =
 * 0.8236475079774124, rand(MersenneTwister(1), 2) = [0.23603334566204692, 0.34651701419196046], randn(MersenneTwister(1234)) =
 * 0.8673472019512456 (and the next values of rand(MersenneTwister(1234), 5)); eight entries of the tables of normal.jl (ki[1], ki[3], wi[1],
 * wi[2], ke[1], ke[3], we[1], we[2]); SHA-256 by the FIPS 180 vectors.  NOT verified against a running Julia (none here): the SHA-256
 * seeding of 1.11, the order of draws inside the unlikely branches, and the last bit of results that pass through exp / log in the
 * unlikely branches (Julia has its own implementations of both; here: the host libm).  "Unverified against Julia" is therefore what every result produced through this source says.
 */
#include "julia_rng.h"
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4) */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2 };
static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
void jl_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8] = { 0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19 };
    uint8_t block[64];
    const size_t total = ((len + 9 + 63) / 64) * 64;
    for (size_t off = 0; off < total; off += 64) {
        for (int i = 0; i < 64; ++i) {
            const size_t p = off + (size_t)i;
            uint8_t b = 0;
            if (p < len) b = msg[p];
            else if (p == len) b = 0x80;
            else if (p >= total - 8) b = (uint8_t)(((uint64_t)len * 8) >> (8 * (total - 1 - p)));
            block[i] = b;
        }
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)block[4 * i] << 24) | ((uint32_t)block[4 * i + 1] << 16) | ((uint32_t)block[4 * i + 2] << 8) | block[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K256[i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}

/* ------------------------------------------------------------------ dSFMT-19937 (dSFMT.c, dSFMT-params19937.h; little-endian host) */
#define DS_POS1 117
#define DS_SL1 19
#define DS_SR 12
#define DS_MSK1 0x000ffafffffffb3fULL
#define DS_MSK2 0x000ffdfffc90fffdULL
#define DS_FIX1 0x90014964b32f4329ULL
#define DS_FIX2 0x3b8d12ac548a7c7aULL
#define DS_PCV1 0x3d84e1ac0dc82880ULL
#define DS_PCV2 0x0000000000000001ULL
#define DS_LOW_MASK 0x000FFFFFFFFFFFFFULL
#define DS_HIGH_CONST 0x3FF0000000000000ULL

static inline void ds_recursion(uint64_t r[2], const uint64_t a[2], const uint64_t b[2], uint64_t lung[2]) {
    const uint64_t t0 = a[0], t1 = a[1], L0 = lung[0], L1 = lung[1];
    lung[0] = (t0 << DS_SL1) ^ (L1 >> 32) ^ (L1 << 32) ^ b[0];
    lung[1] = (t1 << DS_SL1) ^ (L0 >> 32) ^ (L0 << 32) ^ b[1];
    r[0] = (lung[0] >> DS_SR) ^ (lung[0] & DS_MSK1) ^ t0;
    r[1] = (lung[1] >> DS_SR) ^ (lung[1] & DS_MSK2) ^ t1;
}
static void ds_gen_rand_all(jl_mt_t *m) {                       /* dsfmt_gen_rand_all: the next JL_DSFMT_N 128-bit words, in place */
    jl_w128_t *w = m->st.w;
    uint64_t lung[2] = { w[JL_DSFMT_N].u[0], w[JL_DSFMT_N].u[1] };
    int i;
    ds_recursion(w[0].u, w[0].u, w[DS_POS1].u, lung);
    for (i = 1; i < JL_DSFMT_N - DS_POS1; ++i) ds_recursion(w[i].u, w[i].u, w[i + DS_POS1].u, lung);
    for (; i < JL_DSFMT_N; ++i) ds_recursion(w[i].u, w[i].u, w[i + DS_POS1 - JL_DSFMT_N].u, lung);
    w[JL_DSFMT_N].u[0] = lung[0]; w[JL_DSFMT_N].u[1] = lung[1];
}
static inline uint32_t ini1(uint32_t x) { return (x ^ (x >> 27)) * 1664525u; }
static inline uint32_t ini2(uint32_t x) { return (x ^ (x >> 27)) * 1566083941u; }
static void ds_init_by_array(jl_mt_t *m, const uint32_t *key, int key_length) {   /* dsfmt_chk_init_by_array */
    uint32_t *p = m->st.flat32;                                  /* (a union member: the 32-bit writes and the 64-bit reads below alias legally) */
    const int size = (JL_DSFMT_N + 1) * 4, lag = 11, mid = (size - lag) / 2;      /* size = 768 >= 623 -> lag 11 */
    int i, j, count;
    uint32_t r;
    memset(&m->st, 0x8b, sizeof m->st);
    count = (key_length + 1 > size) ? key_length + 1 : size;
    r = ini1(p[0] ^ p[mid % size] ^ p[(size - 1) % size]);
    p[mid % size] += r;
    r += (uint32_t)key_length;
    p[(mid + lag) % size] += r;
    p[0] = r;
    count--;
    for (i = 1, j = 0; (j < count) && (j < key_length); ++j) {
        r = ini1(p[i] ^ p[(i + mid) % size] ^ p[(i + size - 1) % size]);
        p[(i + mid) % size] += r;
        r += key[j] + (uint32_t)i;
        p[(i + mid + lag) % size] += r;
        p[i] = r;
        i = (i + 1) % size;
    }
    for (; j < count; ++j) {
        r = ini1(p[i] ^ p[(i + mid) % size] ^ p[(i + size - 1) % size]);
        p[(i + mid) % size] += r;
        r += (uint32_t)i;
        p[(i + mid + lag) % size] += r;
        p[i] = r;
        i = (i + 1) % size;
    }
    for (j = 0; j < size; ++j) {
        r = ini2(p[i] + p[(i + mid) % size] + p[(i + size - 1) % size]);
        p[(i + mid) % size] ^= r;
        r -= (uint32_t)i;
        p[(i + mid + lag) % size] ^= r;
        p[i] = r;
        i = (i + 1) % size;
    }
    for (i = 0; i < 2 * JL_DSFMT_N; ++i) m->st.flat[i] = (m->st.flat[i] & DS_LOW_MASK) | DS_HIGH_CONST;   /* initial_mask: doubles in [1, 2) */
    {                                                           /* period_certification */
        const uint64_t t0 = m->st.w[JL_DSFMT_N].u[0] ^ DS_FIX1, t1 = m->st.w[JL_DSFMT_N].u[1] ^ DS_FIX2;
        uint64_t inner = (t0 & DS_PCV1) ^ (t1 & DS_PCV2);
        for (i = 32; i > 0; i >>= 1) inner ^= inner >> i;
        if ((inner & 1) == 0) m->st.w[JL_DSFMT_N].u[1] ^= 1;   /* (DSFMT_PCV2 & 1) == 1: flip the lowest bit of the lung's second word */
    }
    m->idx = 2 * JL_DSFMT_N;                                     /* the buffer is consumed: the first draw generates */
}

/* dsfmt_chk_init_gen_rand (dSFMT.c): the generator's own 32-bit seeding -- Julia never uses it; here for the known-answer vector that ships
 * with the dSFMT sources (dSFMT.19937.out.txt: init_gen_rand(0), the first values in [1, 2)). */
void jl_dsfmt_init_gen_rand(jl_mt_t *m, uint32_t seed) {
    uint32_t *p = m->st.flat32;
    const int size = (JL_DSFMT_N + 1) * 4;
    p[0] = seed;
    for (int i = 1; i < size; ++i) p[i] = 1812433253u * (p[i - 1] ^ (p[i - 1] >> 30)) + (uint32_t)i;
    for (int i = 0; i < 2 * JL_DSFMT_N; ++i) m->st.flat[i] = (m->st.flat[i] & DS_LOW_MASK) | DS_HIGH_CONST;
    {
        const uint64_t t0 = m->st.w[JL_DSFMT_N].u[0] ^ DS_FIX1, t1 = m->st.w[JL_DSFMT_N].u[1] ^ DS_FIX2;
        uint64_t inner = (t0 & DS_PCV1) ^ (t1 & DS_PCV2);
        for (int i = 32; i > 0; i >>= 1) inner ^= inner >> i;
        if ((inner & 1) == 0) m->st.w[JL_DSFMT_N].u[1] ^= 1;
    }
    m->idx = 2 * JL_DSFMT_N;
}
void jl_dsfmt_close1_open2(uint32_t seed, int n, double *out) {
    jl_mt_t m;
    jl_dsfmt_init_gen_rand(&m, seed);
    for (int i = 0; i < n; ++i) {
        if (m.idx >= 2 * JL_DSFMT_N) { ds_gen_rand_all(&m); m.idx = 0; }
        union { uint64_t u; double d; } c; c.u = m.st.flat[m.idx++]; out[i] = c.d;
    }
}

void jl_mt_seed(jl_mt_t *m, int64_t seed, int seeding) {
    /* the 32-bit limbs of the (non-negative) seed, little end first: Julia's make_seed(n) (1.0 - 1.10: the key itself); 1.11 hashes the same
     * limbs (and a trailing 0x01 byte for a negative seed, whose limbs are those of ~seed) with SHA-256: hash_seed(seed::Integer) */
    uint32_t limbs[2];
    int nl = 0;
    const int neg = seed < 0;
    uint64_t s = neg ? ~(uint64_t)seed : (uint64_t)seed;
    do { limbs[nl++] = (uint32_t)s; s >>= 32; } while (s != 0 && nl < 2);
    if (seeding == JL_SEED_SHA256) {
        uint8_t bytes[9], dig[32];
        size_t nb = 0;
        for (int i = 0; i < nl; ++i) for (int b = 0; b < 4; ++b) bytes[nb++] = (uint8_t)(limbs[i] >> (8 * b));
        if (neg) bytes[nb++] = 0x01;
        jl_sha256(bytes, nb, dig);
        uint32_t key[8];
        for (int i = 0; i < 8; ++i) key[i] = (uint32_t)dig[4 * i] | ((uint32_t)dig[4 * i + 1] << 8) | ((uint32_t)dig[4 * i + 2] << 16) | ((uint32_t)dig[4 * i + 3] << 24);
        ds_init_by_array(m, key, 8);
    } else {
        ds_init_by_array(m, limbs, nl);
    }
}

static inline uint64_t jl_next_bits(jl_mt_t *m) {                /* the next [1, 2) double of the sequence, as its bit pattern */
    if (m->idx >= 2 * JL_DSFMT_N) { ds_gen_rand_all(m); m->idx = 0; }
    return m->st.flat[m->idx++];
}
double jl_rand(jl_mt_t *m) {                                      /* rand(rng) = rand(rng, CloseOpen12()) - 1.0 */
    const uint64_t b = jl_next_bits(m);
    double x; memcpy(&x, &b, 8);
    return x - 1.0;
}

/* ------------------------------------------------------------------ ziggurat (Random/src/normal.jl; tables by randmtzig.c's formulas) */
#define ZIG_NOR_R 3.6541528853610087963519472518            /* ziggurat_nor_r */
#define ZIG_NOR_INV_R (1.0 / ZIG_NOR_R)                     /* ziggurat_nor_inv_r = inv(ziggurat_nor_r) */
#define ZIG_EXP_R 7.6971174701310497140446280481            /* ziggurat_exp_r */
#include "julia_zig_tables.inc"                             /* g_ki, g_wi, g_fi, g_ke, g_we, g_fe: oracle/make_julia_zig_tables.py */
void jl_zig_tables(const uint64_t **ki, const double **wi, const double **fi, const uint64_t **ke, const double **we, const double **fe) {
    if (ki) *ki = g_ki;
    if (wi) *wi = g_wi;
    if (fi) *fi = g_fi;
    if (ke) *ke = g_ke;
    if (we) *we = g_we;
    if (fe) *fe = g_fe;
}
void jl_init_tables(void) {}                                /* (the tables are constants; kept for callers of the first version) */

double jl_randexp(jl_mt_t *m) {                                   /* randexp(rng), normal.jl */
    for (;;) {
        const uint64_t ri = jl_next_bits(m) & DS_LOW_MASK;        /* rand(rng, UInt52Raw()) & 0x000fffffffffffff */
        const int idx = (int)(ri & 0xFF);
        const double x = (double)ri * g_we[idx];
        if (ri < g_ke[idx]) return x;                             /* 98.9 % */
        if (idx == 0) return ZIG_EXP_R - log(jl_rand(m));         /* randexp_unlikely: the tail */
        if ((g_fe[idx - 1] - g_fe[idx]) * jl_rand(m) + g_fe[idx] < exp(-x)) return x;   /* fe[idx] - fe[idx+1] in 1-based indexing: the wedge */
    }
}
/* _randn(rng, r) of normal.jl: one ziggurat attempt on the 52 random bits r; a rejection in the wedge falls back to a fresh randn(rng)
 * (randn_unlikely ends in `return randn(rng)`), the tail loops on its own two draws */
static double jl_randn_bits(jl_mt_t *m, uint64_t r) {
    for (;;) {
        r &= DS_LOW_MASK;
        const int64_t rabs = (int64_t)(r >> 1);                   /* one bit for the sign */
        const int idx = (int)(rabs & 0xFF);
        const double x = (double)((r & 1) ? -rabs : rabs) * g_wi[idx];
        if ((uint64_t)rabs < g_ki[idx]) return x;                 /* 99.3 % */
        if (idx == 0) {                                           /* randn_unlikely: the tail */
            for (;;) {
                const double xx = -ZIG_NOR_INV_R * log(jl_rand(m));
                const double yy = -log(jl_rand(m));
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -ZIG_NOR_R - xx : ZIG_NOR_R + xx;
            }
        }
        if ((g_fi[idx - 1] - g_fi[idx]) * jl_rand(m) + g_fi[idx] < exp(-0.5 * x * x)) return x;   /* the wedge */
        r = jl_next_bits(m);                                      /* rejected: randn(rng) again */
    }
}
double jl_randn(jl_mt_t *m) { return jl_randn_bits(m, jl_next_bits(m)); }   /* randn(rng), normal.jl */

/* randn!(rng::MersenneTwister, A::Array{Float64}) (normal.jl, the MersenneTwister method generated for randn! / randexp!): arrays shorter than
 * 13 are filled by scalar draws; longer ones are FIRST filled with rand!(rng, A, CloseOpen12()) -- for n <= 383 that is _rand_max383!: the next n
 * doubles of the generator's cache, in order, exactly what n scalar draws would take -- and THEN every element is passed through
 * _randn(rng, reinterpret(UInt64, A[i])): the draws a rejected element needs (0.7 % of them) come after the whole fill, not in between.
 * This is what `randn!(rng, v)` (BouncyParticleSamplers.jl:67) and `randn(key, dim)` / `randn(key, 2, dim)` (ForwardEventChainMonteCarlo.jl)
 * do on a plain MersenneTwister for dim >= 13; an AbstractRNG WRAPPED around it (julia/dump_tape.jl's TapeRNG) takes the generic loop of
 * scalar draws instead (`scalar` != 0).  n > 383 (the dSFMT array-fill path with its alignment rules) is not restated: scalar order.
 * Recalled from the Julia sources, unverified against a running Julia like the rest of this file. */
void jl_randn_fill(jl_mt_t *m, int n, double *out, int scalar) {
    if (scalar || n < 13 || n > 383) {
        for (int i = 0; i < n; ++i) out[i] = jl_randn(m);
        return;
    }
    uint64_t bits[383];
    for (int i = 0; i < n; ++i) bits[i] = jl_next_bits(m);
    for (int i = 0; i < n; ++i) out[i] = jl_randn_bits(m, bits[i]);
}
