/* sanitize_driver.c -- runs the checker (pdmp_oracle.c + ad_potentials.c, test infrastructure) under AddressSanitizer and
 * UndefinedBehaviorSanitizer on the CPU (`make -C oracle check-sanitize`; tests/test_oracle_sanitizers.py).  GPU sanitizers are not
 * available on the pool, so the C restatement -- the one piece of native code that runs on the CPU -- is what gets them: every sampler,
 * every target, both slope modes, all three summation orders, the Brent bound, tapes (incl. an exhausted one), the time-horizon
 * variant, sample_from_skeleton with ragged / prefix / dt methods.  Exit code 0 = no finding (the sanitizers abort otherwise). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pdmp_oracle.h"

static double urand(uint64_t *s) { *s = *s * 6364136223846793005ull + 1442695040888963407ull; return ((*s >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

static int run_case(int sampler, int target, int d, int arith, int G, int deriv, int grid, int signed_bound, int C, int n_sk) {
    orc_config_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.sampler = sampler; cfg.dim = d; cfg.grid_size = grid; cfg.signed_bound = signed_bound; cfg.adaptive = 1;
    cfg.deriv_mode = deriv; cfg.target_id = target; cfg.arith = arith; cfg.group_lanes = G;
    cfg.vectorized_bound = (sampler == ORC_ZIGZAG || sampler == ORC_SUZZ || sampler == ORC_STICKY);
    cfg.tmax = (sampler == ORC_BPS || sampler == ORC_BOOMERANG) ? 1.0 : 2.0;
    cfg.refresh_rate = (sampler == ORC_BPS || sampler == ORC_BOOMERANG) ? 0.3 : 0.0;
    cfg.fe_mix_p = 0.5; cfg.fe_switch = 1; cfg.fe_positive = 1; cfg.fe_speed_factor = 1.0;
    cfg.rh_step_size = 0.05; cfg.rh_phi = 1.5707963267948966;
    if (sampler == ORC_RHMC) { cfg.grid_size = 0; cfg.refresh_rate = 1.0; cfg.tmax = 10.0; cfg.adaptive = 0; cfg.signed_bound = 0; }
    uint64_t s = 12345u + (uint64_t)sampler * 977u + (uint64_t)d;
    int np = 0;
    double *params = NULL;
    if (target == ORC_T_GAUSS_DIAG) np = 2 * d;
    if (target == ORC_T_GAUSS_DENSE) np = d * d;
    if (sampler == ORC_STICKY) np += d;
    if (np) {
        params = (double *)malloc(sizeof(double) * (size_t)np);
        for (int i = 0; i < np; ++i) params[i] = 0.5 + urand(&s);
        if (target == ORC_T_GAUSS_DENSE) for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) params[i * d + j] = (i == j) ? 2.0 : 0.1 / (1 + abs(i - j));
        cfg.target_params = params; cfg.n_target_params = np;
    }
    if (orc_normalize_config(&cfg) != 0) { fprintf(stderr, "normalize: %s\n", orc_last_error()); free(params); return 1; }
    const size_t ne = (size_t)C * n_sk;
    double *x = malloc(8 * (size_t)C * d), *v = malloc(8 * (size_t)C * d);
    for (int i = 0; i < C * d; ++i) {
        x[i] = (target == ORC_T_BANANA) ? 1.0 + 0.1 * (urand(&s) - 0.5) : 2.0 * urand(&s) - 1.0;
        v[i] = (sampler == ORC_ZIGZAG || sampler == ORC_SUZZ || sampler == ORC_STICKY) ? (urand(&s) < 0.5 ? -1.0 : 1.0) : 2.0 * urand(&s) - 1.0;
    }
    if (sampler == ORC_BPS || sampler == ORC_FECMC)
        for (int c = 0; c < C; ++c) { double n = 0; for (int i = 0; i < d; ++i) n += v[c * d + i] * v[c * d + i]; n = sqrt(n); for (int i = 0; i < d; ++i) v[c * d + i] /= n; }
    double *X = malloc(8 * ne * d), *V = malloc(8 * ne * d), *t = malloc(8 * ne), *H = malloc(8 * ne), *AR = malloc(8 * ne), *EV = malloc(40 * ne);
    int32_t *EB = malloc(4 * ne), *RJ = malloc(4 * ne), *HH = malloc(4 * ne);
    uint8_t *IA = malloc(ne * d);
    if (sampler == ORC_STICKY) cfg.is_active_out = IA;
    int rc = orc_sample_skeleton(&cfg, C, n_sk, 3, x, v, 99, X, V, t, H, AR, EB, EV, RJ, HH, 2);
    if (rc != 0) fprintf(stderr, "case sampler %d target %d d %d: rc %d (%s)\n", sampler, target, d, rc, orc_last_error());
    if (rc == 0) {                                    /* the callers either side: the three sample_from_skeleton methods, the time-horizon variant */
        const int64_t N = 17;
        double *smp = malloc(8 * (size_t)C * N * (2 * d + 1));
        int64_t *cnt = malloc(8 * (size_t)C), *lens = malloc(8 * (size_t)C);
        for (int c = 0; c < C; ++c) lens[c] = n_sk - c;                 /* ragged histories */
        rc |= orc_sample_from_skeleton_ex(&cfg, 0, C, n_sk, X, V, t, sampler == ORC_STICKY ? IA : NULL, lens, 0, N, 0.0, 0, smp, NULL, 1);
        rc |= orc_sample_from_skeleton_ex(&cfg, 1, C, n_sk, X, V, t, sampler == ORC_STICKY ? IA : NULL, NULL, 0, N, t[n_sk - 1] / 40.0, 1, smp, cnt, 1);
        rc |= orc_sample_from_skeleton_ex(&cfg, 2, C, n_sk, X, V, t, sampler == ORC_STICKY ? IA : NULL, NULL, n_sk / 2, N, t[n_sk / 2] / 30.0, 0, smp, cnt, 1);
        double T = t[n_sk - 1];                                         /* a horizon every chain reaches well within the capacity */
        for (int c = 1; c < C; ++c) if (t[(size_t)c * n_sk + n_sk - 1] < T) T = t[(size_t)c * n_sk + n_sk - 1];
        T *= 0.6;
        rc |= orc_sample_skeleton_until(&cfg, C, n_sk, 3, x, v, 99, T, X, V, t, H, AR, EB, EV, RJ, HH, cnt, 1) < 0;
        for (int c = 0; c < C; ++c) if (cnt[c] < 2) rc |= 1;
        rc |= orc_sample_skeleton_until(&cfg, C, 4, 3, x, v, 99, T, X, V, t, H, AR, EB, EV, RJ, HH, cnt, 1) < 0;   /* capacity too small: negative counts, no overrun */
        if (rc != 0) fprintf(stderr, "callers of sampler %d target %d d %d deriv %d arith %d: %s (counts %lld %lld)\n", sampler, target, d, deriv, arith, orc_last_error(), (long long)cnt[0], (long long)cnt[1]);
        free(smp); free(cnt); free(lens);
    }
    double chk = 0;
    for (size_t i = 0; i < ne; ++i) chk += t[i];
    if (!(chk == chk)) { fprintf(stderr, "NaN event times\n"); rc = 1; }
    free(x); free(v); free(X); free(V); free(t); free(H); free(AR); free(EV); free(EB); free(RJ); free(HH); free(IA); free(params);
    return rc;
}

int main(void) {
    int bad = 0, n = 0;
    const int samplers[] = {ORC_ZIGZAG, ORC_BPS, ORC_BOOMERANG, ORC_FECMC, ORC_SUZZ, ORC_RHMC, ORC_STICKY};
    const int targets[] = {ORC_T_GAUSS_ISO, ORC_T_GAUSS_DIAG, ORC_T_BANANA};
    for (int si = 0; si < 7; ++si)
        for (int ti = 0; ti < 3; ++ti)
            for (int deriv = 0; deriv < 2; ++deriv) {
                const int d = 2 + 3 * si + ti;
                bad += run_case(samplers[si], targets[ti], d, ORC_ARITH_REFERENCE, 1, deriv, 10, 1, 3, 40); ++n;
                bad += run_case(samplers[si], targets[ti], d, ORC_ARITH_GROUP, 1 << (ti % 3), deriv, 10, 1, 3, 40); ++n;
                bad += run_case(samplers[si], targets[ti], d, ORC_ARITH_WAVE64, 1, deriv, 10, ti != 1, 2, 25); ++n;
            }
    bad += run_case(ORC_ZIGZAG, ORC_T_GAUSS_ISO, 5, ORC_ARITH_GROUP, 1, 0, 0, 1, 2, 30); ++n;          /* Brent constant bound */
    bad += run_case(ORC_BPS, ORC_T_BANANA, 4, ORC_ARITH_REFERENCE, 1, 0, 0, 1, 2, 30); ++n;
    bad += run_case(ORC_ZIGZAG, ORC_T_GAUSS_DENSE, 9, ORC_ARITH_REFERENCE, 1, 0, 10, 1, 2, 30); ++n;     /* dense precision */
    bad += run_case(ORC_ZIGZAG, ORC_T_GAUSS_ISO, 50, ORC_ARITH_GROUP, 4, 0, 10, 1, 5, 60); ++n;          /* config 2's shape */
    bad += run_case(ORC_ZIGZAG, ORC_T_GAUSS_ISO, 7, ORC_ARITH_REFERENCE, 1, 0, 64, 1, 2, 20); ++n;       /* the largest grid */
    /* Julia's random source restated (oracle/julia_rng.c): the three draw kinds under both seedings, then skeletons drawn from it with the
     * tapes of what every chain consumed recorded (the producer side of pdmp_set_tape) -- incl. a tape capacity that is too small (-6, no overrun) */
    for (int seeding = 1; seeding <= 2; ++seeding)
        for (int kind = 0; kind < 3; ++kind) {
            double buf[1500];
            orc_julia_draws(1234 + kind, seeding, kind, 1500, buf);
            for (int i = 0; i < 1500; ++i) if (!(buf[i] == buf[i]) || (kind != 2 && buf[i] < 0.0) || (kind == 0 && buf[i] >= 1.0)) { fprintf(stderr, "julia draw kind %d seeding %d: %g\n", kind, seeding, buf[i]); bad++; break; }
            ++n;
        }
    {
        const int smp[] = {ORC_ZIGZAG, ORC_BPS, ORC_BOOMERANG, ORC_FECMC};
        for (int si = 0; si < 4; ++si)
            for (int seeding = 1; seeding <= 2; ++seeding)
                for (int small = 0; small < 2; ++small) {
                    orc_config_t cfg;
                    memset(&cfg, 0, sizeof cfg);
                    const int d = 6 + si, C = 3, n_sk = 30;
                    cfg.sampler = smp[si]; cfg.dim = d; cfg.grid_size = 10; cfg.arith = ORC_ARITH_REFERENCE; cfg.group_lanes = 1;
                    cfg.target_id = ORC_T_GAUSS_ISO; cfg.adaptive = 1; cfg.signed_bound = 1; cfg.max_iter_per_event = 100000;
                    cfg.vectorized_bound = smp[si] == ORC_ZIGZAG;
                    cfg.tmax = smp[si] == ORC_ZIGZAG ? 2.0 : 1.0; cfg.refresh_rate = (smp[si] == ORC_BPS || smp[si] == ORC_BOOMERANG) ? 0.3 : 0.0;
                    cfg.fe_mix_p = 0.5; cfg.fe_switch = 1; cfg.fe_positive = 1; cfg.fe_speed_factor = 1.0;
                    cfg.julia_rng = seeding;
                    const int64_t cap = small ? 7 : 6000;
                    double *te = malloc(8 * (size_t)C * cap), *tu = malloc(8 * (size_t)C * cap), *tn = malloc(8 * (size_t)C * cap);
                    int64_t cnt[9];
                    cfg.jl_tape_exp = te; cfg.jl_tape_uni = tu; cfg.jl_tape_nrm = tn; cfg.jl_cap_exp = cfg.jl_cap_uni = cfg.jl_cap_nrm = cap; cfg.jl_count = cnt;
                    int rc = orc_normalize_config(&cfg);
                    double x[3 * 9], v[3 * 9];
                    uint64_t s = 777u + (uint64_t)si;
                    for (int i = 0; i < C * d; ++i) { x[i] = 2.0 * urand(&s) - 1.0; v[i] = smp[si] == ORC_ZIGZAG ? (urand(&s) < 0.5 ? -1.0 : 1.0) : 2.0 * urand(&s) - 1.0; }
                    const size_t ne = (size_t)C * n_sk;
                    double *X = malloc(8 * ne * d), *V = malloc(8 * ne * d), *t = malloc(8 * ne);
                    if (rc == 0) rc = orc_sample_skeleton(&cfg, C, n_sk, 0, x, v, 42, X, V, t, NULL, NULL, NULL, NULL, NULL, NULL, 1);
                    if (small ? rc != -6 : rc != 0) { fprintf(stderr, "julia source sampler %d seeding %d cap %lld: rc %d (%s)\n", smp[si], seeding, (long long)cap, rc, orc_last_error()); bad++; }
                    if (!small && rc == 0) for (int c = 0; c < C; ++c) if (cnt[3 * c] < n_sk - 1 || cnt[3 * c] > cap) { fprintf(stderr, "julia source: exponential count %lld\n", (long long)cnt[3 * c]); bad++; }
                    free(te); free(tu); free(tn); free(X); free(V); free(t);
                    ++n;
                }
    }
    printf("sanitize_driver: %d cases, %d failed\n", n, bad);
    return bad ? 1 : 0;
}
