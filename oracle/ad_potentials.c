/* ad_potentials.c -- CHECKER ONLY (test infrastructure; see pdmp_oracle.h).
 *
 * grad-U obtained from a scalar potential U by forward-mode differentiation, as create_gradient_function does with
 * ForwardDiff.gradient (src/ADBackend.jl:95-109), nested inside the forward-mode time derivative of the rate
 * (ForwardDiff.derivative, src/UpperBound.jl:102,210).  ForwardDiff is a dependency of the reference, absent from its
 * tree (Project.toml compat "ForwardDiff"); restated here from the published forward-mode rules:
 *     (a*b)' = a'*b + a*b'     (a/b)' = (a' - (a/b)*b')/b     log' = a'/a     sin' = a'*cos     x^2 = x*x
 * with every product and sum rounded separately (no fused multiply-add), the partial with respect to x_i taken in a pass
 * of its own.  Potentials: the reference's own test potentials, test/test_config.jl:33-46, operation for operation
 * (Julia's `sum` left to right).
 */
#include "pdmp_oracle.h"
#include <math.h>
#include <stdlib.h>

/* first level: (value, d/dt) */
typedef struct { double v, e; } du;
static inline du du_add(du a, du b) { return (du){a.v + b.v, a.e + b.e}; }
static inline du du_sub(du a, du b) { return (du){a.v - b.v, a.e - b.e}; }
static inline du du_mul(du a, du b) { return (du){a.v * b.v, a.e * b.v + a.v * b.e}; }
static inline du du_div(du a, du b) { double q = a.v / b.v; return (du){q, (a.e - q * b.e) / b.v}; }
static inline du du_neg(du a) { return (du){-a.v, -a.e}; }
static inline du du_addd(du a, double b) { return (du){a.v + b, a.e}; }
static inline du du_subd(du a, double b) { return (du){a.v - b, a.e}; }
static inline du du_divd(du a, double b) { return (du){a.v / b, a.e / b}; }
static inline du du_dmul(double s, du a) { return (du){s * a.v, s * a.e}; }
static inline du du_log(du a) { return (du){(a.v > 0.0) ? orc_log(a.v) : NAN, a.e / a.v}; }
static inline void du_sincos(du a, du *s, du *c) {
    double sv, cv;
    orc_sincos(a.v, &sv, &cv);
    *s = (du){sv, a.e * cv};
    *c = (du){cv, -(a.e * sv)};
}

/* second level: (value, d/dx_i), each a first-level number */
typedef struct { du v, e; } d2;
static inline d2 d2_add(d2 a, d2 b) { return (d2){du_add(a.v, b.v), du_add(a.e, b.e)}; }
static inline d2 d2_sub(d2 a, d2 b) { return (d2){du_sub(a.v, b.v), du_sub(a.e, b.e)}; }
static inline d2 d2_mul(d2 a, d2 b) { return (d2){du_mul(a.v, b.v), du_add(du_mul(a.e, b.v), du_mul(a.v, b.e))}; }
static inline d2 d2_div(d2 a, d2 b) { du q = du_div(a.v, b.v); return (d2){q, du_div(du_sub(a.e, du_mul(q, b.e)), b.v)}; }
static inline d2 d2_neg(d2 a) { return (d2){du_neg(a.v), du_neg(a.e)}; }
static inline d2 d2_subd(d2 a, double b) { return (d2){du_subd(a.v, b), a.e}; }
static inline d2 d2_divd(d2 a, double b) { return (d2){du_divd(a.v, b), du_divd(a.e, b)}; }
static inline d2 d2_dmul(double s, d2 a) { return (d2){du_dmul(s, a.v), du_dmul(s, a.e)}; }
static inline d2 d2_sq(d2 a) { return d2_mul(a, a); }
static inline d2 d2_log(d2 a) { return (d2){du_log(a.v), du_div(a.e, a.v)}; }
static inline d2 d2_sin(d2 a) { du s, c; du_sincos(a.v, &s, &c); return (d2){s, du_mul(a.e, c)}; }

typedef struct { const double *xv, *xe; int i; } seeded_t;
static inline d2 X(const seeded_t *s, int j) { return (d2){{s->xv[j], s->xe[j]}, {j == s->i ? 1.0 : 0.0, 0.0}}; }

static d2 U_banana(const seeded_t *x, int d) {                 /* test/test_config.jl:33-36 */
    d2 mean_x2 = d2_subd(d2_sq(X(x, 0)), 1.0);
    d2 s = d2_add(d2_neg(d2_sq(X(x, 0))), d2_neg(d2_sq(d2_sub(X(x, 1), mean_x2))));
    if (d > 2) {
        d2 t = d2_sq(X(x, 2));
        for (int j = 3; j < d; ++j) t = d2_add(t, d2_sq(X(x, j)));
        s = d2_sub(s, t);
    }
    return d2_divd(d2_neg(s), 2.0);
}
static d2 U_funnel(const seeded_t *x, int d) {                 /* test/test_config.jl:38-42 */
    d2 v = X(x, 0);
    d2 t = d2_sq(X(x, 1));
    for (int j = 2; j < d; ++j) t = d2_add(t, d2_sq(X(x, j)));
    return d2_add(d2_add(d2_divd(d2_sq(v), 2.0), d2_dmul((double)(d - 1), d2_log(v))), d2_div(t, d2_dmul(2.0, d2_sq(v))));
}
static d2 U_ridged_gauss(const seeded_t *x, int d) {           /* test/test_config.jl:44-46 */
    d2 a = d2_sq(X(x, 0));
    d2 b = d2_sin(d2_dmul(10.0, X(x, 0)));
    for (int j = 1; j < d; ++j) { a = d2_add(a, d2_sq(X(x, j))); b = d2_add(b, d2_sin(d2_dmul(10.0, X(x, j)))); }
    return d2_add(d2_divd(a, 2.0), d2_dmul(0.1, b));
}

static d2 eval(int target, const seeded_t *s, int d) {
    switch (target) {
    case ORC_T_AD_BANANA: return U_banana(s, d);
    case ORC_T_AD_FUNNEL: return U_funnel(s, d);
    default: return U_ridged_gauss(s, d);
    }
}

/* gv[i] = dU/dx_i at x = xv, ge[i] = its derivative along the direction xe (the d/dt of the flow) */
void orc_ad_grad_dual(int target, int d, const double *xv, const double *xe, double *gv, double *ge) {
    for (int i = 0; i < d; ++i) {
        seeded_t s = {xv, xe, i};
        d2 u = eval(target, &s, d);
        gv[i] = u.e.v;
        ge[i] = u.e.e;
    }
}
double orc_ad_potential(int target, int d, const double *x) {
    double *z = calloc((size_t)d, sizeof(double));
    seeded_t s = {x, z, -1};
    d2 u = eval(target, &s, d);
    free(z);
    return u.v.v;
}
