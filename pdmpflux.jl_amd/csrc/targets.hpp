// targets.hpp -- built-in potentials U: gradient evaluated on doubles or forward-mode duals, in the
// wave layout (lane l owns coordinates l, l+64, ... in registers).  These replace the Julia closure
// `grad-U` of ZigZagSamplers.jl:37 / ADBackend.jl:30-142 (closures cannot cross a C ABI).
#pragma once
#include "wave_math.hpp"

PDMP_NS_BEGIN

enum : int { T_GAUSS_ISO = 0, T_GAUSS_DIAG = 1, T_BANANA = 2, T_GAUSS_DENSE = 3, T_USER = 4 };

template <int TARGET, int K>
struct Target;

// U = |x|^2 / 2  (README.md:36-38):  g = x
template <int K>
struct Target<T_GAUSS_ISO, K> {
    __device__ __forceinline__ void load(const double *, int, int) {}
    template <class T>
    __device__ __forceinline__ void grad(const T (&x)[K], T (&g)[K], int) const {
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = x[k];
    }
};

// U = sum prec_i (x_i - mu_i)^2 / 2 ; params = mu[d], prec[d]
template <int K>
struct Target<T_GAUSS_DIAG, K> {
    double mu[K], prec[K];
    __device__ __forceinline__ void load(const double *params, int d, int lane) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = lane + WAVE * k;
            mu[k] = (i < d) ? params[i] : 0.0;
            prec[k] = (i < d) ? params[d + i] : 0.0;
        }
    }
    __device__ __forceinline__ void grad(const double (&x)[K], double (&g)[K], int) const {
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = prec[k] * (x[k] - mu[k]);
    }
    __device__ __forceinline__ void grad(const Dual (&x)[K], Dual (&g)[K], int) const {
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = {prec[k] * (x[k].v - mu[k]), prec[k] * x[k].e};
    }
};

// U_banana (test/test_config.jl:33-36): r = x2 - (x1^2 - 1); g1 = x1 - 2 x1 r, g2 = r, g_k = x_k
template <int K>
struct Target<T_BANANA, K> {
    int d;
    __device__ __forceinline__ void load(const double *, int dim, int) { d = dim; }
    template <class T>
    __device__ __forceinline__ void grad(const T (&x)[K], T (&g)[K], int lane) const {
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = x[k];
        if (d < 2) return;
        const T x0 = readlane_t(x[0], 0), x1 = readlane_t(x[0], 1);
        const T sq = x0 * x0;
        const T m = sq - 1.0;
        const T r = x1 - m;
        const T a = 2.0 * x0;
        const T pr = a * r;
        const T g0 = x0 - pr;
        if (lane == 0) g[0] = g0;
        if (lane == 1) g[0] = r;
    }
};

PDMP_NS_END
