// user_ad.hpp -- forward-mode numbers for potentials given as source (user_jit.hip): the engine's replacement for
// ForwardDiff.gradient in create_gradient_function (src/ADBackend.jl:30-142) and for its nesting inside
// ForwardDiff.derivative (src/UpperBound.jl:102,210).  ForwardDiff is a dependency of the reference, not part of its
// tree; the rules below are the published forward-mode rules, written out so that the CPU checker can restate them
// operation for operation (DESIGN.md "Arithmetic contract": no fused multiply-add, the order written here).
//
//   Fwd<T>{v, e}: value and one partial, over T = double (plain gradient) or T = Dual (gradient of a position that
//   itself carries d/dt: the nested case of the grid bound).  One partial per pass: coordinate i of grad-U is the
//   partial of U evaluated on x seeded with e_j = (j == i) -- ForwardDiff's chunks compute the same partials, several
//   per pass.
#pragma once
#include "wave_math.hpp"

PDMP_NS_BEGIN

template <class T> struct Fwd { T v, e; };

template <class T> __device__ __forceinline__ T lift(double a);
template <> __device__ __forceinline__ double lift<double>(double a) { return a; }
template <> __device__ __forceinline__ Dual lift<Dual>(double a) { return {a, 0.0}; }

// ---- arithmetic: Fwd op Fwd, Fwd op double, double op Fwd
template <class T> __device__ __forceinline__ Fwd<T> operator+(Fwd<T> a, Fwd<T> b) { return {a.v + b.v, a.e + b.e}; }
template <class T> __device__ __forceinline__ Fwd<T> operator-(Fwd<T> a, Fwd<T> b) { return {a.v - b.v, a.e - b.e}; }
template <class T> __device__ __forceinline__ Fwd<T> operator*(Fwd<T> a, Fwd<T> b) { return {a.v * b.v, a.e * b.v + a.v * b.e}; }
template <class T> __device__ __forceinline__ Fwd<T> operator/(Fwd<T> a, Fwd<T> b) { const T q = a.v / b.v; return {q, (a.e - q * b.e) / b.v}; }
template <class T> __device__ __forceinline__ Fwd<T> operator-(Fwd<T> a) { return {-a.v, -a.e}; }
template <class T> __device__ __forceinline__ Fwd<T> operator+(Fwd<T> a, double b) { return {a.v + b, a.e}; }
template <class T> __device__ __forceinline__ Fwd<T> operator+(double a, Fwd<T> b) { return {a + b.v, b.e}; }
template <class T> __device__ __forceinline__ Fwd<T> operator-(Fwd<T> a, double b) { return {a.v - b, a.e}; }
template <class T> __device__ __forceinline__ Fwd<T> operator-(double a, Fwd<T> b) { return {a - b.v, -b.e}; }
template <class T> __device__ __forceinline__ Fwd<T> operator*(Fwd<T> a, double b) { return {a.v * b, a.e * b}; }
template <class T> __device__ __forceinline__ Fwd<T> operator*(double a, Fwd<T> b) { return {a * b.v, a * b.e}; }
template <class T> __device__ __forceinline__ Fwd<T> operator/(Fwd<T> a, double b) { return {a.v / b, a.e / b}; }
template <class T> __device__ __forceinline__ Fwd<T> operator/(double a, Fwd<T> b) { const T q = a / b.v; return {q, (-(q * b.e)) / b.v}; }

// ---- elementary functions on double, Dual and Fwd<T> (same rule at every level):
//   sq(a) = a*a (Julia's x^2, literal_pow)   sqrt' = 1/(2 sqrt)   log' = 1/a   exp' = exp   sin' = cos   cos' = -sin
__device__ __forceinline__ double sq(double a) { return a * a; }
__device__ __forceinline__ Dual sq(Dual a) { return a * a; }
template <class T> __device__ __forceinline__ Fwd<T> sq(Fwd<T> a) { return a * a; }

__device__ __forceinline__ double sqrt_(double a) { return __dsqrt_rn(a); }
__device__ __forceinline__ Dual sqrt_(Dual a) { const double r = __dsqrt_rn(a.v); return {r, a.e / (2.0 * r)}; }
template <class T> __device__ __forceinline__ Fwd<T> sqrt_(Fwd<T> a) { const T r = sqrt_(a.v); return {r, a.e / (2.0 * r)}; }

__device__ __forceinline__ double log_(double a) { return (a > 0.0) ? det_log(a) : __builtin_nan(""); }   // normal positive arguments
__device__ __forceinline__ Dual log_(Dual a) { return {log_(a.v), a.e / a.v}; }
template <class T> __device__ __forceinline__ Fwd<T> log_(Fwd<T> a) { return {log_(a.v), a.e / a.v}; }

// exp' = exp (round 3: the reference's own test potentials need it -- U_Mixture of test/test_integration.jl:67-75 is a log of a sum of
// exponentials, the strict gradient of test/test_samplers.jl:142 is tanh(x/2) = (e^x - 1) / (e^x + 1))
__device__ __forceinline__ double exp_(double a) { return det_exp(a); }
__device__ __forceinline__ Dual exp_(Dual a) { const double r = det_exp(a.v); return {r, a.e * r}; }
template <class T> __device__ __forceinline__ Fwd<T> exp_(Fwd<T> a) { const T r = exp_(a.v); return {r, a.e * r}; }

__device__ __forceinline__ void sincos_(double a, double &s, double &c) { det_sincos(a, s, c); }
__device__ __forceinline__ void sincos_(Dual a, Dual &s, Dual &c) {
    double sv, cv;
    det_sincos(a.v, sv, cv);
    s = {sv, a.e * cv};
    c = {cv, -(a.e * sv)};
}
template <class T> __device__ __forceinline__ void sincos_(Fwd<T> a, Fwd<T> &s, Fwd<T> &c) {
    T sv, cv;
    sincos_(a.v, sv, cv);
    s = {sv, a.e * cv};
    c = {cv, -(a.e * sv)};
}
template <class N> __device__ __forceinline__ N sin_(N a) { N s, c; sincos_(a, s, c); return s; }
template <class N> __device__ __forceinline__ N cos_(N a) { N s, c; sincos_(a, s, c); return c; }

PDMP_NS_END
