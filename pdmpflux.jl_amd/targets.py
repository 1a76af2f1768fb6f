"""Target descriptors: the potentials the engine can differentiate on the GPU.

The reference takes a Julia closure ``U`` (``ZigZagAD(dim, U)``, ZigZagSamplers.jl:118) or ``grad-U``
(``ZigZag(dim, grad-U)``, :58); a closure cannot cross a C ABI, so the drop-in takes a descriptor.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

T_GAUSS_ISO, T_GAUSS_DIAG, T_BANANA, T_GAUSS_DENSE, T_USER = 0, 1, 2, 3, 4
SRC_POTENTIAL, SRC_GRADIENT = 0, 1


@dataclass(frozen=True)
class GaussianIso:
    """U(x) = sum(x.^2)/2  (README.md:36-38, test/test_config.jl:25-31)."""
    target_id: int = T_GAUSS_ISO

    def params(self, dim):
        return None


@dataclass(frozen=True)
class GaussianDiag:
    """U(x) = sum(prec .* (x - mu).^2)/2."""
    mu: tuple
    prec: tuple
    target_id: int = T_GAUSS_DIAG

    def __init__(self, mu, prec):
        object.__setattr__(self, "mu", tuple(np.asarray(mu, dtype=np.float64).ravel()))
        object.__setattr__(self, "prec", tuple(np.asarray(prec, dtype=np.float64).ravel()))
        object.__setattr__(self, "target_id", T_GAUSS_DIAG)

    def params(self, dim):
        if len(self.mu) != dim or len(self.prec) != dim:
            from ._lib import DimensionMismatch
            raise DimensionMismatch(f"GaussianDiag parameters must have length dim ({dim})")
        return np.concatenate([np.asarray(self.mu), np.asarray(self.prec)])


@dataclass(frozen=True)
class Banana:
    """U_banana of test/test_config.jl:33-36 (the README's ``grad-U_banana`` returns a scalar and is
    not a gradient, README.md:62-65)."""
    target_id: int = T_BANANA

    def params(self, dim):
        return None


@dataclass(frozen=True)
class GaussianPrecision:
    """U(x) = x' P x / 2 with a dense symmetric precision matrix P (BASELINE config 4: grad-U = P x).  Zig-Zag, BPS and Boomerang
    (forward-mode slopes, grid_size = 10); ensembles of >= 64 chains run the batched forms of csrc/dense_batched.hip (products with P
    as fp64 GEMMs over all chains: Zig-Zag incrementally, BPS / Boomerang one GEMM per event)."""
    P: tuple
    dim: int
    target_id: int = T_GAUSS_DENSE

    def __init__(self, P):
        P = np.asarray(P, dtype=np.float64)
        if P.ndim != 2 or P.shape[0] != P.shape[1]:
            from ._lib import DimensionMismatch
            raise DimensionMismatch("P must be a square matrix")
        object.__setattr__(self, "P", tuple(np.asfortranarray(P).ravel(order="F")))   # column-major, as Julia
        object.__setattr__(self, "dim", P.shape[0])
        object.__setattr__(self, "target_id", T_GAUSS_DENSE)

    def params(self, dim):
        if dim != self.dim:
            from ._lib import DimensionMismatch
            raise DimensionMismatch(f"P is {self.dim}x{self.dim} but dim = {dim}")
        return np.asarray(self.P)

    def __repr__(self):
        return f"GaussianPrecision(dim={self.dim})"


@dataclass(frozen=True)
class SourcePotential:
    """A potential ``U`` given as HIP source -- the ``U::Function`` of ``ZigZagAD(dim, U)`` / ``BPSAD`` / ``BoomerangAD``
    (ZigZagSamplers.jl:118-126), differentiated on the GPU by forward mode as create_gradient_function does with
    ForwardDiff (ADBackend.jl:95-109) and compiled at run time (pdmp_create_from_source)::

        SourcePotential('''
            template <class X> __device__ auto U(const X &x, int d) {
                auto s = sq(x[0]);                       // x[j]: coordinate j, 0-based;  sq(a) = a^2
                for (int j = 1; j < d; ++j) s = s + sq(x[j]);
                return s / 2.0;
            }''')

    Available on every number type the engine differentiates with: ``+ - * /``, ``sq``, ``sqrt_``, ``log_``, ``exp_``, ``sin_``, ``cos_``.
    """
    source: str
    kind: int = SRC_POTENTIAL
    target_id: int = T_USER

    def params(self, dim):
        return None

    def __repr__(self):
        return f"{type(self).__name__}(<{len(self.source)} chars>)"


@dataclass(frozen=True)
class SourceGradient(SourcePotential):
    """An already-differentiated gradient given as HIP source -- the ``grad-U`` of ``ZigZag(dim, grad-U)``
    (ZigZagSamplers.jl:58; ADBackend.jl:55-66), one coordinate at a time::

        SourceGradient("template <class X> __device__ auto grad_i(int i, const X &x, int d) { return x[i]; }")
    """
    kind: int = SRC_GRADIENT


# The reference's own test potentials (test/test_config.jl:33-46) restated as sources, operation for operation
# (Julia's x^2 is x*x; `sum` runs left to right).
U_BANANA_SOURCE = """
template <class X> __device__ auto U(const X &x, int d) {            // test/test_config.jl:33-36
    auto mean_x2 = sq(x[0]) - 1.0;
    auto s = -sq(x[0]) + -sq(x[1] - mean_x2);
    if (d > 2) {
        auto t = sq(x[2]);
        for (int j = 3; j < d; ++j) t = t + sq(x[j]);
        s = s - t;
    }
    return -s / 2.0;
}
"""
U_FUNNEL_SOURCE = """
template <class X> __device__ auto U(const X &x, int d) {            // test/test_config.jl:38-42
    auto v = x[0];
    auto t = sq(x[1]);
    for (int j = 2; j < d; ++j) t = t + sq(x[j]);
    return sq(v) / 2.0 + (double)(d - 1) * log_(v) + t / (2.0 * sq(v));
}
"""
U_RIDGED_GAUSS_SOURCE = """
template <class X> __device__ auto U(const X &x, int d) {            // test/test_config.jl:44-46
    auto a = sq(x[0]);
    auto b = sin_(10.0 * x[0]);
    for (int j = 1; j < d; ++j) { a = a + sq(x[j]); b = b + sin_(10.0 * x[j]); }
    return a / 2.0 + 0.1 * b;
}
"""
