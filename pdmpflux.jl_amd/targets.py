"""Target descriptors: the potentials the engine can differentiate on the GPU.

The reference takes a Julia closure ``U`` (``ZigZagAD(dim, U)``, ZigZagSamplers.jl:118) or ``grad-U``
(``ZigZag(dim, grad-U)``, :58); a closure cannot cross a C ABI, so the drop-in takes a descriptor.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

T_GAUSS_ISO, T_GAUSS_DIAG, T_BANANA, T_GAUSS_DENSE = 0, 1, 2, 3


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
    """U(x) = x' P x / 2 with a dense symmetric precision matrix P (BASELINE config 4: grad-U = P x)."""
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
