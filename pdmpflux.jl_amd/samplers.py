"""Sampler constructors with the reference's names, keyword arguments and defaults:
``ZigZag``/``ZigZagAD`` (src/Samplers/ZigZagSamplers.jl:58-126), ``BPS``/``BPSAD``
(BouncyParticleSamplers.jl:21-94), ``Boomerang``/``BoomerangAD`` (BoomerangSamplers.jl:21-87).

The second positional argument is a target descriptor (``targets.py``) instead of a Julia closure.
Objects are cheap host-side records; the GPU engine handle is created on first use.
"""
from __future__ import annotations

import ctypes as C
import math
import warnings

import numpy as np

from . import _lib
from ._lib import ArgumentError, DimensionMismatch

ZIGZAG, BPS_ID, BOOMERANG = 0, 1, 2
DERIV_FORWARD, DERIV_FINITE_DIFF = 0, 1

# AbstractPDMP.jl:18-28 (_pdmp_ad_backend) and UpperBound.jl:209-226: every AD backend yields the exact
# t-derivative, which the engine computes with forward-mode duals; "FiniteDiff" is central differences.
_EXACT = {"ForwardDiff", "Zygote", "ReverseDiff", "Enzyme", "PolyesterForwardDiff"}
_FD = {"FiniteDiff", "Undefined", ""}


def _deriv_mode(ad_backend) -> int:
    ab = "FiniteDiff" if ad_backend is None else str(ad_backend)
    if ab in _EXACT:
        return DERIV_FORWARD
    if ab in _FD:
        return DERIV_FINITE_DIFF
    raise ArgumentError(f"Unsupported AD_backend: {ab}")            # UpperBound.jl:120,226


class AbstractPDMP:
    """Common record (fields of ZigZagSamplers.jl:35-52 that the sampling loop reads)."""

    sampler_id = -1
    name = "AbstractPDMP"

    def __init__(self, dim, target, *, grid_size, tmax, refresh_rate, vectorized_bound, signed_bound,
                 adaptive, AD_backend, gaussian_velocity=False, device=-1, max_iter_per_event=0, rng_version=0):
        if not isinstance(dim, (int, np.integer)) or isinstance(dim, bool):
            raise TypeError("dim must be an Int")                    # MethodError in Julia
        if target is None or not hasattr(target, "target_id"):
            raise TypeError("target must be a target descriptor (GaussianIso(), GaussianDiag(..), Banana())")
        if dim <= 0:
            raise ArgumentError(f"dimension dim must be positive. Current value: {dim}")      # ZigZagSamplers.jl:62-64
        if grid_size < 0:
            raise ArgumentError(f"grid_size must be non-negative. Current value: {grid_size}")  # :66-68
        tmax = float(tmax)
        if tmax == 0.0:                                              # :73
            tmax, adaptive = 1.0, True
        self.dim = int(dim)
        self.target = target
        self.grid_size = int(grid_size)
        self.tmax = tmax
        self.refresh_rate = float(refresh_rate)
        self.vectorized_bound = bool(vectorized_bound)
        self.signed_bound = bool(signed_bound)
        self.adaptive = bool(adaptive)
        self.AD_backend = "FiniteDiff" if AD_backend in (None, "", "Undefined") else str(AD_backend)
        self.deriv_mode = _deriv_mode(AD_backend)
        self.gaussian_velocity = bool(gaussian_velocity)
        self.device = int(device)
        self.max_iter_per_event = int(max_iter_per_event)
        # engine-only keyword (not in the reference): the arithmetic contract of the build-defined random stream and of log / exp / sin / cos
        # (pdmp_config_t.rng_version: 0 = the default = 2; 1 = the contract of rounds 1-4, for the fixtures frozen then)
        self.rng_version = int(rng_version)
        self._params = target.params(self.dim)
        self._handle = None
        self.state = None

    # ---- engine handle -------------------------------------------------------------------
    def _config(self) -> _lib.Config:
        p, n = None, 0
        if self._params is not None:
            self._params = np.ascontiguousarray(self._params, dtype=np.float64)
            p, n = self._params.ctypes.data_as(C.POINTER(C.c_double)), self._params.size
        return _lib.Config(self.sampler_id, self.dim, self.grid_size, int(self.vectorized_bound),
                           int(self.signed_bound), int(self.adaptive), int(self.gaussian_velocity),
                           self.deriv_mode, self.target.target_id, self.device, self.tmax, self.refresh_rate,
                           p, n, self.max_iter_per_event, float(getattr(self, "mix_p", 0.5)), int(getattr(self, "ran_p", False)),
                           int(getattr(self, "switch", True)), int(getattr(self, "positive", True)), 0,
                           float(getattr(self, "step_size", 0.05)), float(getattr(self, "phi", math.pi / 2)),
                           float(getattr(self, "speed_factor", 1.0)), int(self.rng_version), 0)

    def handle(self):
        if self._handle is None:
            h = C.c_void_p()
            cfg = self._config()
            src = getattr(self.target, "source", None)
            if src is not None:     # potential / gradient given as source: compiled by the engine at run time
                _lib.check(_lib.lib().pdmp_create_from_source(C.byref(cfg), src.encode(), int(self.target.kind), C.byref(h)))
            else:
                _lib.check(_lib.lib().pdmp_create(C.byref(cfg), C.byref(h)))
            self._handle = h
        return self._handle

    def group_lanes(self) -> int:
        """Arithmetic layout of the engine for this configuration (pdmp_get_group_lanes): lanes per chain
        G of the group kernel, or 0 for the wave-per-chain kernel."""
        return int(_lib.lib().pdmp_get_group_lanes(self.handle()))

    def last_launch_form(self) -> int:
        """Kernel form of the last launch (pdmp_last_launch_form): 1 wave per chain, 2 group (LDS), 3 group (registers), 4 one lane
        per chain, 5 dense-precision path, 6 potential given as source, 7 a pair of lanes per chain (lane kernel, d = 53..104 and
        mid-sized launches at d <= 52)."""
        return int(_lib.lib().pdmp_last_launch_form(self.handle()))

    def suggest_group_lanes(self, C: int) -> int:
        """Lanes per chain that fill the GPU for launches of ``C`` chains (pdmp_suggest_group_lanes)."""
        return int(_lib.lib().pdmp_suggest_group_lanes(self.handle(), int(C)))

    def set_group_lanes(self, G: int) -> None:
        """Select the lanes per chain of the following launches (pdmp_set_group_lanes; 0 = the default).  Changes the order of
        the sums over coordinates: compare with a checker in ``ARITH_GROUP(G)`` of the same G."""
        _lib.check(_lib.lib().pdmp_set_group_lanes(self.handle(), int(G)))

    def reload_policy(self) -> None:
        """Re-read the PDMP_* environment overrides of the launch policy (pdmp_reload_policy): the policy of a handle is fixed when the
        handle is created; a caller that changes the environment afterwards (tests, tuning scripts) asks for the re-read."""
        if self._handle is not None:
            _lib.check(_lib.lib().pdmp_reload_policy(self._handle))

    def set_tape(self, exp_tape, uni_tape, nrm_tape):
        """Replay pre-recorded Exp(1) / U(0,1) / N(0,1) variates (arrays of shape ``(C, n_kind)``, row c = chain c,
        each kind in consumption order) instead of the Philox stream (pdmp_set_tape)."""
        arrs = [np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float64))) for a in (exp_tape, uni_tape, nrm_tape)]
        if len({a.shape[0] for a in arrs}) != 1:
            raise ArgumentError("the three tapes must hold the same number of chains")
        _lib.check(_lib.lib().pdmp_set_tape(self.handle(), arrs[0].shape[0], arrs[0].shape[1], arrs[1].shape[1], arrs[2].shape[1],
                                            C.c_void_p(arrs[0].ctypes.data), C.c_void_p(arrs[1].ctypes.data),
                                            C.c_void_p(arrs[2].ctypes.data)))

    def clear_tape(self):
        _lib.check(_lib.lib().pdmp_clear_tape(self.handle()))

    def close(self):
        if self._handle is not None:
            _lib.lib().pdmp_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):
        return (f"{self.name}(dim={self.dim}, target={self.target}, grid_size={self.grid_size}, tmax={self.tmax}, "
                f"refresh_rate={self.refresh_rate}, signed_bound={self.signed_bound}, adaptive={self.adaptive}, "
                f"AD_backend={self.AD_backend!r})")


class ZigZag(AbstractPDMP):
    sampler_id = ZIGZAG
    name = "ZigZag"

    def __init__(self, dim, target, *, grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=True,
                 signed_bound=True, adaptive=True, AD_backend="FiniteDiff", **kw):   # ZigZagSamplers.jl:58-60
        if signed_bound and not vectorized_bound:                                    # :75-78
            signed_bound = False
            warnings.warn("Signed bound is not compatible with non-vectorized bound for ZigZag, "
                          "switching to unsigned bound")
        super().__init__(dim, target, grid_size=grid_size, tmax=tmax, refresh_rate=refresh_rate,
                         vectorized_bound=vectorized_bound, signed_bound=signed_bound, adaptive=adaptive,
                         AD_backend=AD_backend, **kw)


def ZigZagAD(dim, target, *, refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=True, signed_bound=True,
             adaptive=True, AD_backend="ForwardDiff", **kw):                         # ZigZagSamplers.jl:118-126
    return ZigZag(dim, target, refresh_rate=refresh_rate, grid_size=grid_size, tmax=tmax,
                  vectorized_bound=vectorized_bound, signed_bound=signed_bound, adaptive=adaptive,
                  AD_backend=AD_backend, **kw)


class BPS(AbstractPDMP):
    sampler_id = BPS_ID
    name = "BPS"

    def __init__(self, dim, target, *, grid_size=10, tmax=1.0, refresh_rate=0.1, vectorized_bound=False,
                 signed_bound=True, adaptive=True, AD_backend="ForwardDiff", Gaussian_velocity=False, **kw):
        # BouncyParticleSamplers.jl:21-24; vectorized_bound forced false (:37)
        super().__init__(dim, target, grid_size=grid_size, tmax=tmax, refresh_rate=refresh_rate,
                         vectorized_bound=False, signed_bound=signed_bound, adaptive=adaptive,
                         AD_backend=AD_backend, gaussian_velocity=Gaussian_velocity, **kw)


def BPSAD(dim, target, *, refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=True, signed_bound=True,
          adaptive=True, AD_backend="ForwardDiff", **kw):                            # BouncyParticleSamplers.jl:86-94
    return BPS(dim, target, refresh_rate=refresh_rate, grid_size=grid_size, tmax=tmax, signed_bound=signed_bound,
               adaptive=adaptive, AD_backend=AD_backend, **kw)


class Boomerang(AbstractPDMP):
    sampler_id = BOOMERANG
    name = "Boomerang"

    def __init__(self, dim, target, *, grid_size=10, tmax=1.0, refresh_rate=0.1, vectorized_bound=False,
                 signed_bound=True, adaptive=True, AD_backend="FiniteDiff", **kw):    # BoomerangSamplers.jl:21-23
        super().__init__(dim, target, grid_size=grid_size, tmax=tmax, refresh_rate=refresh_rate,
                         vectorized_bound=False, signed_bound=signed_bound, adaptive=adaptive,
                         AD_backend=AD_backend, **kw)


def BoomerangAD(dim, target, *, refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=True, signed_bound=True,
                adaptive=True, AD_backend="ForwardDiff", **kw):                      # BoomerangSamplers.jl:79-87
    return Boomerang(dim, target, refresh_rate=refresh_rate, grid_size=grid_size, tmax=tmax,
                     signed_bound=signed_bound, adaptive=adaptive, AD_backend=AD_backend, **kw)


class ForwardECMC(AbstractPDMP):
    """ForwardECMC(dim, grad-U; grid_size=10, tmax=2.0, signed_bound=true, adaptive=true, ran_p=false, mix_p=0.5,
    switch=true, positive=true, AD_backend="ForwardDiff") (ForwardEventChainMonteCarlo.jl:290-344): the flow, rate and
    bound of BPS without refreshment and the event-chain velocity jump (:127-175; ``speed_factor != 1``: :177-218).
    ``normal=True`` is rejected: that variant of the reference cannot run (it takes the square root of a vector, :228)."""
    sampler_id = 3
    name = "ForwardECMC"

    def __init__(self, dim, target, *, grid_size=10, tmax=2.0, signed_bound=True, adaptive=True, ran_p=False, mix_p=0.5,
                 switch=True, positive=True, AD_backend="ForwardDiff", speed_factor=1.0, normal=False, **kw):
        if isinstance(dim, (int, np.integer)) and not isinstance(dim, bool) and 0 < dim < 2:      # :298-300
            raise ArgumentError(f"The dimension must be at least 2 to use the ForwardEventChain. Got dimension {dim}")
        if normal:      # the reference's `normal` variant takes sqrt of a vector (ForwardEventChainMonteCarlo.jl:228) and cannot run
            raise ArgumentError("ForwardECMC: normal = true is not implemented")
        super().__init__(dim, target, grid_size=grid_size, tmax=tmax, refresh_rate=0.0, vectorized_bound=False,
                         signed_bound=signed_bound, adaptive=adaptive, AD_backend=AD_backend, **kw)
        self.mix_p = 0.0 if self.dim == 2 else float(mix_p)                                         # :301-303
        self.ran_p, self.switch, self.positive = bool(ran_p), bool(switch), bool(positive)
        self.speed_factor = float(speed_factor)


def ForwardECMCAD(dim, target, *, grid_size=10, tmax=2.0, signed_bound=True, adaptive=True, AD_backend="ForwardDiff",
                  ran_p=False, mix_p=0.5, switch=True, positive=True, speed_factor=1.0, **kw):   # ForwardEventChainMonteCarlo.jl:366-377
    return ForwardECMC(dim, target, grid_size=grid_size, tmax=tmax, signed_bound=signed_bound, adaptive=adaptive,
                       ran_p=ran_p, mix_p=mix_p, switch=switch, positive=positive, AD_backend=AD_backend,
                       speed_factor=speed_factor, **kw)


class SpeedUpZigZag(AbstractPDMP):
    """SpeedUpZigZag(dim, grad-U; grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=true, signed_bound=true,
    adaptive=true, AD_backend="FiniteDiff") (SpeedUpZigZagSamplers.jl:56-59): the Zig-Zag loop with the closed-form
    non-linear flow (:71-79) and grad-U_effective = speed grad-U - grad speed, speed(x) = sqrt(1 + |x|^2) (:81-83).
    ``sample_from_skeleton``, the fused moment sums and the time-horizon cut all interpolate with this flow (sample.jl:499)."""
    sampler_id = 4
    name = "SpeedUpZigZag"

    def __init__(self, dim, target, *, grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=True,
                 signed_bound=True, adaptive=True, AD_backend="FiniteDiff", **kw):
        if signed_bound and not vectorized_bound:                                      # :65-68
            warnings.warn("Signed bound is not compatible with non-vectorized bound for ZigZag, switching to unsigned bound")
            signed_bound = False
        super().__init__(dim, target, grid_size=grid_size, tmax=tmax, refresh_rate=refresh_rate,
                         vectorized_bound=vectorized_bound, signed_bound=signed_bound, adaptive=adaptive,
                         AD_backend=AD_backend, **kw)


def SpeedUpZigZagAD(dim, target, *, refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=True, signed_bound=True,
                    adaptive=True, AD_backend="ForwardDiff", **kw):                   # SpeedUpZigZagSamplers.jl:120-130
    return SpeedUpZigZag(dim, target, refresh_rate=refresh_rate, grid_size=grid_size, tmax=tmax,
                         vectorized_bound=vectorized_bound, signed_bound=signed_bound, adaptive=adaptive,
                         AD_backend=AD_backend, **kw)


class RHMC(AbstractPDMP):
    """RHMC(dim, grad-U; mean_duration=nothing, refresh_rate=1.0, phi=pi/2, step_size=0.05, tmax=10.0, adaptive=false,
    AD_backend="Undefined") (RandomizedHamiltonianMonteCarlo.jl:48-57): Hamiltonian flow by velocity-Verlet (:97-130), a constant
    Poisson refresh clock (:133, the 2-point BoundBox of :208-218) and Horowitz refreshment v <- cos(phi) v + sin(phi) xi
    (:144-149).  ``sample_from_skeleton``, the fused moment sums and the time-horizon cut integrate the same flow from the last
    event (built-in targets, dim <= 256)."""
    sampler_id = 5
    name = "RHMC"

    def __init__(self, dim, target, *, mean_duration=None, refresh_rate=1.0, phi=math.pi / 2, step_size=0.05, tmax=10.0,
                 adaptive=False, AD_backend="Undefined", **kw):
        if mean_duration is not None:                                                    # :63-69
            md = float(mean_duration)
            if not math.isfinite(md) or md <= 0:
                raise ArgumentError(f"mean_duration must be finite and positive. Current value: {mean_duration}")
            refresh_rate = 1.0 / md
        refresh_rate = float(refresh_rate)
        if not math.isfinite(refresh_rate) or refresh_rate <= 0:                         # :71-74
            raise ArgumentError(f"refresh_rate must be finite and positive. Current value: {refresh_rate}")
        phi = float(phi)
        if not (0.0 < phi <= math.pi / 2):                                               # :76-79
            raise ArgumentError(f"phi must satisfy 0 < phi <= pi/2. Current value: {phi}")
        step_size = float(step_size)
        if not math.isfinite(step_size) or step_size <= 0:                               # :81-84
            raise ArgumentError(f"step_size must be finite and positive. Current value: {step_size}")
        tmax = float(tmax)
        if not math.isfinite(tmax) or tmax < 0:                                          # :86-89
            raise ArgumentError(f"tmax must be finite and non-negative. Current value: {tmax}")
        super().__init__(dim, target, grid_size=0, tmax=tmax, refresh_rate=refresh_rate, vectorized_bound=False,
                         signed_bound=False, adaptive=adaptive, AD_backend=AD_backend, **kw)      # :91-95
        self.step_size, self.phi = step_size, phi


def RHMCAD(dim, target, *, AD_backend="ForwardDiff", **kw):                              # RandomizedHamiltonianMonteCarlo.jl:178-182
    return RHMC(dim, target, AD_backend=AD_backend, **kw)


class StickyZigZag(AbstractPDMP):
    """StickyZigZag(dim, grad-U, kappa; refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=true, signed_bound=true,
    adaptive=true, AD_backend="FiniteDiff") (StickyZigZagSamplers.jl:59-61) with the sampling loop of src/StickySamplingLoop.jl:
    a coordinate that reaches 0 sticks there (its velocity is masked out of flows, rates and bounds) and thaws at rate
    ``kappa[i]``; accepted flips, sticking and thawing are all recorded events, ``PDMPHistory.is_active`` tells which
    coordinates move.  ``sample_from_skeleton`` dispatches to the reference's ``StickyPDMP`` methods (sample.jl:516-561, :588-596):
    the flow uses ``v_used = is_active ? v : 0``, so a frozen coordinate stays exactly 0 between its sticking and thawing events."""
    sampler_id = 6
    name = "StickyZigZag"

    def __init__(self, dim, target, kappa, *, refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=True,
                 signed_bound=True, adaptive=True, AD_backend="FiniteDiff", **kw):
        if signed_bound and not vectorized_bound:                                      # :70-73
            warnings.warn("Signed bound is not compatible with non-vectorized bound for ZigZag, switching to unsigned bound")
            signed_bound = False
        super().__init__(dim, target, grid_size=grid_size, tmax=tmax, refresh_rate=refresh_rate,
                         vectorized_bound=vectorized_bound, signed_bound=signed_bound, adaptive=adaptive,
                         AD_backend=AD_backend, **kw)
        kappa = np.asarray(kappa, dtype=np.float64).ravel()
        if kappa.size != self.dim:
            raise DimensionMismatch(f"kappa must have length dim ({self.dim})")
        self.kappa = kappa
        base = np.zeros(0) if self._params is None else np.asarray(self._params, dtype=np.float64).ravel()
        self._params = np.concatenate([base, kappa])                                   # kappa travels as the last dim target_params


def StickyZigZagAD(dim, target, kappa, *, refresh_rate=0.0, grid_size=10, tmax=2.0, vectorized_bound=True, signed_bound=True,
                   adaptive=True, AD_backend="ForwardDiff", **kw):                    # StickyZigZagSamplers.jl:116-126
    return StickyZigZag(dim, target, kappa, refresh_rate=refresh_rate, grid_size=grid_size, tmax=tmax,
                        vectorized_bound=vectorized_bound, signed_bound=signed_bound, adaptive=adaptive,
                        AD_backend=AD_backend, **kw)
