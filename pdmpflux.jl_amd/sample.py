"""``sample``, ``sample_skeleton``, ``sample_from_skeleton`` with the reference's signatures
(src/sample.jl:27-58, 253-311, 475-513), running on the HIP engine through the C ABI.

Single-chain calls take ``xinit``/``vinit`` vectors (or scalars in 1-D, sample.jl:289-311) and return
a ``PDMPHistory`` / a ``d x N`` matrix.  Ensemble calls take ``(C, d)`` arrays (row c = chain c) and
return dicts of ``(C, n_sk, ...)`` arrays; chain ``c`` uses the stream ``(seed, chain_id_offset + c)``.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _lib
from ._lib import ArgumentError, DimensionMismatch
from .history import PDMPHistory, history_from_ensemble
from .samplers import AbstractPDMP


def _vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _seed(seed):
    if seed is None:                       # Random.default_rng(), AbstractPDMP.jl:100
        return int.from_bytes(os.urandom(8), "little")
    return int(seed) & (2**64 - 1)


def _as_init(sampler: AbstractPDMP, xinit, vinit):
    scalar = np.isscalar(xinit) and np.isscalar(vinit)
    if scalar:                             # sample.jl:289-311
        if not (math.isfinite(xinit) and math.isfinite(vinit)):
            raise ArgumentError("initial values contain NaN, Inf, or -Inf.")   # :303-305
        xinit, vinit = [float(xinit)], [float(vinit)]
    x = np.ascontiguousarray(np.asarray(xinit, dtype=np.float64))
    v = np.ascontiguousarray(np.asarray(vinit, dtype=np.float64))
    single = x.ndim == 1
    if single:
        x, v = x[None, :], v[None, :] if v.ndim == 1 else v
    if x.ndim != 2 or v.ndim != 2 or x.shape[1] != sampler.dim or v.shape[1] != sampler.dim or x.shape != v.shape:
        raise DimensionMismatch(                                                 # AbstractPDMP.jl:96-98
            f"xinit and vinit must have the same dimension as pdmp.dim ({sampler.dim}). "
            f"Current dimensions: xinit ({x.shape[-1]}), vinit ({v.shape[-1]})")
    return x, v, single


def sample_skeleton_ensemble(sampler: AbstractPDMP, n_sk: int, xinit, vinit, *, seed=None,
                             chain_id_offset: int = 0, diagnostics: bool = True) -> dict:
    """Host-buffer ensemble call (pdmp_sample_skeleton).  Returns X, V (C, n_sk, d), t (C, n_sk) and,
    with ``diagnostics``, the PDMPHistory diagnostic arrays."""
    if n_sk <= 0:
        raise ArgumentError(f"n_sk must be positive. Current value: {n_sk}")     # sample.jl:262-264
    x, v, _ = _as_init(sampler, xinit, vinit)
    Cn, d = x.shape
    out = {"X": np.empty((Cn, n_sk, d)), "V": np.empty((Cn, n_sk, d)), "t": np.empty((Cn, n_sk))}
    diag = None
    if diagnostics:
        out.update(horizon=np.empty((Cn, n_sk)), ar=np.empty((Cn, n_sk)),
                   errored_bound=np.empty((Cn, n_sk), np.int32), error_value_ar=np.empty((Cn, n_sk, 5)),
                   rejected=np.empty((Cn, n_sk), np.int32), hitting_horizon=np.empty((Cn, n_sk), np.int32))
        diag = _lib.Diag(out["horizon"].ctypes.data, out["ar"].ctypes.data, out["errored_bound"].ctypes.data,
                         out["error_value_ar"].ctypes.data, out["rejected"].ctypes.data,
                         out["hitting_horizon"].ctypes.data, None)
    if sampler.sampler_id == 6:                              # StickyZigZag: PDMPHistory.is_active (Composites.jl:142)
        out["is_active"] = np.ones((Cn, n_sk, d), np.uint8)
        if diag is None:
            diag = _lib.Diag(None, None, None, None, None, None, None)
        diag.is_active = out["is_active"].ctypes.data
    L = _lib.lib()
    _lib.check(L.pdmp_sample_skeleton(sampler.handle(), Cn, int(n_sk), int(chain_id_offset), _vp(x), _vp(v),
                                      C.c_uint64(_seed(seed)), _vp(out["X"]), _vp(out["V"]), _vp(out["t"]),
                                      C.byref(diag) if diag is not None else None))
    return out


def sample_skeleton_until(sampler: AbstractPDMP, T: float, xinit, vinit, *, seed=None, verbose: bool = True,
                          init_capacity: int = 1024):
    """Time-horizon variant ``sample_skeleton(sampler, T::Float64, xinit, vinit; seed, init_capacity)``
    (src/sample.jl:323-439): advance every chain to time ``T``; the history ends with one closing point at
    exactly ``t == T`` obtained by the sampler's own deterministic flow from the last event before ``T`` (:385-420).

    The events before ``T`` are, by construction of the reference loop, the first events of the fixed-count
    skeleton for the same seed, so the engine's fixed-count kernel runs with a capacity that doubles
    (as ``_grow_history`` does, Composites.jl:172-191) until every chain has passed ``T``; the closing point is
    written on the device (``pdmp_close_at_time_device``: every sampler's flow, the Sticky mask, zeroed statistics)
    and the ragged histories are packed by ``pdmp_compact_history_device`` (``_trim_history``, :198-214).
    Returns a PDMPHistory for vector inputs, a list of PDMPHistory (one per chain, ragged lengths) for ``(C, d)`` inputs."""
    import torch
    from . import ensemble
    T = float(T)
    if not math.isfinite(T) or T < 0:
        raise ArgumentError(f"T must be finite and non-negative. Current value: {T}")      # sample.jl:335-337
    x, v, single = _as_init(sampler, xinit, vinit)
    if not torch.cuda.is_available():
        raise _lib.EngineError(_lib.ERR_RUNTIME, "no HIP device visible: the PDMP engine has no CPU fallback")
    dev = torch.device("cuda", sampler.device if sampler.device >= 0 else torch.cuda.current_device())
    res = ensemble.sample_skeleton_until_device(sampler, T, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=_seed(seed),
                                                diagnostics=True, init_capacity=init_capacity, compact=True)
    off = res["offsets"].cpu().numpy()
    host = {k: res[k].cpu().numpy() for k in ("X", "V", "t", "horizon", "ar", "errored_bound", "error_value_ar", "rejected",
                                              "hitting_horizon") + (("is_active",) if "is_active" in res else ())}
    hists = []
    for c in range(x.shape[0]):
        sl = slice(int(off[c]), int(off[c + 1]))
        hists.append(history_from_ensemble({k: a[None, sl] for k, a in host.items()}, 0))
    if single:
        sampler.state = hists[0]
        return hists[0]
    return hists


def sample_skeleton(sampler: AbstractPDMP, n_sk, xinit, vinit, *, seed=None, verbose: bool = True, init_capacity: int = 1024):
    """sample.jl:253-284 (``n_sk::Int``: fixed number of skeleton points) and :323-439 (``T::Float64``: time
    horizon) -- dispatch on the type of the second argument, as Julia does.  Vectors -> PDMPHistory; (C, d)
    arrays -> ensemble dict (fixed count) / list of PDMPHistory (time horizon)."""
    if isinstance(n_sk, float):
        return sample_skeleton_until(sampler, n_sk, xinit, vinit, seed=seed, verbose=verbose, init_capacity=init_capacity)
    if n_sk <= 0:
        raise ArgumentError(f"n_sk must be positive. Current value: {n_sk}")
    x, v, single = _as_init(sampler, xinit, vinit)
    out = sample_skeleton_ensemble(sampler, n_sk, x, v, seed=seed)
    if not single:
        return out
    hist = history_from_ensemble(out, 0)
    sampler.state = hist                       # sampler.state = state, sample.jl:281 (final state is the last column)
    return hist


def sample_from_skeleton_ensemble(sampler: AbstractPDMP, N: int, X, V, t, *, discard_vt: bool = True, dt: float = 0.0,
                                  is_active=None, n_prefix: int = 0, return_counts: bool = False):
    """X, V: (C, n_sk, d); t: (C, n_sk); is_active: (C, n_sk, d) uint8 / bool (PDMPHistory.is_active; required for StickyZigZag).
    Returns (C, N, rows) with rows = d or 2d+1.  ``dt > 0``: samples at ``j*dt`` (the ``dt::Float64`` method); ``n_prefix > 0``
    with ``dt``: the ``(N, dt)`` method on the first ``n_prefix`` columns.  ``return_counts``: also the per-chain number of
    samples ``floor(t_end/dt)`` (columns beyond it are zero)."""
    if N <= 0:
        raise ArgumentError(f"N must be positive. Current value: {N}")          # sample.jl:476-478
    X = np.ascontiguousarray(X, dtype=np.float64); V = np.ascontiguousarray(V, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    Cn, n_sk, d = X.shape
    if d != sampler.dim or V.shape != X.shape or t.shape != (Cn, n_sk):
        raise DimensionMismatch("history arrays do not match the sampler dimension")
    ia = None
    if is_active is not None:
        ia = np.ascontiguousarray(np.asarray(is_active).astype(np.uint8))
        if ia.shape != X.shape:
            raise DimensionMismatch("is_active does not match the history")
    rows = d if discard_vt else 2 * d + 1
    out = np.zeros((Cn, N, rows)) if return_counts else np.empty((Cn, N, rows))
    counts = np.zeros(Cn, np.int64) if return_counts else None
    L = _lib.lib()
    if n_prefix > 0:
        _lib.check(L.pdmp_sample_from_skeleton_prefix(sampler.handle(), Cn, n_sk, _vp(X), _vp(V), _vp(t), _vp(ia), int(n_prefix),
                                                      float(dt), int(N), int(bool(discard_vt)), _vp(out), _vp(counts)))
    elif dt > 0.0:
        _lib.check(L.pdmp_sample_from_skeleton_dt(sampler.handle(), Cn, n_sk, _vp(X), _vp(V), _vp(t), _vp(ia), int(N),
                                                  float(dt), int(bool(discard_vt)), _vp(out), _vp(counts)))
    else:
        _lib.check(L.pdmp_sample_from_skeleton(sampler.handle(), Cn, n_sk, _vp(X), _vp(V), _vp(t), _vp(ia), int(N),
                                               int(bool(discard_vt)), _vp(out)))
    return (out, counts) if return_counts else out


def _history_arrays(history):
    X = np.ascontiguousarray(history.X.T)[None]
    V = np.ascontiguousarray(history.V.T)[None]
    t = np.ascontiguousarray(history.t)[None]
    ia = np.ascontiguousarray(np.asarray(history.is_active).T.astype(np.uint8))[None]
    return X, V, t, ia


def sample_from_skeleton(sampler: AbstractPDMP, N, history, dt=None, *, discard_vt: bool = True):
    """``N::Int``: sample.jl:475-513 (N equispaced samples, dt = t[end]/N; StickyPDMP: :516-561).  ``dt::Float64``: :573-646
    (samples at j*dt, j = 1..floor(t[end]/dt)).  ``(N::Int, dt::Float64, history)``: :649-682 (the first N history columns,
    samples at dt:dt:t[N]) -- written ``sample_from_skeleton(sampler, N, dt, history)`` as in Julia, or with ``dt=`` as a
    keyword.  Dispatch on the types of the arguments, as Julia does.  ``history`` is a PDMPHistory (-> ``rows x N`` matrix,
    Fortran-ordered view) or, for the Int method, an ensemble dict (-> (C, N, rows))."""
    if isinstance(history, float) and not isinstance(dt, float) and dt is not None:      # Julia's positional order (sampler, N, dt, history)
        history, dt = dt, history
    if dt is not None:                                                                   # the (N, dt) method, :649-682
        if isinstance(history, dict):
            raise ArgumentError("the (N, dt) method takes one PDMPHistory")
        n_pre, dt = int(N), float(dt)
        if not (dt > 0.0) or not math.isfinite(dt):
            raise ArgumentError(f"dt must be positive and finite. Current value: {dt}")
        if n_pre < 1 or n_pre > len(history):
            raise ArgumentError(f"N must be in 1..length(history) = {len(history)}. Current value: {N}")   # BoundsError in Julia
        M = int(math.floor(history.t[n_pre - 1] / dt))                                   # length(dt:dt:t[end])
        rows = history.X.shape[0] if discard_vt else 2 * history.X.shape[0] + 1
        if M <= 0:
            return np.empty((rows, 0))
        X, V, t, ia = _history_arrays(history)
        return sample_from_skeleton_ensemble(sampler, M, X, V, t, discard_vt=discard_vt, dt=dt, is_active=ia, n_prefix=n_pre)[0].T
    if isinstance(N, float):
        dt = N
        if isinstance(history, dict):
            raise ArgumentError("the dt::Float64 method takes one PDMPHistory (sample counts differ per chain)")
        if not (dt > 0.0) or not math.isfinite(dt):
            raise ArgumentError(f"dt must be positive and finite. Current value: {dt}")
        n = int(math.floor(history.t[-1] / dt))                                  # n_skel = Int(floor(T_end / dt)), :608
        rows = history.X.shape[0] if discard_vt else 2 * history.X.shape[0] + 1
        if n <= 0:
            return np.empty((rows, 0))
        X, V, t, ia = _history_arrays(history)
        return sample_from_skeleton_ensemble(sampler, n, X, V, t, discard_vt=discard_vt, dt=dt, is_active=ia)[0].T
    if N <= 0:
        raise ArgumentError(f"N must be positive. Current value: {N}")
    if isinstance(history, dict):
        return sample_from_skeleton_ensemble(sampler, N, history["X"], history["V"], history["t"],
                                             discard_vt=discard_vt, is_active=history.get("is_active"))
    X, V, t, ia = _history_arrays(history)
    return sample_from_skeleton_ensemble(sampler, N, X, V, t, discard_vt=discard_vt, is_active=ia)[0].T


def sample(sampler: AbstractPDMP, N_sk: int, N_samples: int, xinit, vinit, *, seed=None, verbose: bool = True,
           discard_vt: bool = True):
    """sample.jl:27-41: sample_skeleton then sample_from_skeleton."""
    history = sample_skeleton(sampler, N_sk, xinit, vinit, seed=seed, verbose=verbose)
    return sample_from_skeleton(sampler, N_samples, history, discard_vt=discard_vt)
