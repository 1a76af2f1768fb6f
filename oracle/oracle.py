"""ctypes binding of the CPU oracle (oracle/pdmp_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never from the product package.  PARITY UNPINNED: see the
header of ``pdmp_oracle.h``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpdmp_oracle.so")

ZIGZAG, BPS, BOOMERANG, FECMC, SUZZ, RHMC, STICKY = 0, 1, 2, 3, 4, 5, 6
T_GAUSS_ISO, T_GAUSS_DIAG, T_BANANA, T_GAUSS_DENSE = 0, 1, 2, 3
T_AD_BANANA, T_AD_FUNNEL, T_AD_RIDGED = 16, 17, 18      # grad-U by forward mode over U (ad_potentials.c)
DERIV_FORWARD, DERIV_FINITE_DIFF = 0, 1
ARITH_REFERENCE, ARITH_WAVE64, ARITH_GROUP = 0, 1, 2


class _Cfg(C.Structure):
    _fields_ = [
        ("sampler", C.c_int32), ("dim", C.c_int32), ("grid_size", C.c_int32),
        ("vectorized_bound", C.c_int32), ("signed_bound", C.c_int32), ("adaptive", C.c_int32),
        ("gaussian_velocity", C.c_int32), ("deriv_mode", C.c_int32), ("target_id", C.c_int32),
        ("arith", C.c_int32), ("tmax", C.c_double), ("refresh_rate", C.c_double),
        ("target_params", C.POINTER(C.c_double)), ("n_target_params", C.c_int64),
        ("max_iter_per_event", C.c_int64), ("group_lanes", C.c_int32),
        ("tape_exp", C.POINTER(C.c_double)), ("tape_uni", C.POINTER(C.c_double)), ("tape_nrm", C.POINTER(C.c_double)),
        ("tape_n_exp", C.c_int64), ("tape_n_uni", C.c_int64), ("tape_n_nrm", C.c_int64),
        ("fe_mix_p", C.c_double), ("fe_ran_p", C.c_int32), ("fe_switch", C.c_int32), ("fe_positive", C.c_int32),
        ("no_identities", C.c_int32), ("rh_step_size", C.c_double), ("rh_phi", C.c_double),
        ("fe_speed_factor", C.c_double), ("is_active_out", C.c_void_p),
        ("julia_rng", C.c_int32), ("pad0_", C.c_int32),
        ("jl_tape_exp", C.POINTER(C.c_double)), ("jl_tape_uni", C.POINTER(C.c_double)), ("jl_tape_nrm", C.POINTER(C.c_double)),
        ("jl_cap_exp", C.c_int64), ("jl_cap_uni", C.c_int64), ("jl_cap_nrm", C.c_int64), ("jl_count", C.POINTER(C.c_int64)),
        ("rng_version", C.c_int32), ("pad1_", C.c_int32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pdmp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.orc_last_error.restype = C.c_char_p
        L.orc_normalize_config.argtypes = [C.POINTER(_Cfg)]
        L.orc_sample_skeleton.argtypes = [C.POINTER(_Cfg), C.c_int64, C.c_int64, C.c_int64, dp, dp, C.c_uint64,
                                          dp, dp, dp, dp, dp, ip, dp, ip, ip, C.c_int]
        L.orc_sample_from_skeleton.argtypes = [C.POINTER(_Cfg), C.c_int64, C.c_int64, dp, dp, dp, C.c_int64,
                                               C.c_int, dp, C.c_int]
        L.orc_sample_from_skeleton_dt.argtypes = [C.POINTER(_Cfg), C.c_int64, C.c_int64, dp, dp, dp, C.c_int64,
                                                  C.c_double, C.c_int, dp, C.c_int]
        L.orc_sample_from_skeleton_ex.argtypes = [C.POINTER(_Cfg), C.c_int, C.c_int64, C.c_int64, dp, dp, dp, C.c_void_p, C.c_void_p,
                                                  C.c_int64, C.c_int64, C.c_double, C.c_int, dp, C.c_void_p, C.c_int]
        L.orc_sample_skeleton_until.argtypes = [C.POINTER(_Cfg), C.c_int64, C.c_int64, C.c_int64, dp, dp, C.c_uint64, C.c_double,
                                                dp, dp, dp, dp, dp, ip, dp, ip, ip, C.c_void_p, C.c_int]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_philox4x32_r.argtypes = [C.c_int] + [C.POINTER(C.c_uint32)] * 3
        L.orc_set_arith_version.argtypes = [C.c_int]
        L.orc_exp.argtypes = [C.c_double]; L.orc_exp.restype = C.c_double
        L.orc_log.argtypes = [C.c_double]; L.orc_log.restype = C.c_double
        L.orc_sincos.argtypes = [C.c_double, dp, dp]
        L.orc_sincos_2pi.argtypes = [C.c_double, dp, dp]
        L.orc_rng_dump.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int64, dp]
        L.orc_init_states.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int, dp, dp]
        L.orc_init_states.restype = None
        L.orc_upper_bound.argtypes = [C.POINTER(_Cfg), dp, dp, C.c_double, dp, dp, dp]
        L.orc_upper_bound.restype = C.c_double
        L.orc_next_event.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, dp, dp]
        L.orc_grad.argtypes = [C.POINTER(_Cfg), dp, dp]
        L.orc_potential.argtypes = [C.POINTER(_Cfg), dp]; L.orc_potential.restype = C.c_double
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


class OracleError(ValueError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


@dataclass
class Config:
    """Mirror of the reference constructor kwargs + target descriptor."""
    sampler: int = ZIGZAG
    dim: int = 1
    grid_size: int = 10
    tmax: float = 2.0
    refresh_rate: float = 0.0
    vectorized_bound: bool = True
    signed_bound: bool = True
    adaptive: bool = True
    gaussian_velocity: bool = False
    deriv_mode: int = DERIV_FORWARD
    target_id: int = T_GAUSS_ISO
    target_params: np.ndarray | None = None
    arith: int = ARITH_REFERENCE
    max_iter_per_event: int = 0
    group_lanes: int = 1
    tape: tuple | None = None          # (exp, uni, nrm) arrays of shape (C, n_kind): replay instead of Philox
    mix_p: float = 0.5                 # ForwardECMC keywords (ForwardEventChainMonteCarlo.jl:290-292)
    ran_p: bool = False
    switch: bool = True
    positive: bool = True
    step_size: float = 0.05            # RHMC keywords (RandomizedHamiltonianMonteCarlo.jl:48-57)
    phi: float = 1.5707963267948966
    speed_factor: float = 1.0          # ForwardECMC
    no_identities: bool = False        # GROUP order without the engine order's three algebraic identities (checker-side comparisons only)
    is_active_out: int | None = None   # StickyZigZag: address of a (C, n_sk, d) uint8 array receiving PDMPHistory.is_active
    julia_rng: int = 0                 # 0 Philox / tapes; JL_SEED_LIMBS (1) / JL_SEED_SHA256 (2): chain c draws from Julia's MersenneTwister(seed + c)
    rng_version: int = 0               # arithmetic contract of the stream and of log / exp / sin / cos: 0 = the default (DEFAULT_RNG_VERSION = 2), 1 = the round 1-4 contract
    _keep: list = field(default_factory=list, repr=False)

    def c(self) -> _Cfg:
        p = None
        n = 0
        if self.target_params is not None:
            arr = np.ascontiguousarray(np.asarray(self.target_params, dtype=np.float64).ravel())
            self._keep = [arr]
            p, n = _dp(arr), arr.size
        te = tu = tn = None
        ne = nu = nn = 0
        if self.tape is not None:
            arrs = [np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float64))) for a in self.tape]
            self._keep = list(self._keep) + arrs
            (te, tu, tn), (ne, nu, nn) = [_dp(a) for a in arrs], [a.shape[1] for a in arrs]
        return _Cfg(self.sampler, self.dim, self.grid_size, int(self.vectorized_bound), int(self.signed_bound),
                    int(self.adaptive), int(self.gaussian_velocity), self.deriv_mode, self.target_id, self.arith,
                    float(self.tmax), float(self.refresh_rate), p, n, self.max_iter_per_event, self.group_lanes,
                    te, tu, tn, ne, nu, nn, float(self.mix_p), int(self.ran_p), int(self.switch), int(self.positive), int(self.no_identities),
                    float(self.step_size), float(self.phi), float(self.speed_factor), self.is_active_out,
                    int(self.julia_rng), 0, None, None, None, 0, 0, 0, None, int(self.rng_version), 0)


def _check(rc):
    if rc != 0:
        raise OracleError(rc, lib().orc_last_error().decode())


def normalize(cfg: Config) -> Config:
    c = cfg.c()
    _check(lib().orc_normalize_config(C.byref(c)))
    out = Config(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__ if k != "_keep"})
    out.tmax, out.adaptive = c.tmax, bool(c.adaptive)
    out.signed_bound, out.vectorized_bound = bool(c.signed_bound), bool(c.vectorized_bound)
    return out


def sample_skeleton(cfg: Config, n_sk: int, xinit, vinit, seed: int, chain_id_offset: int = 0,
                    diagnostics: bool = True, n_threads: int = 0, out: dict | None = None) -> dict:
    """xinit, vinit: (C, d) arrays (row c = chain c; memory layout equals the reference's d x C
    column-major).  Returns dict of arrays: X, V (C, n_sk, d), t (C, n_sk), + diagnostics."""
    x = np.ascontiguousarray(np.atleast_2d(np.asarray(xinit, dtype=np.float64)))
    v = np.ascontiguousarray(np.atleast_2d(np.asarray(vinit, dtype=np.float64)))
    if x.shape != v.shape or x.shape[1] != cfg.dim:
        raise OracleError(-2, f"xinit and vinit must have the same dimension as dim ({cfg.dim})")
    Cn, d = x.shape
    n = max(int(n_sk), 0)
    if out is not None:                 # caller-owned (already touched) buffers: bench.py's cpu_baseline times the loop, not page faults
        assert not diagnostics and all(out[k].flags.c_contiguous and out[k].shape[:2] == (Cn, n) for k in ("X", "V", "t"))
    else:
        out = {"X": np.empty((Cn, n, d)), "V": np.empty((Cn, n, d)), "t": np.empty((Cn, n))}
    if diagnostics:
        out.update(horizon=np.empty((Cn, n)), ar=np.empty((Cn, n)),
                   errored_bound=np.empty((Cn, n), np.int32), error_value_ar=np.empty((Cn, n, 5)),
                   rejected=np.empty((Cn, n), np.int32), hitting_horizon=np.empty((Cn, n), np.int32))
    g = out.get
    c = cfg.c()
    if cfg.sampler == STICKY:           # PDMPHistory.is_active (Composites.jl:142)
        out["is_active"] = np.empty((Cn, n, d), np.uint8)
        c.is_active_out = out["is_active"].ctypes.data
    _check(lib().orc_sample_skeleton(C.byref(c), Cn, int(n_sk), int(chain_id_offset), _dp(x), _dp(v),
                                     C.c_uint64(seed & (2**64 - 1)), _dp(out["X"]), _dp(out["V"]), _dp(out["t"]),
                                     _dp(g("horizon")), _dp(g("ar")), _ip(g("errored_bound")),
                                     _dp(g("error_value_ar")), _ip(g("rejected")), _ip(g("hitting_horizon")),
                                     n_threads))
    return out


JL_SEED_LIMBS, JL_SEED_SHA256 = 1, 2
JL_SCALAR_RANDN = 4        # + 4: vectors of normals as scalar draws (an AbstractRNG wrapped around the MersenneTwister); default: randn!'s bulk method from 13 elements on


def sample_skeleton_julia(cfg: Config, n_sk: int, xinit, vinit, seed: int, seeding: int = JL_SEED_SHA256, tape_capacity=None,
                          diagnostics: bool = True) -> dict:
    """``sample_skeleton`` with the reference's own random source restated (oracle/julia_rng.c): chain c draws from Julia's
    ``MersenneTwister(seed + c)`` -- ``randexp`` / ``rand`` / ``randn`` in the reference's program order, ONE stream for the three kinds --
    seeded as Julia 1.11 (``JL_SEED_SHA256``) or as 1.0 - 1.10 (``JL_SEED_LIMBS``).  Returns the skeleton plus ``tape`` = (exp, uni, nrm)
    arrays ``(C, n_kind)`` holding what every chain consumed (zero-padded) and ``tape_counts`` ``(C, 3)``: replayed through ``Config(tape=...)``
    or the engine's ``set_tape`` they reproduce the run draw for draw.  Unverified against a running Julia (none in this pipeline)."""
    x = np.ascontiguousarray(np.atleast_2d(np.asarray(xinit, dtype=np.float64)))
    Cn, d = x.shape
    n = int(n_sk)
    cap = tape_capacity or (64 * n + 64, 64 * n + 64, (n + 8) * max(2 * d, 8) if cfg.sampler != ZIGZAG else 8)
    te, tu, tn = np.zeros((Cn, cap[0])), np.zeros((Cn, cap[1])), np.zeros((Cn, cap[2]))
    counts = np.zeros((Cn, 3), dtype=np.int64)
    jcfg = Config(**{**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__ if k != "_keep"}, "julia_rng": int(seeding), "tape": None})
    v = np.ascontiguousarray(np.atleast_2d(np.asarray(vinit, dtype=np.float64)))
    out = {"X": np.empty((Cn, n, d)), "V": np.empty((Cn, n, d)), "t": np.empty((Cn, n))}
    if diagnostics:
        out.update(horizon=np.empty((Cn, n)), ar=np.empty((Cn, n)), errored_bound=np.empty((Cn, n), np.int32),
                   error_value_ar=np.empty((Cn, n, 5)), rejected=np.empty((Cn, n), np.int32), hitting_horizon=np.empty((Cn, n), np.int32))
    g = out.get
    c = jcfg.c()
    c.jl_tape_exp, c.jl_tape_uni, c.jl_tape_nrm = _dp(te), _dp(tu), _dp(tn)
    c.jl_cap_exp, c.jl_cap_uni, c.jl_cap_nrm = cap
    c.jl_count = counts.ctypes.data_as(C.POINTER(C.c_int64))
    if cfg.sampler == STICKY:
        out["is_active"] = np.empty((Cn, n, d), np.uint8)
        c.is_active_out = out["is_active"].ctypes.data
    _check(lib().orc_sample_skeleton(C.byref(c), Cn, n, 0, _dp(x), _dp(v), C.c_uint64(seed & (2**64 - 1)), _dp(out["X"]), _dp(out["V"]),
                                     _dp(out["t"]), _dp(g("horizon")), _dp(g("ar")), _ip(g("errored_bound")), _dp(g("error_value_ar")),
                                     _ip(g("rejected")), _ip(g("hitting_horizon")), 0))
    out["tape"] = (te, tu, tn)
    out["tape_counts"] = counts
    return out


def julia_draws(seed: int, n: int, kind: str = "rand", seeding: int = JL_SEED_LIMBS) -> np.ndarray:
    """First n draws of ``rand`` / ``randn`` / ``randexp`` from Julia's ``MersenneTwister(seed)`` as restated in oracle/julia_rng.c."""
    L = lib()
    L.orc_julia_draws.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_double)]
    out = np.empty(n)
    L.orc_julia_draws(int(seed), int(seeding), {"rand": 0, "randexp": 1, "randn": 2}[kind], n, _dp(out))
    return out


def julia_randn_fill(seed: int, n: int, scalar: bool = False, seeding: int = JL_SEED_LIMBS) -> np.ndarray:
    """``randn!(MersenneTwister(seed), zeros(n))`` as restated (oracle/julia_rng.c: jl_randn_fill): the bulk method of a plain MersenneTwister
    from 13 elements on, or -- ``scalar`` -- n scalar draws in index order."""
    L = lib()
    L.orc_julia_randn_fill.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    out = np.empty(n)
    L.orc_julia_randn_fill(int(seed), int(seeding), int(n), int(bool(scalar)), _dp(out))
    return out


def sample_from_skeleton(cfg: Config, N: int, X, V, t, discard_vt: bool = True, n_threads: int = 0, dt: float = 0.0,
                         is_active=None, lens=None, n_prefix: int = 0, return_counts: bool = False):
    """The reference's three sample_from_skeleton methods: ``N::Int`` (dt == 0; sample.jl:475-513, StickyPDMP :516-561),
    ``dt::Float64`` (dt > 0; :573-646), ``(N, dt)`` (n_prefix > 0 and dt > 0; :649-682: N = capacity of samples per chain)."""
    X = np.ascontiguousarray(X, dtype=np.float64); V = np.ascontiguousarray(V, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    if X.ndim == 2:
        X, V, t = X[None], V[None], t[None]
        if is_active is not None:
            is_active = np.asarray(is_active)[None]
    Cn, n_sk, d = X.shape
    rows = d if discard_vt else 2 * d + 1
    out = np.zeros((Cn, max(int(N), 0), rows)) if return_counts else np.empty((Cn, max(int(N), 0), rows))
    ia = None if is_active is None else np.ascontiguousarray(np.asarray(is_active).astype(np.uint8))
    ln = None if lens is None else np.ascontiguousarray(np.asarray(lens, dtype=np.int64))
    cnt = np.zeros(Cn, np.int64) if return_counts else None
    method = 2 if n_prefix > 0 else (1 if dt > 0.0 else 0)
    c = cfg.c()
    vp = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    _check(lib().orc_sample_from_skeleton_ex(C.byref(c), method, Cn, n_sk, _dp(X), _dp(V), _dp(t), vp(ia), vp(ln), int(n_prefix), int(N),
                                             float(dt), int(discard_vt), _dp(out), vp(cnt), n_threads))
    return (out, cnt) if return_counts else out


def sample_skeleton_until(cfg: Config, T: float, xinit, vinit, seed: int, chain_id_offset: int = 0, init_capacity: int = 1024,
                          n_threads: int = 0) -> dict:
    """sample_skeleton(sampler, T::Float64, xinit, vinit; seed, init_capacity) (sample.jl:323-439) for C chains: capacity doubles
    until every chain has reached T (_grow_history, Composites.jl:172-191).  Returns arrays shaped (C, capacity, ...) and ``count``
    (C,): chain c's history is columns [0, count[c])."""
    x = np.ascontiguousarray(np.atleast_2d(np.asarray(xinit, dtype=np.float64)))
    v = np.ascontiguousarray(np.atleast_2d(np.asarray(vinit, dtype=np.float64)))
    Cn, d = x.shape
    cap = max(2, int(init_capacity))
    while True:
        out = {"X": np.zeros((Cn, cap, d)), "V": np.zeros((Cn, cap, d)), "t": np.zeros((Cn, cap)),
               "horizon": np.zeros((Cn, cap)), "ar": np.zeros((Cn, cap)), "errored_bound": np.zeros((Cn, cap), np.int32),
               "error_value_ar": np.zeros((Cn, cap, 5)), "rejected": np.zeros((Cn, cap), np.int32),
               "hitting_horizon": np.zeros((Cn, cap), np.int32)}
        c = cfg.c()
        if cfg.sampler == STICKY:
            out["is_active"] = np.ones((Cn, cap, d), np.uint8)
            c.is_active_out = out["is_active"].ctypes.data
        count = np.zeros(Cn, np.int64)
        _check(lib().orc_sample_skeleton_until(C.byref(c), Cn, cap, int(chain_id_offset), _dp(x), _dp(v), C.c_uint64(seed & (2**64 - 1)),
                                               float(T), _dp(out["X"]), _dp(out["V"]), _dp(out["t"]), _dp(out["horizon"]), _dp(out["ar"]),
                                               _ip(out["errored_bound"]), _dp(out["error_value_ar"]), _ip(out["rejected"]),
                                               _ip(out["hitting_horizon"]), C.c_void_p(count.ctypes.data), n_threads))
        if (count > 0).all():
            break
        cap *= 2
    out["count"] = count
    return out


DEFAULT_RNG_VERSION = 2


def _ver(v: int):
    """The cfg-less helpers below run in the arithmetic contract `v` (0 = the default); thread-local in the C library."""
    lib().orc_set_arith_version(int(v) if v else DEFAULT_RNG_VERSION)


def philox(ctr, key, rounds: int = 10):
    a = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_r(int(rounds), a, k, o)
    return list(o)


def log(x: float, rng_version: int = 0) -> float:
    _ver(rng_version)
    return lib().orc_log(x)


def exp(x: float, rng_version: int = 0) -> float:
    _ver(rng_version)
    return lib().orc_exp(x)


def sincos(x: float, rng_version: int = 0):
    _ver(rng_version)
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(x, C.byref(s), C.byref(c))
    return s.value, c.value


def sincos_2pi(u: float, rng_version: int = 0):
    _ver(rng_version)
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos_2pi(u, C.byref(s), C.byref(c))
    return s.value, c.value


def rng_dump(seed: int, chain: int, kind: int, n: int, rng_version: int = 0) -> np.ndarray:
    _ver(rng_version)
    out = np.empty(n)
    lib().orc_rng_dump(C.c_uint64(seed), chain, kind, n, _dp(out))
    return out


V_RADEMACHER, V_GAUSSIAN, V_SPHERE = 0, 1, 2


def init_states(seed: int, C_: int, d: int, *, chain_id_offset: int = 0, x_mean: float = 0.0, x_scale: float = 1.0,
                v_kind: int = V_RADEMACHER, rng_version: int = 0):
    """Synthetic initial states of chains ``chain_id_offset .. +C`` from stream 1 of each chain's Philox key;
    the checker's side of pdmp_init_states_device.  Returns ``(x, v)``, each ``(C, d)``."""
    x, v = np.empty((C_, d)), np.empty((C_, d))
    _ver(rng_version)
    lib().orc_init_states(C.c_uint64(seed), int(chain_id_offset), int(C_), int(d), float(x_mean), float(x_scale), int(v_kind), _dp(x), _dp(v))
    return x, v


def upper_bound(cfg: Config, x, v, horizon: float):
    x = np.ascontiguousarray(x, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
    n = cfg.grid_size if cfg.grid_size > 0 else 2
    grid, box, cum = np.empty(n), np.empty(n - 1), np.empty(n)
    c = cfg.c()
    step = lib().orc_upper_bound(C.byref(c), _dp(x), _dp(v), horizon, _dp(grid), _dp(box), _dp(cum))
    return grid, box, cum, step


def next_event(grid, box, cum, step, E):
    tp, lb = C.c_double(), C.c_double()
    lib().orc_next_event(len(grid), _dp(grid), _dp(box), _dp(cum), step, E, C.byref(tp), C.byref(lb))
    return tp.value, lb.value


def grad(cfg: Config, x):
    x = np.ascontiguousarray(x, dtype=np.float64); g = np.empty_like(x)
    c = cfg.c()
    lib().orc_grad(C.byref(c), _dp(x), _dp(g))
    return g


def potential(cfg: Config, x) -> float:
    x = np.ascontiguousarray(x, dtype=np.float64)
    c = cfg.c()
    return lib().orc_potential(C.byref(c), _dp(x))


def num_threads() -> int:
    return lib().orc_num_threads()
