"""Host-side mirror of the reference API and the C-ABI boundary, without a GPU: constructor defaults and
normalisations, error types, that the library loads and exports every symbol include/pdmp_hip.h declares,
that the product fails loudly when no device is visible, and that it never routes through the oracle."""
import ctypes as C
import os
import re
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constructor_defaults_match_reference(engine):
    """SURVEY.md 3.6 / ZigZagSamplers.jl:58-60,118-119, BouncyParticleSamplers.jl:21-24,86-93, BoomerangSamplers.jl:21-23,79-86."""
    P = engine
    t = P.GaussianIso()
    rows = [
        (P.ZigZag(3, t), 10, 2.0, 0.0, True, True, True, "FiniteDiff"),
        (P.ZigZagAD(3, t), 10, 2.0, 0.0, True, True, True, "ForwardDiff"),
        (P.BPS(3, t), 10, 1.0, 0.1, False, True, True, "ForwardDiff"),
        (P.BPSAD(3, t), 10, 2.0, 0.0, False, True, True, "ForwardDiff"),
        (P.Boomerang(3, t), 10, 1.0, 0.1, False, True, True, "FiniteDiff"),
        (P.BoomerangAD(3, t), 10, 2.0, 0.0, False, True, True, "ForwardDiff"),
    ]
    for s, grid, tmax, refresh, vect, signed, adaptive, ad in rows:
        assert (s.grid_size, s.tmax, s.refresh_rate, s.vectorized_bound, s.signed_bound, s.adaptive, s.AD_backend) == \
               (grid, tmax, refresh, vect, signed, adaptive, ad), s
    assert P.ZigZag(3, t).deriv_mode == 1 and P.ZigZagAD(3, t).deriv_mode == 0
    for ab in ("Zygote", "ReverseDiff", "Enzyme", "PolyesterForwardDiff"):      # every AD backend = exact slope
        assert P.ZigZagAD(3, t, AD_backend=ab).deriv_mode == 0
    for ab in ("Undefined", "", None):                                          # AbstractPDMP.jl:18-28
        assert P.ZigZag(3, t, AD_backend=ab).deriv_mode == 1


def test_constructor_normalisations(engine):
    P = engine
    s = P.ZigZagAD(2, P.GaussianIso(), tmax=0, adaptive=False)                 # ZigZagSamplers.jl:73
    assert s.tmax == 1.0 and s.adaptive
    with warnings.catch_warnings(record=True) as w:                            # :75-78
        warnings.simplefilter("always")
        s = P.ZigZagAD(2, P.GaussianIso(), vectorized_bound=False)
    assert not s.signed_bound and any("Signed bound" in str(x.message) for x in w)
    assert not P.BPS(2, P.GaussianIso(), vectorized_bound=True).vectorized_bound   # BouncyParticleSamplers.jl:37
    assert P.BPS(2, P.GaussianIso(), Gaussian_velocity=True).gaussian_velocity


def test_error_conventions(engine):
    """test/test_error_handling.jl:6-20,41-55; test_stability.jl:130-150."""
    P = engine
    t = P.GaussianIso()
    for bad in (0, -1):
        with pytest.raises(P.ArgumentError):
            P.ZigZagAD(bad, t)
    with pytest.raises(P.ArgumentError):
        P.ZigZagAD(2, t, grid_size=-1)
    with pytest.raises(P.ArgumentError):
        P.ZigZagAD(2, t, AD_backend="NoSuchBackend")
    with pytest.raises(TypeError):                                             # MethodError for U = nothing
        P.ZigZagAD(2, None)
    s = P.ZigZagAD(2, t)
    with pytest.raises(P.ArgumentError):
        P.sample_skeleton(s, 0, np.zeros(2), np.ones(2), seed=1)               # sample.jl:262-264
    with pytest.raises(P.ArgumentError):
        P.sample_skeleton(s, -5, np.zeros(2), np.ones(2), seed=1)
    with pytest.raises(P.DimensionMismatch):
        P.sample_skeleton(s, 10, np.zeros(3), np.ones(3), seed=1)              # AbstractPDMP.jl:96-98
    with pytest.raises(P.DimensionMismatch):
        P.sample_skeleton(s, 10, np.zeros(2), np.ones(3), seed=1)
    for badT in (-1.0, float("inf"), float("nan")):                            # sample.jl:335-337
        with pytest.raises(P.ArgumentError):
            P.sample_skeleton(s, badT, np.zeros(2), np.ones(2), seed=1)
    s1 = P.ZigZagAD(1, t)
    for x0, v0 in ((float("nan"), 1.0), (0.0, float("inf")), (float("-inf"), 1.0)):
        with pytest.raises(P.ArgumentError):
            P.sample_skeleton(s1, 10, x0, v0, seed=1)                          # sample.jl:303-305
    h = P.PDMPHistory(np.zeros((1, 3)), np.ones((1, 3)), np.array([0, .5, 1.]), np.ones((1, 3), bool), np.zeros(3),
                      np.zeros(3), np.zeros(3, np.int32), np.zeros((5, 3)), np.zeros(3, np.int32), np.zeros(3, np.int32))
    with pytest.raises(P.ArgumentError):
        P.sample_from_skeleton(s1, 0, h)                                       # sample.jl:476-478
    with pytest.raises(P.DimensionMismatch):
        P.GaussianDiag([0.0, 0.0], [1.0, 1.0]).params(3)


def test_history_accessors(engine):
    """Composites.jl:225-234: .x / .v are copies of the columns."""
    P = engine
    X = np.arange(6.0).reshape(2, 3)
    h = P.PDMPHistory(X, X + 10, np.arange(3.0), np.ones((2, 3), bool), np.zeros(3), np.zeros(3), np.zeros(3, np.int32),
                      np.zeros((5, 3)), np.zeros(3, np.int32), np.zeros(3, np.int32))
    assert len(h) == 3 and len(h.x) == 3 and np.array_equal(h.x[1], X[:, 1]) and np.array_equal(h.v[2], X[:, 2] + 10)
    h.x[0][0] = 99.0
    assert h.X[0, 0] == 0.0


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pdmp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pdmp_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(engine):
    from pdmpflux.jl_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = C.CDLL(_lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 15
    for sym in decl:
        assert hasattr(L, sym), f"{sym} declared in include/pdmp_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == decl, "python binding table and header disagree"
    assert _lib.lib().pdmp_abi_version() == 11


def test_fails_loudly_without_gpu(engine):
    """No CPU fallback: without a visible device the product path raises (on the GPU box this test is skipped)."""
    from pdmpflux.jl_amd import _lib
    if _lib.lib().pdmp_device_count() > 0:
        pytest.skip("a GPU is visible")
    s = engine.ZigZagAD(2, engine.GaussianIso())
    with pytest.raises(engine.EngineError, match="no HIP device"):
        engine.sample_skeleton(s, 10, np.zeros(2), np.ones(2), seed=1)
    cfg = s._config()
    cfg.grid_size = 1                                                          # argument checks come before the device check
    h = C.c_void_p()
    assert _lib.lib().pdmp_create(C.byref(cfg), C.byref(h)) == _lib.ERR_UNSUPPORTED
    assert b"grid_size" in _lib.lib().pdmp_last_error()
    cfg = s._config(); cfg.dim = 0
    assert _lib.lib().pdmp_create(C.byref(cfg), C.byref(h)) == _lib.ERR_ARGUMENT


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import, link or call oracle/."""
    pkg = os.path.join(ROOT, "pdmpflux.jl_amd")
    banned = ("import oracle", "from oracle", "pdmp_oracle", "libpdmp_oracle", "orc_sample", "orc_upper", "oracle/")
    n = 0
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                n += 1
                for b in banned:
                    assert b not in txt, f"{f} uses {b!r}"
    assert n >= 10
    shim = open(os.path.join(ROOT, "pdmpflux", "__init__.py")).read()
    assert "oracle" not in shim
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("from oracle import") == 1 and bench.index("def cpu_baseline") < bench.index("from oracle import") < bench.index("def main")


def test_shard_ranges(engine):
    from pdmpflux.jl_amd.ensemble import shard_range
    for C_total, world in ((100000, 8), (10, 3), (7, 8), (0, 4), (5, 1)):
        parts = [shard_range(C_total, r, world) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == C_total
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1


def test_multi_device_partition_equals_shard_range(engine):
    """pdmp_multi_shard_range (the partition pdmp_multi_sample_skeleton uses inside the library) = ensemble.shard_range (the
    partition of the one-process-per-GPU driver): a chain lands on the same rank either way."""
    from pdmpflux.jl_amd import _lib
    from pdmpflux.jl_amd.ensemble import shard_range
    L = _lib.lib()
    for C_total, world in ((100000, 8), (100000, 3), (10, 3), (7, 8), (0, 4), (5, 1), (12500, 7)):
        for r in range(world):
            lo, hi = C.c_int64(), C.c_int64()
            assert L.pdmp_multi_shard_range(C_total, r, world, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == shard_range(C_total, r, world)
    lo, hi = C.c_int64(), C.c_int64()
    assert L.pdmp_multi_shard_range(10, 3, 3, C.byref(lo), C.byref(hi)) == _lib.ERR_ARGUMENT


def test_sticky_kappa_length_is_checked(engine):
    with pytest.raises(engine.DimensionMismatch):
        engine.StickyZigZag(3, engine.GaussianIso(), [0.5, 0.5])
