"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on identical inputs and
identical Philox streams.  The oracle runs in its WAVE64 summation order, so the comparison is
BIT-EXACT for every field of PDMPHistory (stronger than the 1e-5 relative bound of the north star;
the tolerance test below states that bound explicitly against the reference-order oracle)."""
import ctypes as C
import math

import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu


def run_pair(P, O, kind, dim, target, C_, n_sk, seed=11, offset=0, **kw):
    s, cfg = make_pair(P, O, kind, dim, target, **kw)
    x, v = make_init(kind, dim, C_, target=target)
    got = P.sample_skeleton_ensemble(s, n_sk, x, v, seed=seed, chain_id_offset=offset)
    ref = O.sample_skeleton(cfg, n_sk, x, v, seed, chain_id_offset=offset)
    return s, cfg, got, ref


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_rng_stream_bitwise(engine, oracle, kind):
    s = engine.ZigZagAD(2, engine.GaussianIso())
    from pdmpflux.jl_amd import _lib
    n = 1001
    out = np.empty(n)
    for seed, chain in [(0, 0), (2024, 7), (2**63 + 5, 2**40 + 3)]:
        _lib.check(_lib.lib().pdmp_debug_rng(s.handle(), C.c_uint64(seed), chain, kind, n, C.c_void_p(out.ctypes.data)))
        ref = oracle.rng_dump(seed, chain, kind, n)
        assert np.array_equal(out, ref), (kind, seed, chain)


CASES = [
    # kind, dim, target, chains, n_sk, kwargs
    ("ZigZagAD", 50, "iso", 24, 300, {}),
    ("ZigZagAD", 10, "iso", 5, 400, {}),
    ("ZigZagAD", 1, "iso", 3, 300, {}),
    ("ZigZagAD", 64, "diag", 8, 200, {}),
    ("ZigZagAD", 50, "banana", 16, 300, {}),
    ("ZigZagAD", 100, "iso", 6, 150, {}),           # two coordinates per lane
    ("ZigZagAD", 130, "diag", 4, 100, {}),          # K = 4 path, ragged last chunk
    ("ZigZag", 50, "iso", 8, 200, {}),              # default FiniteDiff slopes (Q13)
    ("ZigZag", 20, "banana", 8, 200, {}),
    ("ZigZagAD", 50, "iso", 8, 200, {"signed_bound": False}),
    ("ZigZagAD", 50, "banana", 8, 200, {"adaptive": False, "tmax": 0.5}),
    ("ZigZagAD", 50, "iso", 8, 200, {"grid_size": 5}),
    ("ZigZagAD", 50, "banana", 8, 200, {"grid_size": 20}),
    ("ZigZagAD", 50, "iso", 8, 100, {"tmax": 0}),   # tmax = 0 -> 1.0 + adaptive
    ("ZigZagAD", 2, "banana", 8, 1500, {"tmax": 20.0, "adaptive": False}),     # forces ar > 1 (error path), group kernel
    ("ZigZagAD", 2, "banana", 8, 1500, {"tmax": 2.0, "adaptive": False, "grid_size": 2}),   # same in the wave kernel
    ("BPS", 2, "banana", 8, 1500, {"tmax": 20.0, "adaptive": False}),
    ("BPS", 50, "banana", 16, 300, {}),
    ("BPS", 50, "iso", 8, 300, {}),
    ("BPS", 100, "diag", 4, 150, {}),
    ("BPS", 10, "iso", 4, 300, {"Gaussian_velocity": True}),
    ("BPS", 50, "banana", 8, 200, {"signed_bound": False}),
    ("BPS", 50, "banana", 8, 200, {"AD_backend": "FiniteDiff"}),
    ("BPSAD", 20, "iso", 4, 200, {}),               # refresh_rate = 0 (Q9)
    ("ZigZagAD", 100, "dense", 5, 120, {}),         # dense precision: grad-U = P x by per-chain mat-vecs (config 4 shape)
    ("ZigZagAD", 7, "dense", 3, 200, {}),
    ("BPS", 70, "dense", 4, 120, {}),
    ("ZigZagAD", 300, "dense", 2, 40, {}),
    ("ZigZagAD", 50, "iso", 8, 150, {"grid_size": 0}),          # Brent constant bound (what most reference tests use)
    ("ZigZagAD", 3, "banana", 8, 400, {"grid_size": 0}),
    ("BPSAD", 20, "banana", 6, 150, {"grid_size": 0}),
    ("BoomerangAD", 10, "diag", 6, 150, {"grid_size": 0, "refresh_rate": 0.1}),
    ("ZigZagAD", 70, "dense", 3, 60, {"grid_size": 0}),
    ("Boomerang", 100, "iso", 6, 200, {}),          # default FiniteDiff
    ("BoomerangAD", 50, "diag", 6, 200, {"refresh_rate": 0.1}),
    ("Boomerang", 50, "banana", 6, 200, {"AD_backend": "ForwardDiff"}),
]


@pytest.mark.parametrize("kind,dim,target,chains,n_sk,kw", CASES,
                         ids=[f"{c[0]}-d{c[1]}-{c[2]}-{'_'.join(f'{k}{v}' for k, v in c[5].items()) or 'default'}" for c in CASES])
def test_skeleton_bitwise(engine, oracle, kind, dim, target, chains, n_sk, kw):
    _, _, got, ref = run_pair(engine, oracle, kind, dim, target, chains, n_sk, **kw)
    assert_bitwise(got, ref)
    if kw.get("adaptive") is False and kw.get("tmax", 0) >= 2.0 and target == "banana":
        assert ref["errored_bound"].sum() > 10      # the rare branch really ran (and matched bit for bit)


@pytest.mark.parametrize("G", [0, 1, 2, 4, 8, 16, 32, 64])
@pytest.mark.parametrize("kind,dim,target", [("ZigZagAD", 50, "iso"), ("ZigZagAD", 37, "banana"), ("BPS", 50, "banana"),
                                             ("BoomerangAD", 24, "diag"), ("ZigZagAD", 3, "diag"),
                                             ("ZigZag", 20, "banana"), ("Boomerang", 30, "iso"), ("BPS", 9, "diag")])
def test_group_layouts_bitwise(engine, oracle, monkeypatch, G, kind, dim, target):
    """Every lanes-per-chain layout of the group kernel (and G=0: the wave kernel) against the oracle run
    in the same GROUP(G) summation order; 70 chains so that waves are ragged for every G."""
    monkeypatch.setenv("PDMP_GROUP_LANES", str(G))
    kw = {"refresh_rate": 0.1} if kind == "BoomerangAD" else ({"AD_backend": "FiniteDiff"} if kind == "BPS" and dim == 9 else {})
    s, cfg, got, ref = run_pair(engine, oracle, kind, dim, target, 70, 120, **kw)
    assert s.group_lanes() == G
    assert_bitwise(got, ref)


@pytest.mark.parametrize("segments", [2, 7, 1000])
@pytest.mark.parametrize("kind,dim,target", [("ZigZagAD", 50, "iso"), ("BPS", 50, "banana"), ("BoomerangAD", 10, "iso")])
def test_segment_handoff_bitwise(engine, oracle, monkeypatch, segments, kind, dim, target):
    """The dynamic schedule splits a chain's events into segments executed by different waves with a
    (x, v, t, horizon, ar, rng counter) hand-off through HBM; forcing 2 / 7 / one-event segments on a small
    ensemble must not change a single bit."""
    monkeypatch.setenv("PDMP_FORCE_SEGMENTS", str(segments))
    kw = {"refresh_rate": 0.1} if kind == "BoomerangAD" else {}
    _, _, got, ref = run_pair(engine, oracle, kind, dim, target, 70, 90, **kw)
    assert_bitwise(got, ref)


def test_full_machine_schedule_bitwise(engine, oracle):
    """More batches than resident wave slots (40 000 chains -> 2 500 batches of 16): the persistent
    kernel really schedules several segments per chain across waves; a strided sample of chains must
    equal the oracle bit for bit, and every chain must satisfy the path's invariants."""
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    x, v = make_init("ZigZagAD", 50, 40000)
    got = engine.sample_skeleton_ensemble(s, 64, x, v, seed=77, diagnostics=False)
    idx = np.arange(0, 40000, 797)
    # chain ids are global: chain idx[j] of the big run uses stream (seed, idx[j])
    for j, c in enumerate(idx):
        r1 = oracle.sample_skeleton(cfg, 64, x[c:c + 1], v[c:c + 1], 77, chain_id_offset=int(c), diagnostics=False)
        for k in ("X", "V", "t"):
            assert np.array_equal(got[k][c], r1[k][0]), (k, c)
    assert (np.diff(got["t"], axis=1) > 0).all()                                  # test_comprehensive.jl:147
    assert np.abs(np.linalg.norm(got["V"], axis=2) - np.sqrt(50)).max() < 1e-12   # :172-180 speed conservation


@pytest.mark.parametrize("kind,dim,target,C", [("ZigZagAD", 48, "iso", 40000), ("ZigZagAD", 50, "banana", 40000),
                                               ("BPS", 40, "banana", 40000), ("BPS", 50, "banana", 40000),
                                               ("BoomerangAD", 90, "iso", 20000), ("Boomerang", 44, "diag", 40000)])
def test_full_machine_schedule_other_register_kernels(engine, oracle, kind, dim, target, C):
    """The same multi-segment schedule through the exact-slot-count kernels (K = 12 / 10 / 11) and the FULL banana variants:
    a strided sample of chains bit for bit against the oracle, event times increasing for every chain."""
    s, cfg = make_pair(engine, oracle, kind, dim, target)
    x, v = make_init(kind, dim, C, target=target)
    got = engine.sample_skeleton_ensemble(s, 64, x, v, seed=78, diagnostics=False)
    for c in np.arange(0, C, 1999):
        r1 = oracle.sample_skeleton(cfg, 64, x[c:c + 1], v[c:c + 1], 78, chain_id_offset=int(c), diagnostics=False)
        for k in ("X", "V", "t"):
            assert np.array_equal(got[k][c], r1[k][0]), (k, c)
    assert (np.diff(got["t"], axis=1) > 0).all()


def test_group1_sums_in_reference_order(engine, oracle, monkeypatch):
    """With one lane per chain every sum over coordinates runs in the reference's own left-to-right order; only
    cum_sum (cumsum after, not before, the sum over dimensions) can differ in the last bit.  Velocities are
    identical, times and positions agree to 1e-12 relative with the reference-order restatement."""
    monkeypatch.setenv("PDMP_GROUP_LANES", "1")
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso", arith=oracle.ARITH_REFERENCE)
    x, v = make_init("ZigZagAD", 50, 40)
    got = engine.sample_skeleton_ensemble(s, 150, x, v, seed=9)
    ref = oracle.sample_skeleton(cfg, 150, x, v, 9)
    assert np.array_equal(got["V"], ref["V"])
    assert np.allclose(got["t"], ref["t"], rtol=1e-12, atol=0) and np.allclose(got["X"], ref["X"], rtol=1e-10, atol=1e-12)
    for k in ("rejected", "hitting_horizon", "errored_bound"):
        assert np.array_equal(got[k], ref[k])


def test_chain_offset_and_sharding_invariance(engine, oracle):
    """Chain c of a shard starting at offset o equals chain o+c of the full ensemble (multi-GPU contract)."""
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    x, v = make_init("ZigZagAD", 50, 12)
    full = engine.sample_skeleton_ensemble(s, 100, x, v, seed=5)
    part = engine.sample_skeleton_ensemble(s, 100, x[5:9], v[5:9], seed=5, chain_id_offset=5)
    for k in ("X", "V", "t"):
        assert np.array_equal(full[k][5:9], part[k])


def test_reference_order_within_tolerance(engine, oracle):
    """North-star bound: event times/positions within 1e-5 relative of the reference-order (sequential
    sums, per-dimension cumsum) restatement on identical streams."""
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso", arith=oracle.ARITH_REFERENCE)
    x, v = make_init("ZigZagAD", 50, 8)
    got = engine.sample_skeleton_ensemble(s, 200, x, v, seed=3)
    ref = oracle.sample_skeleton(cfg, 200, x, v, 3)
    rtol = 1e-5
    assert np.allclose(got["t"], ref["t"], rtol=rtol, atol=0)
    assert np.allclose(got["X"], ref["X"], rtol=rtol, atol=1e-9)
    assert np.array_equal(got["V"], ref["V"])


@pytest.mark.parametrize("kind,dim,discard", [("ZigZagAD", 50, True), ("ZigZagAD", 100, False), ("Boomerang", 50, True),
                                              ("Boomerang", 10, False), ("BPS", 3, True)])
def test_sample_from_skeleton_bitwise(engine, oracle, kind, dim, discard):
    s, cfg, got, _ = run_pair(engine, oracle, kind, dim, "iso", 6, 120)
    for N in (1, 63, 64, 65, 500):
        a = engine.sample_from_skeleton_ensemble(s, N, got["X"], got["V"], got["t"], discard_vt=discard)
        b = oracle.sample_from_skeleton(cfg, N, got["X"], got["V"], got["t"], discard_vt=discard)
        assert np.array_equal(a, b), N


def test_single_chain_api_matches_reference_shapes(engine, oracle):
    """test/test_quick.jl:15-26, test_samplers.jl:33-48: shapes and finiteness through the drop-in API."""
    s = engine.ZigZagAD(3, engine.GaussianIso())
    h = engine.sample_skeleton(s, 100, np.zeros(3), np.ones(3), seed=42)
    assert h.X.shape == (3, 100) and h.V.shape == (3, 100) and len(h.t) == 100
    assert np.isfinite(h.X).all() and np.isfinite(h.V).all() and np.isfinite(h.t).all()
    assert (np.diff(h.t) > 0).all()                                  # test_comprehensive.jl:147
    smp = engine.sample_from_skeleton(s, 50, h)
    assert smp.shape == (3, 50)
    smp2 = engine.sample(s, 100, 50, np.zeros(3), np.ones(3), seed=42)
    assert np.array_equal(smp, smp2)                                 # reproducibility, test_quick.jl:96-108
    h1 = engine.sample_skeleton(engine.ZigZagAD(1, engine.GaussianIso()), 50, 0.0, 1.0, seed=1)   # scalar dispatch
    assert h1.X.shape == (1, 50)


@pytest.mark.parametrize("kind,dim", [("ZigZagAD", 50), ("Boomerang", 70), ("BPS", 300)])
def test_fused_moments_equal_sequential_sums_of_samples(engine, oracle, kind, dim):
    """pdmp_sample_moments_device against sequential sums over the oracle's sample_from_skeleton."""
    import torch
    from pdmpflux.jl_amd import ensemble
    s, cfg, got, _ = run_pair(engine, oracle, kind, dim, "iso", 5, 80)
    dev = torch.device("cuda:0")
    X, V, t = (torch.from_numpy(got[k]).to(dev) for k in ("X", "V", "t"))
    for N in (1, 64, 130):
        s1, s2 = ensemble.sample_moments_device(s, N, X, V, t)
        smp = oracle.sample_from_skeleton(cfg, N, got["X"], got["V"], got["t"])       # (C, N, d)
        assert np.array_equal(s1.cpu().numpy(), np.cumsum(smp, axis=1)[:, -1, :])      # cumsum = left-to-right sums
        assert np.array_equal(s2.cpu().numpy(), np.cumsum(smp * smp, axis=1)[:, -1, :])


def test_sharded_driver_single_rank_moments(engine, oracle):
    """ensemble.sample_skeleton_sharded on one rank (no process group): device path, fused moments."""
    import torch
    from pdmpflux.jl_amd import ensemble
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    x, v = make_init("ZigZagAD", 50, 256)
    dev = torch.device("cuda:0")
    r = ensemble.sample_skeleton_sharded(s, 400, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=3,
                                         gather="moments", N_samples=400)
    assert r["chain_range"] == (0, 256) and int(r["count"]) == 256 * 400
    assert float(r["mean"].abs().max()) < 0.15 and float((r["var"] - 1).abs().max()) < 0.25
    ref = oracle.sample_skeleton(cfg, 400, x[:4], v[:4], 3, diagnostics=False)
    assert np.array_equal(r["t"][:4].cpu().numpy(), ref["t"])


def test_time_horizon_variant(engine, oracle):
    """sample_skeleton(sampler, T::Float64, ...) (src/sample.jl:323-439): t[end] == T exactly and monotone
    (test/test_samplers.jl:56-62); the events before T are the fixed-count skeleton's; closing point by flow."""
    P = engine
    s = P.ZigZagAD(3, P.GaussianIso())
    x0, v0 = np.zeros(3), np.ones(3)
    T = 25.0
    h = P.sample_skeleton(s, T, x0, v0, seed=42, init_capacity=8)        # forces several capacity doublings
    assert h.t[-1] == T and h.t[0] == 0.0 and (np.diff(h.t) > 0).all()
    n = len(h.t)
    fixed = P.sample_skeleton(s, n + 3, x0, v0, seed=42)
    assert np.array_equal(h.t[:-1], fixed.t[:n - 1]) and np.array_equal(h.X[:, :-1], fixed.X[:, :n - 1])
    assert fixed.t[n - 1] > T                                             # the next event overshoots
    assert np.array_equal(h.X[:, -1], fixed.X[:, n - 2] + fixed.V[:, n - 2] * (T - fixed.t[n - 2]))
    assert np.array_equal(h.V[:, -1], fixed.V[:, n - 2])
    assert h.ar[-1] == 0.0 and h.rejected[-1] == 0 and h.horizon[-1] == fixed.horizon[n - 1]
    smp = P.sample_from_skeleton(s, 100, h)                               # t[end] == T is the time scale of the samples
    assert smp.shape == (3, 100) and np.isfinite(smp).all()
    h0 = P.sample_skeleton(s, 0.0, x0, v0, seed=42)                       # T == 0: the initial point alone (:356-359)
    assert len(h0.t) == 1 and h0.t[0] == 0.0
    hs = P.sample_skeleton(s, 5.0, np.zeros((4, 3)), np.ones((4, 3)), seed=1)   # ensemble: ragged lengths
    assert len(hs) == 4 and all(hh.t[-1] == 5.0 for hh in hs) and len({len(hh.t) for hh in hs}) > 1


def test_sample_from_skeleton_dt_method(engine, oracle):
    """sample_from_skeleton(sampler, dt::Float64, history) (src/sample.jl:573-646): n = floor(t[end]/dt) samples at j*dt."""
    s, cfg, got, _ = run_pair(engine, oracle, "ZigZagAD", 5, "iso", 1, 200)
    h = engine.sample_skeleton(s, 200, *make_init("ZigZagAD", 5, 1), seed=11)
    h = engine.PDMPHistory(got["X"][0].T, got["V"][0].T, got["t"][0], np.ones((5, 200), bool), got["horizon"][0], got["ar"][0],
                           got["errored_bound"][0], got["error_value_ar"][0].T, got["rejected"][0], got["hitting_horizon"][0])
    dt = 0.37
    a = engine.sample_from_skeleton(s, dt, h, discard_vt=False)
    n = int(np.floor(h.t[-1] / dt))
    assert a.shape == (11, n)
    b = oracle.sample_from_skeleton(cfg, n, got["X"], got["V"], got["t"], discard_vt=False, dt=dt)[0].T
    assert np.array_equal(a, b)
    assert np.array_equal(a[10], np.arange(1, n + 1) * dt)                 # out[2d+1, j] = j*dt, :640


def test_reference_ergodicity_test_with_brent_bound(engine):
    """test/test_property_based.jl:87-100 verbatim: ZigZagAD(1, U_Gauss_1D, grid_size=0), 5000 events, 5000 samples,
    seed 42: |mean| < 0.2 and 0.8 < var < 1.2 -- through the drop-in API on the GPU."""
    s = engine.ZigZagAD(1, engine.GaussianIso(), grid_size=0)
    out = engine.sample_skeleton(s, 5000, 0.0, 1.0, seed=42)
    smp = engine.sample_from_skeleton(s, 5000, out)
    assert abs(smp.mean()) < 0.2 and 0.8 < smp.var() < 1.2
    assert len(out.t) == 5000 and (np.diff(out.t) > 0).all()


def test_iteration_cap_is_reported_not_hung(engine):
    """A chain that exceeds max_iter_per_event thinning steps for one event stops and the call returns
    PDMP_ERR_ITERCAP (-4) -- never an endless loop on the GPU (failure detection)."""
    from pdmpflux.jl_amd import _lib
    x, v = make_init("ZigZagAD", 2, 8, target="banana")
    for kw in ({}, {"grid_size": 5}):                      # group kernel and wave kernel
        s = engine.ZigZagAD(2, engine.Banana(), tmax=20.0, adaptive=False, max_iter_per_event=1, **kw)
        with pytest.raises(engine.EngineError) as e:
            engine.sample_skeleton_ensemble(s, 400, x, v, seed=3)
        assert e.value.code == _lib.ERR_ITERCAP
        ok = engine.ZigZagAD(2, engine.Banana(), tmax=20.0, adaptive=False, **kw)     # same run with the default cap is fine
        out = engine.sample_skeleton_ensemble(ok, 400, x, v, seed=3)
        assert np.isfinite(out["t"]).all()


@pytest.mark.parametrize("dim,mode", [(3, 2), (64, 2), (256, 2), (512, 2), (513, 2), (1024, 2), (1025, 1)])
def test_dense_precision_slot_boundaries(engine, oracle, dim, mode):
    """The incremental kernel is instantiated for 4 / 8 / 16 slots of 64 coordinates (d <= 256 / 512 / 1024); beyond that the
    GEMM-per-phase form serves the call.  Every boundary against the reference-order restatement, to the tolerance of
    test_dense_precision_batched_paths_match_oracle."""
    from pdmpflux.jl_amd import _lib
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, "dense", arith=oracle.ARITH_REFERENCE)
    C, n_sk = 64, 10
    x, v = make_init("ZigZagAD", dim, C)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=2)
    assert _lib.lib().pdmp_dense_batched_last_mode() == mode
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 2)
    assert np.array_equal(got["V"], ref["V"])
    assert np.allclose(got["t"], ref["t"], rtol=1e-12, atol=0)
    assert np.allclose(got["X"], ref["X"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("mode", ["incr", "gemm"])
def test_iteration_cap_on_the_dense_paths(engine, monkeypatch, mode):
    """The same guarantee on the batched dense-precision paths (dense_batched.hip): a fixed horizon far beyond the event
    scale makes almost every proposal a rejection, the cap stops the chains and the call reports PDMP_ERR_ITERCAP."""
    from pdmpflux.jl_amd import _lib
    monkeypatch.setenv("PDMP_DENSE_MODE", mode)
    rs = np.random.default_rng(5)
    A = rs.normal(size=(40, 40))
    tg = engine.GaussianPrecision(A @ A.T / 40 + np.eye(40))
    x, v = make_init("ZigZagAD", 40, 64)
    s = engine.ZigZagAD(40, tg, tmax=40.0, adaptive=False, max_iter_per_event=1)
    with pytest.raises(engine.EngineError) as e:
        engine.sample_skeleton_ensemble(s, 50, x, v, seed=3)
    assert e.value.code == _lib.ERR_ITERCAP
    ok = engine.ZigZagAD(40, tg)
    out = engine.sample_skeleton_ensemble(ok, 50, x, v, seed=3)
    assert np.isfinite(out["t"]).all() and (np.diff(out["t"], axis=1) > 0).all()


def test_empty_and_degenerate_ensembles(engine, oracle):
    """Edge cases: no chains, one skeleton point (initial state only), one chain, ragged last wave."""
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    e = engine.sample_skeleton_ensemble(s, 10, np.zeros((0, 50)), np.zeros((0, 50)), seed=1)
    assert e["X"].shape == (0, 10, 50)
    x, v = make_init("ZigZagAD", 50, 17)
    one = engine.sample_skeleton_ensemble(s, 1, x, v, seed=1)
    assert np.array_equal(one["X"][:, 0], x) and np.array_equal(one["V"][:, 0], v) and (one["t"] == 0).all()
    got = engine.sample_skeleton_ensemble(s, 33, x, v, seed=1)
    ref = oracle.sample_skeleton(cfg, 33, x, v, 1)
    assert_bitwise(got, ref)
    assert np.array_equal(x, make_init("ZigZagAD", 50, 17)[0])          # inputs are never mutated (Q7)


@pytest.mark.parametrize("kind,dim,target,kw", [("ZigZagAD", 50, "iso", {}), ("BPS", 20, "banana", {"refresh_rate": 0.5}),
                                                ("Boomerang", 9, "diag", {})])
def test_tape_replay_bitwise(engine, oracle, monkeypatch, kind, dim, target, kw):
    """Tape source: both sides replay the same externally generated variates (numpy's generators here,
    a Julia dump in the intended use) instead of Philox; trajectories must still agree bit for bit.  Here on the wave-per-chain
    kernel (PDMP_TAPE_WAVE=1; round 1's only replaying form); every other kernel form: tests/test_gpu_tape_replay.py."""
    monkeypatch.setenv("PDMP_TAPE_WAVE", "1")
    rs = np.random.default_rng(5)
    C_ = 6
    tape = (rs.exponential(size=(C_, 3000)), rs.random(size=(C_, 3000)), rs.normal(size=(C_, 6000)))
    s, cfg = make_pair(engine, oracle, kind, dim, target, arith=oracle.ARITH_WAVE64, **kw)
    s.set_tape(*tape)
    assert s.group_lanes() == 0
    cfg.tape = tape
    x, v = make_init(kind, dim, C_, target=target)
    got = engine.sample_skeleton_ensemble(s, 150, x[2:5], v[2:5], seed=0, chain_id_offset=2)   # tapes follow the global chain id
    ref = oracle.sample_skeleton(cfg, 150, x[2:5], v[2:5], 0, chain_id_offset=2)
    assert_bitwise(got, ref)
    with pytest.raises(engine.ArgumentError):                                   # tape too short for 5000 events
        engine.sample_skeleton_ensemble(s, 5000, x, v, seed=0)
    s.clear_tape()
    again = engine.sample_skeleton_ensemble(s, 20, x, v, seed=3)               # back on Philox and the fast kernel
    assert np.isfinite(again["t"]).all() and s.group_lanes() > 0


@pytest.mark.parametrize("mode,resync", [("incr", "256"), ("incr", "16"), ("incr", "1"), ("gemm", "256")])
@pytest.mark.parametrize("dim,chains,n_sk", [(100, 64, 80), (257, 70, 40), (600, 64, 24)])
def test_dense_precision_batched_paths_match_oracle(engine, oracle, monkeypatch, dim, chains, n_sk, mode, resync):
    """Config-4 path, >= 64 chains.  Default ("incr"): grad-U is linear and the flow is x + v t, so P x and P v are kept current
    by O(d) updates (P x += tau P v at a flow, P v -= 2 v_m P[:, m] at a flip) and re-based on exact products by one fp64
    GEMM over all chains every PDMP_DENSE_RESYNC events.  "gemm" (d > 1024): grad-U = P x of ALL chains as one GEMM per
    phase (dense_batched.hip).  Neither has the checker's left-to-right row sums (rocBLAS' MFMA accumulation order; the
    recurrences), so both are held to a tolerance -- event times within 1e-12 relative (measured: 1.5e-15 over 300 events) of the reference-order
    restatement, velocities and counters identical -- while PDMP_DENSE_MODE=matvec (per-chain mat-vec kernel) is bit-exact."""
    monkeypatch.setenv("PDMP_DENSE_MODE", mode)
    monkeypatch.setenv("PDMP_DENSE_RESYNC", resync)
    kw = {"signed_bound": False} if dim == 257 else {}          # the unsigned bound keeps the division of the tangent intersection
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, "dense", arith=oracle.ARITH_REFERENCE, **kw)
    x, v = make_init("ZigZagAD", dim, chains)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=21, diagnostics=True)
    from pdmpflux.jl_amd import _lib
    assert _lib.lib().pdmp_dense_batched_last_mode() == (2 if mode == "incr" else 1)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 21, diagnostics=True)
    assert np.array_equal(got["V"], ref["V"])
    for k in ("rejected", "hitting_horizon", "errored_bound"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.allclose(got["t"], ref["t"], rtol=1e-12, atol=0)
    assert np.allclose(got["X"], ref["X"], rtol=1e-10, atol=1e-12)
    assert np.allclose(got["horizon"], ref["horizon"], rtol=1e-12) and np.allclose(got["ar"], ref["ar"], rtol=1e-7, atol=1e-12)
    monkeypatch.setenv("PDMP_DENSE_MODE", "matvec")
    s2, cfg2 = make_pair(engine, oracle, "ZigZagAD", dim, "dense")
    got2 = engine.sample_skeleton_ensemble(s2, 12, x[:6], v[:6], seed=21)
    assert_bitwise(got2, oracle.sample_skeleton(cfg2, 12, x[:6], v[:6], 21))


@pytest.mark.parametrize("dim,chains,n_sk,resync", [(100, 70, 60, "256"), (600, 66, 24, "8"), (300, 64, 40, "256")])
def test_dense_precision_unit_and_other_speeds_in_one_ensemble(engine, oracle, monkeypatch, dim, chains, n_sk, resync):
    """The incremental form with velocities other than the +-1 the constructors make (ZigZag takes any v0, AbstractPDMP.jl:93-153; a
    flip keeps |v_i|): an ensemble that mixes unit-speed chains with chains of speeds 1/2, 1, 2 by coordinate, segments of 8 and of
    256 events, against the reference-order restatement.  (Round 5 built a second segment kernel for unit speeds -- v as sign bits,
    the signed products (P x) v, (P v) v in registers -- with this test as its gate: 22 % less arithmetic time, the same launch time,
    not shipped)"""
    monkeypatch.setenv("PDMP_DENSE_MODE", "incr")
    monkeypatch.setenv("PDMP_DENSE_RESYNC", resync)
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, "dense", arith=oracle.ARITH_REFERENCE)
    x, v = make_init("ZigZagAD", dim, chains)
    rs = np.random.default_rng(dim)
    v[1::3] *= rs.choice([0.5, 2.0, 1.0], size=v[1::3].shape)          # every third chain: speeds 1/2, 1, 2 by coordinate
    v[5, :] *= 1.0; v[5, dim - 1] = -3.0                                  # one chain that differs in its last coordinate only
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=33, diagnostics=True)
    from pdmpflux.jl_amd import _lib
    assert _lib.lib().pdmp_dense_batched_last_mode() == 2
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 33, diagnostics=True)
    assert np.array_equal(got["V"], ref["V"])
    for k in ("rejected", "hitting_horizon", "errored_bound"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.allclose(got["t"], ref["t"], rtol=1e-12, atol=0)
    assert np.allclose(got["X"], ref["X"], rtol=1e-10, atol=1e-12)
    assert np.allclose(got["horizon"], ref["horizon"], rtol=1e-12) and np.allclose(got["ar"], ref["ar"], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("kind,dim,target,kw", [
    ("ZigZagAD", 5, "iso", {}), ("ZigZagAD", 13, "diag", {}), ("ZigZagAD", 26, "banana", {}), ("ZigZagAD", 50, "iso", {}),
    ("ZigZag", 50, "banana", {}), ("ZigZagAD", 100, "diag", {}), ("ZigZagAD", 200, "iso", {}),
    ("ZigZagAD", 50, "iso", {"signed_bound": False}), ("ZigZagAD", 50, "diag", {"grid_size": 0}),
    ("BPS", 50, "iso", {}), ("BPS", 50, "banana", {}), ("BPS", 30, "diag", {"AD_backend": "FiniteDiff"}), ("BPS", 13, "iso", {"Gaussian_velocity": True}),
    ("BPS", 104, "banana", {"AD_backend": "FiniteDiff"}), ("BPS", 50, "iso", {"grid_size": 0}),
    ("Boomerang", 100, "iso", {}), ("BoomerangAD", 100, "iso", {}), ("BoomerangAD", 60, "diag", {}), ("Boomerang", 26, "iso", {}),
    ("BoomerangAD", 50, "banana", {}),
    ("RHMC", 50, "iso", {}), ("RHMC", 50, "banana", {"step_size": 0.013}), ("RHMC", 13, "diag", {"phi": 0.7}), ("RHMC", 100, "iso", {"refresh_rate": 2.5}),
    ("ForwardECMC", 50, "iso", {}), ("ForwardECMC", 50, "banana", {"mix_p": 0.9}), ("ForwardECMC", 7, "diag", {}), ("ForwardECMC", 13, "iso", {"switch": False}),
    ("ForwardECMC", 30, "diag", {"ran_p": False}), ("ForwardECMC", 100, "banana", {"positive": False}), ("ForwardECMC", 3, "iso", {}),
])
def test_register_and_lds_forms_agree(engine, oracle, monkeypatch, kind, dim, target, kw):
    """With ceil(d/G) <= 13 the group kernel keeps x, v in registers (inst_reg_*.hip; RegForm says for which sampler /
    target / slope mode); PDMP_REG=0 keeps them in LDS tiles.  Same operations in the same order: both forms equal the
    checker bit for bit."""
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C, n_sk = 21, 40
    x, v = make_init(kind, dim, C, seed=8, target=target)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 17, diagnostics=True)
    reg = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=17, diagnostics=True)
    monkeypatch.setenv("PDMP_REG", "0")
    lds = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=17, diagnostics=True)
    assert_bitwise(reg, ref)
    assert_bitwise(lds, ref)


@pytest.mark.parametrize("dim", [209, 224, 260, 300, 352, 416, 417, 500, 640, 768, 832])   # G = 32: K = 7, 7, 9, 10, 11, 13; G = 64: K = 7, 8, 10, 12, 13 (13 at G = 64: LDS form)
@pytest.mark.parametrize("target", ["iso", "diag", "banana"])
def test_wide_zigzag_register_kernels(engine, oracle, monkeypatch, dim, target):
    """Round 4 (last): Zig-Zag at d = 209 .. 832 keeps x, v in registers with half a wave (G = 32) or a whole wave (G = 64) per chain --
    the exact-slot-count kernels and the 13-slot ones; before, these dimensions ran the LDS form.  Both equal the checker bit for bit."""
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, target)
    assert s.group_lanes() == (32 if dim <= 416 else 64)
    C, n_sk = 9, 30
    x, v = make_init("ZigZagAD", dim, C, seed=8, target=target)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 17, diagnostics=True)
    assert_bitwise(engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=17, diagnostics=True), ref)
    assert s.last_launch_form() == (3 if dim <= 768 else 2)            # (a whole wave per chain with thirteen slots would spill: LDS form)
    monkeypatch.setenv("PDMP_FORCE_SEGMENTS", "3")
    monkeypatch.setenv("PDMP_REG_EXACT", "0")                          # the 13-slot kernel with per-slot skips (G = 32, K < 13), hand-off segments
    s2, _ = make_pair(engine, oracle, "ZigZagAD", dim, target)
    assert_bitwise(engine.sample_skeleton_ensemble(s2, n_sk, x, v, seed=17, diagnostics=True), ref)
    monkeypatch.setenv("PDMP_REG", "0")
    s3, _ = make_pair(engine, oracle, "ZigZagAD", dim, target)
    assert_bitwise(engine.sample_skeleton_ensemble(s3, n_sk, x, v, seed=17, diagnostics=True), ref)
    assert s3.last_launch_form() == 2


@pytest.mark.parametrize("dim", [14, 30, 70, 160, 22, 48, 96, 183])   # (G, K) = (2,7) (4,8) (8,9) (16,10) (2,11) (4,12) (8,12) (16,12)
@pytest.mark.parametrize("kind,target", [("ZigZagAD", "iso"), ("ZigZagAD", "diag"), ("ZigZagAD", "banana"),
                                         ("BPS", "iso"), ("BPS", "diag"), ("BPS", "banana"),
                                         ("Boomerang", "iso"), ("Boomerang", "diag"), ("BoomerangAD", "iso"), ("BoomerangAD", "banana")])
def test_exact_slot_count_kernels(engine, oracle, monkeypatch, kind, dim, target):
    _exact_slot_case(engine, oracle, monkeypatch, kind, dim, target)


@pytest.mark.parametrize("dim", [2, 3, 5, 6, 7, 10, 12])                  # one lane per chain: K = d
@pytest.mark.parametrize("kind,target", [("ZigZagAD", "iso"), ("ZigZagAD", "banana"), ("BPS", "diag"), ("BPS", "banana"),
                                         ("Boomerang", "iso"), ("BoomerangAD", "diag")])
def test_exact_slot_count_kernels_one_lane_per_chain(engine, oracle, monkeypatch, kind, dim, target):
    _exact_slot_case(engine, oracle, monkeypatch, kind, dim, target, one_lane=True)


def _exact_slot_case(engine, oracle, monkeypatch, kind, dim, target, one_lane=False):
    """Forward-mode Zig-Zag / BPS and Boomerang (both slope modes) with K = ceil(d/G) in 7..12 run a register-form kernel
    of exactly K slots (inst_reg_*_k.hip, inst_reg_boomerang_kfd.hip); PDMP_REG_EXACT=0 selects the 13-slot kernel with per-slot skips.  Both equal the checker."""
    s, cfg = make_pair(engine, oracle, kind, dim, target)
    if kind == "ZigZagAD" and target in ("iso", "banana") and dim <= 26:
        s.set_group_lanes(1 if one_lane else 2)                    # (round 4: these default to G = 4, the lane kernel's layout; the (2, K) kernels stay under test)
        cfg.group_lanes = 1 if one_lane else 2
    G = s.group_lanes()
    assert (G == 1 and 2 <= dim <= 12) if one_lane else (G >= 2 and 7 <= -(-dim // G) <= 12)
    C, n_sk = (70, 30) if one_lane else (19, 30)
    x, v = make_init(kind, dim, C, seed=31, target=target)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 23, diagnostics=True)
    assert_bitwise(engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=23, diagnostics=True), ref)
    monkeypatch.setenv("PDMP_REG_EXACT", "0")
    assert_bitwise(engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=23, diagnostics=True), ref)


@pytest.mark.parametrize("budget", ["1", "2", "3", "5", "8", "9"])
@pytest.mark.parametrize("kind,dim,target,kw", [
    ("ZigZagAD", 50, "iso", {}), ("ZigZagAD", 50, "banana", {}), ("ZigZag", 20, "banana", {}), ("ZigZagAD", 7, "diag", {"signed_bound": False}),
    ("ZigZagAD", 300, "iso", {}),
])
def test_bound_instalments_do_not_change_results(engine, oracle, monkeypatch, budget, kind, dim, target, kw):
    """The lazily built Zig-Zag bound is advanced `PDMP_CELL_BUDGET` cells per pass of the state machine (default 3 on a full machine, all
    nine when the launch leaves every wave alone on its SIMD -- as every launch of this test does --, so 3 is set explicitly too) and
    resumed after rejections; per chain the cells, their order and the sums are the same for every budget."""
    monkeypatch.setenv("PDMP_CELL_BUDGET", budget)
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C, n_sk = 37, 50
    x, v = make_init(kind, dim, C, seed=21, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=5, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 5, diagnostics=True)
    assert_bitwise(got, ref)


def _random_cases(n, seed):
    rs = np.random.default_rng(seed)
    cases = []
    for q in range(n):
        kind = ["ZigZag", "ZigZagAD", "BPS", "BPSAD", "Boomerang", "BoomerangAD", "ForwardECMC", "SpeedUpZigZagAD", "RHMC"][rs.integers(9)]
        target = ["iso", "diag", "banana"][rs.integers(3)]
        dim = int(rs.choice([2, 3, 5, 9, 13, 14, 27, 40, 52, 53, 77, 104, 130, 209, 260]))
        kw = {}
        if rs.random() < 0.3:
            kw["AD_backend"] = ["ForwardDiff", "FiniteDiff"][rs.integers(2)]
        if rs.random() < 0.25:
            kw["signed_bound"] = False
        if rs.random() < 0.2:
            kw["adaptive"] = False
        if rs.random() < 0.3:
            kw["tmax"] = float(rs.choice([0.0, 0.3, 1.0, 5.0]))
        if kind.startswith(("ZigZag", "SpeedUp")) and rs.random() < 0.2:
            kw["vectorized_bound"] = False
        if kind.startswith(("BPS", "Boomerang")) and rs.random() < 0.5:
            kw["refresh_rate"] = float(rs.choice([0.0, 0.1, 1.0, 3.0]))
        if kind == "ForwardECMC":
            kw.update(ran_p=bool(rs.integers(2)), switch=bool(rs.integers(2)), positive=bool(rs.integers(2)), mix_p=float(rs.choice([0.0, 0.5, 1.0])))
        if rs.random() < 0.15:
            kw["grid_size"] = int(rs.choice([0, 5, 10, 33]))
        if kind == "RHMC":
            kw = {k: v for k, v in kw.items() if k in ("adaptive",)}
            kw.update(step_size=float(rs.choice([0.05, 0.11])), phi=float(rs.choice([math.pi / 2, 0.9])), tmax=float(rs.choice([10.0, 0.6])))
        if kind.startswith("SpeedUp"):          # exp(sqrt(d) v1 t): a fixed long horizon makes the grid bound fail forever (in the reference too)
            kw.pop("adaptive", None); kw.pop("tmax", None)
        cases.append((kind, dim, target, kw, int(rs.integers(1, 70)), int(rs.integers(2, 45)), int(rs.integers(1 << 30))))
    return cases


@pytest.mark.parametrize("kind,dim,target,kw,C,n_sk,seed", _random_cases(120, 20261003),
                         ids=lambda v: str(v).replace(" ", "") if not isinstance(v, dict) else ",".join(f"{k}={x}" for k, x in v.items()) or "-")
def test_random_configurations_match_checker(engine, oracle, kind, dim, target, kw, C, n_sk, seed):
    """Random sampler / target / dimension / keyword combinations (whatever kernel form the engine picks for them:
    register, LDS or wave-per-chain) against the checker, bit for bit."""
    # (rounds 1-4 skipped grid_size 5 / 33 for ForwardECMC / SpeedUpZigZag and beyond dim 256: the wave-per-chain kernel served those grids;
    #  round 5: grid_size is a run-time parameter of the group kernel, every combination runs)
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    x, v = make_init(kind, dim, C, seed=seed % 1000, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=seed, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, seed, diagnostics=True)
    assert_bitwise(got, ref)


@pytest.mark.parametrize("kind,dim,target,kw", [("ZigZagAD", 50, "iso", {}), ("ZigZagAD", 52, "banana", {}), ("ZigZagAD", 13, "diag", {}),
                                                ("BPS", 50, "banana", {}), ("BPS", 100, "iso", {"refresh_rate": 0.7}), ("BPSAD", 26, "diag", {}),
                                                ("Boomerang", 100, "iso", {"refresh_rate": 0.5}), ("BoomerangAD", 104, "iso", {"refresh_rate": 0.5}),
                                                ("Boomerang", 50, "diag", {})])
def test_lean_throughput_kernels(engine, oracle, monkeypatch, kind, dim, target, kw):
    """Round 3: a launch that records no diagnostic array and uses no time-horizon feature runs the LEAN instantiation of the FULL
    13-slot register kernel (the BASELINE shapes: d = 50 with four lanes per chain, d = 100 with eight, ...), which does not contain
    them.  Same trajectories as the full kernel (PDMP_NO_LEAN=1) and as the checker, hand-off between segments included."""
    monkeypatch.setenv("PDMP_LANE", "0")
    monkeypatch.setenv("PDMP_FORCE_SEGMENTS", "3")
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C, n_sk = 37, 70
    x, v = make_init(kind, dim, C, seed=8, target=target)
    lean = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=19, chain_id_offset=4, diagnostics=False)
    assert s.last_launch_form() == 3
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 19, chain_id_offset=4, diagnostics=False)
    assert_bitwise(lean, ref, keys=("X", "V", "t"))
    monkeypatch.setenv("PDMP_NO_LEAN", "1")
    full = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=19, chain_id_offset=4, diagnostics=False)
    assert_bitwise(full, ref, keys=("X", "V", "t"))


@pytest.mark.parametrize("kw", [{}, {"refresh_rate": 1.5}, {"signed_bound": False}, {"Gaussian_velocity": True, "refresh_rate": 0.7},
                                {"adaptive": False, "tmax": 0.3}])
@pytest.mark.parametrize("dim,chains,n_sk", [(40, 64, 60), (257, 70, 30), (600, 64, 16)])
@pytest.mark.parametrize("resync", [None, "1", "7"])
def test_dense_precision_bps_one_gemm_per_event(engine, oracle, monkeypatch, dim, chains, n_sk, kw, resync):
    """BPS on a dense precision with >= 64 chains (round 3, dense_batched.hip: dr_bps_kernel): the rate along the flow is
    <P x, v> + t <P v, v>, so a thinning iteration is two dot products and scalar arithmetic, and only an accepted event needs
    products with P -- P [x | v'] of all chains as ONE fp64 GEMM per skeleton column.  Against the reference-order restatement (a
    mat-vec per grid time) to a tolerance: the sums are ordered differently (rocBLAS' accumulation, A + B t instead of a fresh
    dot product per grid time); counters identical.  PDMP_DENSE_MODE=matvec keeps the bit-exact per-chain kernel."""
    from pdmpflux.jl_amd import _lib
    if resync:                                # events between two full products P [x | v] (default 64: in between only P v' is multiplied
        monkeypatch.setenv("PDMP_DENSE_RESYNC", resync)   # and P x follows the flows' recurrence); 1 = both halves at every event
    s, cfg = make_pair(engine, oracle, "BPS", dim, "dense", arith=oracle.ARITH_REFERENCE, **kw)
    x, v = make_init("BPS", dim, chains)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=21, diagnostics=True)
    assert _lib.lib().pdmp_dense_batched_last_mode() == 3 and s.last_launch_form() == 5
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 21, diagnostics=True)
    for k in ("rejected", "hitting_horizon", "errored_bound"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.allclose(got["t"], ref["t"], rtol=1e-9, atol=0)
    assert np.allclose(got["X"], ref["X"], rtol=1e-8, atol=1e-10)
    assert np.allclose(got["V"], ref["V"], rtol=1e-8, atol=1e-10)
    assert np.allclose(got["horizon"], ref["horizon"], rtol=1e-12) and np.allclose(got["ar"], ref["ar"], rtol=1e-7, atol=1e-12)


def test_dense_precision_bps_small_ensembles_keep_the_bit_exact_kernel(engine, oracle):
    s, cfg = make_pair(engine, oracle, "BPS", 40, "dense")
    x, v = make_init("BPS", 40, 6)
    got = engine.sample_skeleton_ensemble(s, 20, x, v, seed=21)
    assert s.last_launch_form() != 5
    assert_bitwise(got, oracle.sample_skeleton(cfg, 20, x, v, 21))


@pytest.mark.parametrize("kw", [{}, {"refresh_rate": 1.5}, {"signed_bound": False}, {"adaptive": False, "tmax": 0.3}])
@pytest.mark.parametrize("dim,chains,n_sk", [(40, 64, 60), (257, 70, 30), (600, 64, 16)])
@pytest.mark.parametrize("resync", [None, "1", "7"])
def test_dense_precision_boomerang_one_gemm_per_event(engine, oracle, monkeypatch, dim, chains, n_sk, kw, resync):
    """Boomerang on a dense precision with >= 64 chains (dense_batched.hip: dr_boom_kernel): grad-U_effective = P x - x is linear and
    the flow a rotation, so the rate along the flow is a quadratic form in cos t, sin t with four dot products as coefficients;
    one fp64 GEMM P [x | v'] over all chains per skeleton column.  Same tolerance as the BPS form."""
    from pdmpflux.jl_amd import _lib
    if resync:
        monkeypatch.setenv("PDMP_DENSE_RESYNC", resync)
    s, cfg = make_pair(engine, oracle, "BoomerangAD", dim, "dense", arith=oracle.ARITH_REFERENCE, **kw)
    x, v = make_init("BoomerangAD", dim, chains)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=21, diagnostics=True)
    assert _lib.lib().pdmp_dense_batched_last_mode() == 3 and s.last_launch_form() == 5
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 21, diagnostics=True)
    for k in ("rejected", "hitting_horizon", "errored_bound"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.allclose(got["t"], ref["t"], rtol=1e-9, atol=0)
    assert np.allclose(got["X"], ref["X"], rtol=1e-8, atol=1e-10)
    assert np.allclose(got["V"], ref["V"], rtol=1e-8, atol=1e-10)
    assert np.allclose(got["horizon"], ref["horizon"], rtol=1e-12) and np.allclose(got["ar"], ref["ar"], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("kind,dim,kw", [("BPS", 1100, {}), ("BPS", 2048, {"refresh_rate": 1.5}), ("BPS", 1537, {"signed_bound": False}),
                                         ("BoomerangAD", 1100, {}), ("BoomerangAD", 1536, {"refresh_rate": 1.5})])
def test_dense_precision_one_gemm_per_event_beyond_d_1024(engine, oracle, kind, dim, kw):
    """Round 4: the one-GEMM-per-event forms with 32 (BPS, d <= 2048) / 24 (Boomerang, d <= 1536) slots of v, P x, P v per lane; beyond
    that BPS falls back to the per-chain kernel and the Boomerang is refused at creation.  Same tolerances as at d <= 1024."""
    from pdmpflux.jl_amd import _lib
    s, cfg = make_pair(engine, oracle, kind, dim, "dense", arith=oracle.ARITH_REFERENCE, **kw)
    chains, n_sk = 64, 8
    x, v = make_init(kind, dim, chains)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=23, diagnostics=True)
    assert _lib.lib().pdmp_dense_batched_last_mode() == 3 and s.last_launch_form() == 5
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 23, diagnostics=True)
    for k in ("rejected", "hitting_horizon", "errored_bound"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.allclose(got["t"], ref["t"], rtol=1e-9, atol=0)
    assert np.allclose(got["X"], ref["X"], rtol=1e-8, atol=1e-10)
    assert np.allclose(got["V"], ref["V"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("dim,chains,n_sk,kw", [(40, 6, 20, {}), (40, 6, 20, {"refresh_rate": 1.5, "signed_bound": False}), (130, 3, 12, {"adaptive": False, "tmax": 0.3}),
                                                (1537, 64, 4, {})])
def test_dense_precision_boomerang_per_chain_kernel_is_bit_exact(engine, oracle, monkeypatch, dim, chains, n_sk, kw):
    """Round 4: the Boomerang on a dense precision also has the per-chain mat-vec form of Zig-Zag and BPS (the group kernel's dense
    target: one chain per wave, the checker's GROUP(64) order bit for bit) -- fewer than 64 chains, dimensions beyond the batched
    form's 1536, PDMP_DENSE_MODE=matvec, tape replays.  Its slope needs P v_t, a second product per time point: the flow rotates v."""
    s, cfg = make_pair(engine, oracle, "BoomerangAD", dim, "dense", **kw)
    x, v = make_init("BoomerangAD", dim, chains)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=21, diagnostics=True)
    assert s.last_launch_form() != 5
    assert_bitwise(got, oracle.sample_skeleton(cfg, n_sk, x, v, 21, diagnostics=True))
    if dim == 40:                                                          # the same sampler forced onto it with a larger ensemble
        monkeypatch.setenv("PDMP_DENSE_MODE", "matvec")
        x2, v2 = make_init("BoomerangAD", dim, 70)
        got2 = engine.sample_skeleton_ensemble(s, n_sk, x2, v2, seed=21)
        assert s.last_launch_form() != 5
        assert_bitwise(got2, oracle.sample_skeleton(cfg, n_sk, x2, v2, 21))


@pytest.mark.parametrize("target", ["iso", "banana"])
def test_group_kernel_full_machine_zigzag(engine, oracle, monkeypatch, target):
    """The group kernel on a launch that fills the machine (PDMP_LANE=0, 40 000 chains: 10 000 batches of 16 for 2 048 resident waves):
    dynamic (batch, segment) schedule, hand-off between segments, the bound in instalments of three cells -- the defaults of a full
    machine, which no small test launch takes any more.  A strided sample of chains bit for bit against the checker."""
    monkeypatch.setenv("PDMP_LANE", "0")
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, target)
    C, n_sk = 40_000, 80
    x, v = make_init("ZigZagAD", 50, C, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=31, diagnostics=False)
    assert s.last_launch_form() == 3
    for c in np.arange(0, C, 1733):
        r1 = oracle.sample_skeleton(cfg, n_sk, x[c:c + 1], v[c:c + 1], 31, chain_id_offset=int(c), diagnostics=False)
        for k in ("X", "V", "t"):
            assert np.array_equal(got[k][c], r1[k][0]), (k, c)
    assert (np.diff(got["t"], axis=1) > 0).all() and (np.abs(got["V"]) == 1.0).all()


@pytest.mark.parametrize("dim,G", [(6, None), (50, None), (50, 0), (20, 8)])
def test_diag_precisions_of_either_sign(engine, oracle, monkeypatch, dim, G):
    """Round 5: with every precision of the diagonal Gaussian >= 0 (the host looks once, when the handle is made) the Zig-Zag rate of a coordinate
    never decreases in t and the affine cell's left end is not evaluated; ONE negative or zero precision and the launch takes the general copy of
    the slot loop, max(v0, v1, 0).  Both against the checker, which always computes max(v0, v1, 0): the same doubles."""
    if G is not None:
        monkeypatch.setenv("PDMP_GROUP_LANES", str(G))
    rs = np.random.default_rng(dim)
    mu = rs.normal(size=dim)
    for flip in (False, True):
        prec = np.exp(rs.uniform(-1, 1, size=dim))
        if flip:
            prec[rs.integers(dim)] *= -0.3                 # a saddle direction: the run stays short, the rates stay finite
            prec[rs.integers(dim)] = 0.0
        s = engine.ZigZagAD(dim, engine.GaussianDiag(mu, prec), tmax=0.5)
        lanes = s.group_lanes()
        cfg = oracle.Config(sampler=oracle.ZIGZAG, dim=dim, grid_size=10, tmax=0.5, target_id=oracle.T_GAUSS_DIAG, target_params=np.concatenate([mu, prec]),
                            arith=oracle.ARITH_GROUP if lanes > 0 else oracle.ARITH_WAVE64, group_lanes=max(lanes, 1))
        x, v = make_init("ZigZagAD", dim, 37, seed=3)
        got = engine.sample_skeleton_ensemble(s, 60, x, v, seed=21, diagnostics=True)
        assert_bitwise(got, oracle.sample_skeleton(cfg, 60, x, v, 21, diagnostics=True))
