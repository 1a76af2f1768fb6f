"""The callers either side of the fixed-count path, against the checker's restatements:
  * sample_skeleton(sampler, T::Float64, ...) (src/sample.jl:323-439) -- events up to T, one closing column at t == T by the
    sampler's own flow, StickyPDMP masking (:386-401), zeroed statistics (:402-408), capacity doubling, ragged lengths;
  * _trim_history for an ensemble (Composites.jl:198-214) -- the compaction kernels (offsets by a wave prefix scan + dense copy);
  * the three sample_from_skeleton methods incl. the StickyPDMP ones (:516-561, :588-596) and (N, dt) (:649-682).
The engine side runs pdmp_sample_skeleton_device + pdmp_close_at_time_device + pdmp_compact_*_device through the C ABI."""
import numpy as np
import pytest

from helpers import DIAG_KEYS, make_init, make_pair

pytestmark = pytest.mark.gpu


def sticky_pair(P, O, dim, target="iso", kap=0.5, ad=True, **kw):
    rs = np.random.default_rng(dim * 11 + 3)
    kappa = kap * np.exp(rs.uniform(-0.5, 0.5, size=dim))
    if target == "iso":
        tg, tid, base = P.GaussianIso(), O.T_GAUSS_ISO, np.zeros(0)
    elif target == "diag":
        mu, prec = 0.3 * rs.normal(size=dim), np.exp(rs.uniform(-1, 1, size=dim))
        tg, tid, base = P.GaussianDiag(mu, prec), O.T_GAUSS_DIAG, np.concatenate([mu, prec])
    else:
        tg, tid, base = P.Banana(), O.T_BANANA, np.zeros(0)
    s = (P.StickyZigZagAD if ad else P.StickyZigZag)(dim, tg, kappa, **kw)
    cfg = O.Config(sampler=O.STICKY, dim=dim, grid_size=s.grid_size, tmax=s.tmax, refresh_rate=s.refresh_rate,
                   vectorized_bound=s.vectorized_bound, signed_bound=s.signed_bound, adaptive=s.adaptive, deriv_mode=s.deriv_mode,
                   target_id=tid, target_params=np.concatenate([base, kappa]), arith=O.ARITH_GROUP, group_lanes=s.group_lanes())
    return s, cfg


UNTIL_CASES = [
    ("ZigZagAD", 10, "iso", {}, 3.7), ("ZigZagAD", 50, "banana", {}, 2.2), ("ZigZag", 6, "diag", {}, 5.0),
    ("BPS", 8, "banana", {}, 3.1), ("BPSAD", 50, "iso", {}, 4.0),
    ("Boomerang", 12, "iso", {"refresh_rate": 0.5}, 6.3), ("BoomerangAD", 20, "diag", {"refresh_rate": 0.3}, 4.4),
    ("SpeedUpZigZagAD", 5, "iso", {}, 2.0), ("RHMC", 6, "banana", {}, 4.5), ("ForwardECMC", 9, "diag", {}, 3.3),
]


def _compare_until(P, O, s, cfg, x, v, T, seed, cap, keys):
    import torch
    from pdmpflux.jl_amd import ensemble
    dev = torch.device("cuda:0")
    got = ensemble.sample_skeleton_until_device(s, T, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=seed,
                                                diagnostics=True, init_capacity=cap)
    ref = O.sample_skeleton_until(cfg, T, x, v, seed, init_capacity=cap)
    cnt = got["count"].cpu().numpy()
    assert np.array_equal(cnt, ref["count"]) and (cnt >= 2).all()
    assert got["t"].shape[1] == ref["t"].shape[1]                       # the same number of capacity doublings
    for c in range(x.shape[0]):
        k = int(cnt[c])
        for key in keys:
            a, b = got[key][c, :k].cpu().numpy(), ref[key][c, :k]
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            assert same.all(), f"chain {c} {key}: first mismatch at {np.argwhere(~same)[0]} of {k} columns"
        assert got["t"][c, k - 1].item() == T                          # t[end] == T exactly (test/test_samplers.jl:56-62)
    return got, ref, cnt


@pytest.mark.parametrize("kind,dim,target,kw,T", UNTIL_CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}-T{c[4]}" for c in UNTIL_CASES])
def test_time_horizon_variant_matches_checker(engine, oracle, kind, dim, target, kw, T):
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C = 21
    x, v = make_init(kind, dim, C, seed=2, target=target)
    got, ref, cnt = _compare_until(engine, oracle, s, cfg, x, v, T, 9, 8, ("X", "V", "t") + DIAG_KEYS)
    assert len(set(cnt.tolist())) > 1                                   # ragged
    for c in range(C):                                                  # closing column = a non-event point (:402-408)
        k = int(cnt[c])
        assert got["ar"][c, k - 1].item() == 0.0 and got["rejected"][c, k - 1].item() == 0
        assert got["errored_bound"][c, k - 1].item() == 0 and got["hitting_horizon"][c, k - 1].item() == 0
        assert (got["error_value_ar"][c, k - 1] == 0).all()
        assert got["t"][c, k - 2].item() < T


@pytest.mark.parametrize("dim,target,kap,T", [(4, "iso", 0.5, 6.0), (13, "diag", 0.3, 4.0), (50, "iso", 0.7, 2.5), (20, "banana", 0.5, 3.0)])
def test_time_horizon_variant_sticky_matches_checker(engine, oracle, dim, target, kap, T):
    """StickyPDMP: the closing point flows with v_used = active_prev ? v_prev : 0 and records is_active = active_prev
    (sample.jl:386-401): a coordinate frozen at the last event stays exactly 0 and its recorded velocity is 0."""
    s, cfg = sticky_pair(engine, oracle, dim, target, kap)
    C = 33
    x, v = make_init("ZigZag", dim, C, seed=4, target=target)
    got, ref, cnt = _compare_until(engine, oracle, s, cfg, x, v, T, 17, 16, ("X", "V", "t", "is_active") + DIAG_KEYS)
    frozen_seen = 0
    for c in range(C):
        k = int(cnt[c])
        ia = got["is_active"][c, k - 1].cpu().numpy()
        assert np.array_equal(ia, got["is_active"][c, k - 2].cpu().numpy())
        fz = ia == 0
        frozen_seen += int(fz.sum())
        assert (got["X"][c, k - 1].cpu().numpy()[fz] == 0.0).all() and (got["V"][c, k - 1].cpu().numpy()[fz] == 0.0).all()
    assert frozen_seen > 0


def test_time_horizon_edge_cases(engine, oracle):
    """T == 0: the initial point alone (sample.jl:350-353); a capacity that is hit exactly; single-chain API: PDMPHistory."""
    import torch
    from pdmpflux.jl_amd import ensemble
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 3, "iso")
    x, v = np.zeros((5, 3)), np.ones((5, 3))
    dev = torch.device("cuda:0")
    got = ensemble.sample_skeleton_until_device(s, 0.0, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=1, diagnostics=True)
    ref = oracle.sample_skeleton_until(cfg, 0.0, x, v, 1)
    assert (got["count"].cpu().numpy() == 1).all() and (ref["count"] == 1).all()
    h = engine.sample_skeleton(s, 25.0, x[0], v[0], seed=42, init_capacity=8)          # several doublings
    r1 = oracle.sample_skeleton_until(cfg, 25.0, x[:1], v[:1], 42, init_capacity=8)
    k = int(r1["count"][0])
    assert len(h) == k and h.t[-1] == 25.0 and np.array_equal(h.t, r1["t"][0, :k]) and np.array_equal(h.X, r1["X"][0, :k].T)
    assert np.array_equal(h.V, r1["V"][0, :k].T) and np.array_equal(h.horizon, r1["horizon"][0, :k]) and np.array_equal(h.ar, r1["ar"][0, :k])
    smp = engine.sample_from_skeleton(s, 100, h)                                     # t[end] == T is the time scale of the samples
    assert np.array_equal(smp, oracle.sample_from_skeleton(cfg, 100, r1["X"][:, :k], r1["V"][:, :k], r1["t"][:, :k])[0].T)
    hs = engine.sample_skeleton(s, 5.0, x, v + 0.0, seed=1)                           # ensemble: a list of ragged PDMPHistory
    rs = oracle.sample_skeleton_until(cfg, 5.0, x, v, 1)
    assert [len(hh) for hh in hs] == rs["count"].tolist() and all(hh.t[-1] == 5.0 for hh in hs)
    with pytest.raises(engine.ArgumentError):
        engine.sample_skeleton(s, -1.0, x[0], v[0], seed=1)


@pytest.mark.parametrize("kind,dim,target", [("ZigZagAD", 10, "iso"), ("BPS", 7, "banana"), ("Boomerang", 65, "iso")])
def test_compaction_trims_and_packs(engine, oracle, kind, dim, target):
    """pdmp_compact_offsets_device / pdmp_compact_history_device (_trim_history, Composites.jl:198-214): offsets = exclusive prefix
    sum of count (also across the 1024-chain tiles of the scan), dense buffers = the concatenated valid prefixes."""
    import torch
    from pdmpflux.jl_amd import ensemble
    s, cfg = make_pair(engine, oracle, kind, dim, target)
    C, T = 2500, 1.3                                                     # three tiles of the offsets scan
    x, v = make_init(kind, dim, C, seed=6, target=target)
    dev = torch.device("cuda:0")
    xd, vd = torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev)
    rag = ensemble.sample_skeleton_until_device(s, T, xd, vd, seed=3, diagnostics=True, init_capacity=16)
    cnt = rag["count"].cpu().numpy()
    den = ensemble.compact_history_device(s, rag, rag["count"])
    off = den["offsets"].cpu().numpy()
    assert off[0] == 0 and np.array_equal(off[1:], np.cumsum(cnt)) and den["X"].shape[0] == cnt.sum()
    for key in ("X", "V", "t") + DIAG_KEYS:
        full, dense = rag[key].cpu().numpy(), den[key].cpu().numpy()
        want = np.concatenate([full[c, :cnt[c]] for c in range(C)], 0)
        assert np.array_equal(want, dense, equal_nan=True), key
    ref = oracle.sample_skeleton_until(cfg, T, x[:40], v[:40], 3, init_capacity=16)
    assert np.array_equal(ref["count"], cnt[:40])
    assert np.array_equal(den["t"][: off[40]].cpu().numpy(), np.concatenate([ref["t"][c, :cnt[c]] for c in range(40)]))
    # samples and fused moments straight from the ragged buffers (lens = count): t_end = T for every chain
    N = 50
    smp = ensemble.sample_from_skeleton_device(s, N, rag["X"], rag["V"], rag["t"], lens=rag["count"]).cpu().numpy()
    want = np.stack([oracle.sample_from_skeleton(cfg, N, ref["X"][c, :cnt[c]], ref["V"][c, :cnt[c]], ref["t"][c, :cnt[c]])[0] for c in range(40)])
    assert np.array_equal(smp[:40], want)
    s1, s2 = ensemble.sample_moments_device(s, N, rag["X"], rag["V"], rag["t"], lens=rag["count"])
    seq1, seq2 = np.zeros((40, dim)), np.zeros((40, dim))
    for j in range(N):
        seq1 += want[:, j]; seq2 += want[:, j] * want[:, j]
    assert np.array_equal(s1[:40].cpu().numpy(), seq1) and np.array_equal(s2[:40].cpu().numpy(), seq2)


def test_compaction_ignores_unfinished_chains(engine):
    import torch
    from pdmpflux.jl_amd import _lib
    import ctypes as C
    s = engine.ZigZagAD(2, engine.GaussianIso())
    dev = torch.device("cuda:0")
    count = torch.tensor([3, -5, 0, 2, 7], dtype=torch.int64, device=dev)
    off = torch.empty(6, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().pdmp_compact_offsets_device(s.handle(), 5, C.c_void_p(count.data_ptr()), C.c_void_p(off.data_ptr()), None))
    torch.cuda.synchronize()
    assert off.cpu().tolist() == [0, 3, 3, 3, 5, 12]


@pytest.mark.parametrize("dim,target,kap", [(2, "iso", 0.5), (5, "iso", 1.0), (13, "diag", 0.3), (50, "iso", 0.7), (20, "banana", 0.5)])
@pytest.mark.parametrize("discard", [True, False])
def test_sticky_sample_from_skeleton_masks_frozen_velocities(engine, oracle, dim, target, kap, discard):
    """sample_from_skeleton(sampler::StickyPDMP, N, history) (sample.jl:516-561): v_used = active ? v : 0.  Against the checker,
    and as a property: while a coordinate is frozen every sample of it is exactly 0 (the atom of the spike-and-slab target)."""
    s, cfg = sticky_pair(engine, oracle, dim, target, kap)
    C, n_sk, N = 9, 300, 700
    x, v = make_init("ZigZag", dim, C, seed=9, target=target)
    sk = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=51)
    got = engine.sample_from_skeleton_ensemble(s, N, sk["X"], sk["V"], sk["t"], discard_vt=discard, is_active=sk["is_active"])
    ref = oracle.sample_from_skeleton(cfg, N, sk["X"], sk["V"], sk["t"], discard_vt=discard, is_active=sk["is_active"])
    assert np.array_equal(got, ref)
    unmasked = oracle.sample_from_skeleton(oracle.Config(**{**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__ if k != "_keep"}, "sampler": oracle.ZIGZAG}),
                                           N, sk["X"], sk["V"], sk["t"], discard_vt=discard)
    assert not np.array_equal(got, unmasked)                             # the round-1 behaviour (plain linear flow) drifted off 0
    n_frozen = 0
    for c in range(C):
        t, ia = sk["t"][c], sk["is_active"][c]
        tm = np.arange(1, N + 1) * (t[-1] / N)
        seg = np.searchsorted(t, tm, side="right") - 1
        fz = ia[seg] == 0                                                # (N, d): sample j falls in a segment where coordinate i is frozen
        n_frozen += int(fz.sum())
        assert (got[c, :, :dim][fz] == 0.0).all()
        if not discard:
            assert (got[c, :, dim:2 * dim][fz] == 0.0).all()             # the reconstructed velocity is v_used
    assert n_frozen > 0
    with pytest.raises(engine.ArgumentError):                            # a Sticky history without is_active is refused, not mis-sampled
        engine.sample_from_skeleton_ensemble(s, N, sk["X"], sk["V"], sk["t"])


def test_sticky_fused_moments_and_dt_method(engine, oracle):
    import torch
    from pdmpflux.jl_amd import ensemble
    dim = 6
    s, cfg = sticky_pair(engine, oracle, dim, "iso", 0.4)
    C, n_sk, N = 70, 200, 333
    x, v = make_init("ZigZag", dim, C, seed=1)
    dev = torch.device("cuda:0")
    sk = ensemble.sample_skeleton_device(s, n_sk, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=8)
    assert sk["is_active"].dtype == torch.uint8 and (sk["is_active"] == 0).any()
    host = {k: sk[k].cpu().numpy() for k in ("X", "V", "t", "is_active")}
    want = oracle.sample_from_skeleton(cfg, N, host["X"], host["V"], host["t"], is_active=host["is_active"])
    smp = ensemble.sample_from_skeleton_device(s, N, sk["X"], sk["V"], sk["t"], is_active=sk["is_active"]).cpu().numpy()
    assert np.array_equal(smp, want)
    s1, s2 = ensemble.sample_moments_device(s, N, sk["X"], sk["V"], sk["t"], is_active=sk["is_active"])
    seq1, seq2 = np.zeros((C, dim)), np.zeros((C, dim))
    for j in range(N):
        seq1 += want[:, j]; seq2 += want[:, j] * want[:, j]
    assert np.array_equal(s1.cpu().numpy(), seq1) and np.array_equal(s2.cpu().numpy(), seq2)
    # dt::Float64 method on an ensemble: ragged sample counts
    dt = 0.21
    cap = int(np.floor(host["t"][:, -1].max() / dt)) + 3
    a, ca = engine.sample_from_skeleton_ensemble(s, cap, host["X"], host["V"], host["t"], dt=dt, is_active=host["is_active"],
                                                 discard_vt=False, return_counts=True)
    b, cb = oracle.sample_from_skeleton(cfg, cap, host["X"], host["V"], host["t"], dt=dt, is_active=host["is_active"],
                                        discard_vt=False, return_counts=True)
    assert np.array_equal(ca, cb) and np.array_equal(ca, np.floor(host["t"][:, -1] / dt).astype(np.int64)) and np.array_equal(a, b)
    h = engine.sample_skeleton(s, 120, x[0], v[0], seed=8)                # single-chain API, both methods
    one = engine.sample_from_skeleton(s, 0.3, h)
    assert one.shape == (dim, int(np.floor(h.t[-1] / 0.3)))
    assert np.array_equal(one, oracle.sample_from_skeleton(cfg, one.shape[1], h.X.T, h.V.T, h.t, dt=0.3, is_active=h.is_active.T)[0].T)


@pytest.mark.parametrize("kind,dim,target", [("ZigZagAD", 5, "iso"), ("BPS", 6, "banana"), ("Boomerang", 7, "iso"), ("Sticky", 6, "iso")])
def test_sample_from_skeleton_prefix_method(engine, oracle, kind, dim, target):
    """sample_from_skeleton(sampler, N::Int, dt::Float64, history) (sample.jl:649-682): the first N columns, samples at
    dt:dt:t[N], previous index by a STRICT comparison (searchsortedfirst - 1), v .* is_active for every sampler."""
    if kind == "Sticky":
        s, cfg = sticky_pair(engine, oracle, dim, target, 0.5)
        x, v = make_init("ZigZag", dim, 1, seed=3)
    else:
        s, cfg = make_pair(engine, oracle, kind, dim, target)
        x, v = make_init(kind, dim, 1, seed=3, target=target)
    h = engine.sample_skeleton(s, 150, x[0], v[0], seed=77)
    n_pre, dt = 90, 0.17
    for discard in (True, False):
        a = engine.sample_from_skeleton(s, n_pre, dt, h, discard_vt=discard)          # Julia's argument order
        M = int(np.floor(h.t[n_pre - 1] / dt))
        assert a.shape == ((dim if discard else 2 * dim + 1), M) and M > 5
        b = oracle.sample_from_skeleton(cfg, M, h.X.T, h.V.T, h.t, dt=dt, is_active=h.is_active.T, n_prefix=n_pre, discard_vt=discard)[0].T
        assert np.array_equal(a, b)
    # a sample time that coincides with an event time uses the PREVIOUS column (strict), unlike the dt::Float64 method
    dt2 = float(h.t[7])
    a = engine.sample_from_skeleton(s, n_pre, dt2, h)
    if kind in ("ZigZagAD", "Sticky"):
        va = h.V[:, 6] * h.is_active[:, 6]
        assert np.array_equal(a[:, 0], h.X[:, 6] + va * (dt2 - h.t[6]))
    assert np.array_equal(a, oracle.sample_from_skeleton(cfg, a.shape[1], h.X.T, h.V.T, h.t, dt=dt2, is_active=h.is_active.T, n_prefix=n_pre)[0].T)
    with pytest.raises(engine.ArgumentError):
        engine.sample_from_skeleton(s, 1000, dt, h)


def test_multi_device_collective_through_rccl_with_one_rank(engine, monkeypatch):
    """PDMP_MULTI_FORCE_RCCL: the library binds librccl at run time, builds a one-rank communicator (ncclCommInitAll) and runs the
    grouped in-place ncclAllGather of X, V, t -- all that a one-GPU box can exercise of the exchange; the result must be untouched."""
    import ctypes as C
    import torch
    from pdmpflux.jl_amd import _lib, ensemble
    monkeypatch.setenv("PDMP_MULTI_FORCE_RCCL", "1")
    L = _lib.lib()
    s = engine.ZigZagAD(20, engine.GaussianIso())
    Cn, n_sk, d = 37, 30, 20
    x, v = make_init("ZigZagAD", d, Cn, seed=8)
    dev = torch.device("cuda:0")
    single = ensemble.sample_skeleton_device(s, n_sk, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=3)
    m = C.c_void_p()
    conf = s._config()
    _lib.check(L.pdmp_multi_create(C.byref(conf), 1, None, C.byref(m)))
    try:
        X = torch.zeros((Cn, n_sk, d), dtype=torch.float64, device=dev); V = torch.zeros_like(X)
        t = torch.zeros((Cn, n_sk), dtype=torch.float64, device=dev)
        arr = lambda tt: (C.c_void_p * 1)(tt.data_ptr())
        _lib.check(L.pdmp_multi_sample_skeleton(m, Cn, n_sk, 0, C.c_void_p(x.ctypes.data), C.c_void_p(v.ctypes.data), C.c_uint64(3), 1, arr(X), arr(V), arr(t)))
        assert torch.equal(X, single["X"]) and torch.equal(V, single["V"]) and torch.equal(t, single["t"])
        km, xm = C.c_double(), C.c_double()
        _lib.check(L.pdmp_multi_get_timing(m, C.byref(km), C.byref(xm)))
        assert xm.value >= 0.0
    finally:
        L.pdmp_multi_destroy(m)


def test_multi_device_entry_with_one_device(engine, oracle):
    """pdmp_multi_* (one host process, n devices; include/pdmp_hip.h) with n_devices = 1 equals pdmp_sample_skeleton_device,
    with and without the gather flag, and the checker."""
    import ctypes as C
    import torch
    from pdmpflux.jl_amd import _lib, ensemble
    L = _lib.lib()
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    Cn, n_sk, d = 101, 40, 50
    x, v = make_init("ZigZagAD", d, Cn, seed=5)
    dev = torch.device("cuda:0")
    single = ensemble.sample_skeleton_device(s, n_sk, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=31, chain_id_offset=7)
    m = C.c_void_p()
    conf = s._config()
    _lib.check(L.pdmp_multi_create(C.byref(conf), 1, None, C.byref(m)))
    try:
        for gather in (0, 1):
            X = torch.zeros((Cn, n_sk, d), dtype=torch.float64, device=dev); V = torch.zeros_like(X)
            t = torch.zeros((Cn, n_sk), dtype=torch.float64, device=dev)
            arr = lambda tt: (C.c_void_p * 1)(tt.data_ptr())
            _lib.check(L.pdmp_multi_sample_skeleton(m, Cn, n_sk, 7, C.c_void_p(x.ctypes.data), C.c_void_p(v.ctypes.data), C.c_uint64(31),
                                                    gather, arr(X), arr(V), arr(t)))
            for a, b in ((X, single["X"]), (V, single["V"]), (t, single["t"])):
                assert torch.equal(a, b)
        km, xm = C.c_double(), C.c_double()
        _lib.check(L.pdmp_multi_get_timing(m, C.byref(km), C.byref(xm)))
        assert km.value > 0.0
        # the same step with the shard's initial states already on the device (no PCIe staging), and with the states generated there
        xd, vd = torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev)
        X = torch.zeros((Cn, n_sk, d), dtype=torch.float64, device=dev); V = torch.zeros_like(X)
        t = torch.zeros((Cn, n_sk), dtype=torch.float64, device=dev)
        _lib.check(L.pdmp_multi_sample_skeleton_device(m, Cn, n_sk, 7, arr(xd), arr(vd), C.c_uint64(31), 0, arr(X), arr(V), arr(t)))
        for a, b in ((X, single["X"]), (V, single["V"]), (t, single["t"])):
            assert torch.equal(a, b)
        gx, gv = torch.empty_like(xd), torch.empty_like(vd)
        _lib.check(L.pdmp_multi_init_states(m, Cn, 7, C.c_uint64(2024), 1.0, 0.1, 2, arr(gx), arr(gv)))
        rx, rv = oracle.init_states(2024, Cn, d, chain_id_offset=7, x_mean=1.0, x_scale=0.1, v_kind=2)
        assert np.array_equal(gx.cpu().numpy(), rx) and np.array_equal(gv.cpu().numpy(), rv)
    finally:
        L.pdmp_multi_destroy(m)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 31, chain_id_offset=7, diagnostics=False)
    assert np.array_equal(single["X"].cpu().numpy(), ref["X"]) and np.array_equal(single["t"].cpu().numpy(), ref["t"])
    bad = C.c_void_p()
    assert L.pdmp_multi_create(C.byref(conf), 99, None, C.byref(bad)) == _lib.ERR_ARGUMENT


@pytest.mark.parametrize("kind,T", [("ZigZagAD", 1.5), ("BPS", 4.0), ("BoomerangAD", 6.0)])
def test_time_horizon_on_the_batched_dense_forms(engine, oracle, kind, T):
    """`sample_skeleton(sampler, T::Float64, ...)` on the dense-precision target with >= 64 chains: the batched forms
    (dense_batched.hip: incremental Zig-Zag, one GEMM per event for BPS / Boomerang).  Rounds 1-3: no in-kernel stop there, the call
    took the capacity-doubling host loop over the fixed-count path + `pdmp_close_at_time_device`; round 4: the chunked path with the stop
    inside the kernels (test_gpu_time_horizon_chunks.py has the chunk structure under test).  Either way the sampler's own flow closes at
    t = T (the Boomerang's rotation included).  Against the checker's reference-order restatement: history lengths identical, times 1e-9."""
    import torch
    from pdmpflux.jl_amd import ensemble, _lib
    dim, C = 40, 64
    s, cfg = make_pair(engine, oracle, kind, dim, "dense", arith=oracle.ARITH_REFERENCE)
    x, v = make_init(kind, dim, C, seed=3)
    dev = torch.device("cuda:0")
    got = ensemble.sample_skeleton_until_device(s, T, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=11,
                                                init_capacity=32, compact=True)
    assert s.last_launch_form() == 5 and _lib.lib().pdmp_dense_batched_last_mode() in (2, 3)
    ref = oracle.sample_skeleton_until(cfg, T, x, v, 11, init_capacity=64)
    cnt, off = got["count"].cpu().numpy(), got["offsets"].cpu().numpy()
    assert np.array_equal(cnt, ref["count"])
    tt, X, V = got["t"].cpu().numpy(), got["X"].cpu().numpy(), got["V"].cpu().numpy()
    for c in range(C):
        k = int(cnt[c])
        assert np.allclose(tt[off[c]:off[c + 1]], ref["t"][c, :k], rtol=1e-9, atol=0)
        assert np.allclose(X[off[c]:off[c + 1]], ref["X"][c, :k], rtol=1e-8, atol=1e-10)
        assert np.allclose(V[off[c]:off[c + 1]], ref["V"][c, :k], rtol=1e-8, atol=1e-10)
        assert tt[off[c + 1] - 1] == T
