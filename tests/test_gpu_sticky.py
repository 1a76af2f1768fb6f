"""StickyZigZag (src/Samplers/StickyZigZagSamplers.jl with the loop of src/StickySamplingLoop.jl: coordinates stick at 0 and thaw
at rates kappa; flips, sticking and thawing are all events) on the group kernel against the checker, bit for bit, including
PDMPHistory.is_active."""
import numpy as np
import pytest

from helpers import DIAG_KEYS, assert_bitwise, make_init

pytestmark = pytest.mark.gpu

CASES = [
    ("StickyZigZagAD", 2, "iso", 0.5, {}),
    ("StickyZigZagAD", 5, "iso", 1.0, {}),
    ("StickyZigZagAD", 13, "diag", 0.3, {}),
    ("StickyZigZagAD", 50, "iso", 0.7, {}),                          # G = 4
    ("StickyZigZagAD", 20, "banana", 0.5, {}),
    ("StickyZigZag", 6, "iso", 0.4, {}),                             # default AD_backend = "FiniteDiff"
    ("StickyZigZagAD", 30, "iso", 2.0, {"signed_bound": False}),
    ("StickyZigZagAD", 9, "diag", 0.2, {"vectorized_bound": False}),
    ("StickyZigZagAD", 100, "iso", 0.05, {"tmax": 0.5, "adaptive": False}),
    ("StickyZigZagAD", 7, "iso", 0.6, {"grid_size": 0}),             # Brent constant bound
]


def pair(P, O, kind, dim, target, kap, **kw):
    rs = np.random.default_rng(dim * 11 + 3)
    kappa = kap * np.exp(rs.uniform(-0.5, 0.5, size=dim))
    if target == "iso":
        tg, tid, base = P.GaussianIso(), O.T_GAUSS_ISO, np.zeros(0)
    elif target == "diag":
        mu, prec = 0.3 * rs.normal(size=dim), np.exp(rs.uniform(-1, 1, size=dim))
        tg, tid, base = P.GaussianDiag(mu, prec), O.T_GAUSS_DIAG, np.concatenate([mu, prec])
    else:
        tg, tid, base = P.Banana(), O.T_BANANA, np.zeros(0)
    s = getattr(P, kind)(dim, tg, kappa, **kw)
    G = s.group_lanes()
    cfg = O.Config(sampler=O.STICKY, dim=dim, grid_size=s.grid_size, tmax=s.tmax, refresh_rate=s.refresh_rate,
                   vectorized_bound=s.vectorized_bound, signed_bound=s.signed_bound, adaptive=s.adaptive, deriv_mode=s.deriv_mode,
                   target_id=tid, target_params=np.concatenate([base, kappa]), arith=O.ARITH_GROUP, group_lanes=G)
    return s, cfg


@pytest.mark.parametrize("kind,dim,target,kap,kw", CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}-k{c[3]}-{'-'.join(f'{k}={v}' for k, v in c[4].items())}" for c in CASES])
def test_sticky_zigzag_matches_checker(engine, oracle, kind, dim, target, kap, kw):
    s, cfg = pair(engine, oracle, kind, dim, target, kap, **kw)
    C, n_sk = 17, 80
    x, v = make_init("ZigZag", dim, C, seed=9, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=51, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 51, diagnostics=True)
    assert_bitwise(got, ref, keys=("X", "V", "t", "is_active") + DIAG_KEYS)
    assert (ref["is_active"] == 0).any()                               # the run does stick


def test_sticky_zigzag_freezes_at_zero_and_thaws(engine):
    d, C = 4, 64
    s = engine.StickyZigZagAD(d, engine.GaussianIso(), np.full(d, 0.5))
    rs = np.random.default_rng(0)
    sk = engine.sample_skeleton_ensemble(s, 3000, rs.normal(size=(C, d)), rs.choice([-1.0, 1.0], size=(C, d)), seed=2, diagnostics=True)
    ia, X, t = sk["is_active"], sk["X"], sk["t"]
    assert np.all(X[ia == 0] == 0.0) and np.all(np.diff(t, axis=1) >= 0)
    dt = np.diff(t, axis=1)
    frozen = ((1 - ia[:, :-1, :]) * dt[:, :, None]).sum(1) / t[:, -1][:, None]
    # spike-and-slab weight of the atom at 0: pi(0)/kappa / (1 + pi(0)/kappa) with pi(0) = 1/sqrt(2 pi), kappa = 0.5
    w = (0.3989422804 / 0.5) / (1 + 0.3989422804 / 0.5)
    assert abs(frozen.mean() - w) < 0.03
    h = engine.sample_skeleton(s, 50, np.zeros(d) + 0.3, np.ones(d), seed=4)
    assert h.is_active.shape == (d, 50) and h.is_active.dtype == bool


@pytest.mark.parametrize("segments", [3, 7])
@pytest.mark.parametrize("dim,target", [(5, "iso"), (50, "iso"), (20, "banana")])
def test_sticky_zigzag_segment_handoff(engine, oracle, monkeypatch, segments, dim, target):
    """Round 3: the hand-off state carries the active masks of a chain's lanes, so StickyZigZag runs in segments (the dynamic
    schedule of large ensembles) like every other sampler: a coordinate frozen at the end of a segment is still frozen -- and thaws
    at the same event -- in the next one."""
    monkeypatch.setenv("PDMP_FORCE_SEGMENTS", str(segments))
    s, cfg = pair(engine, oracle, "StickyZigZagAD", dim, target, 0.5)
    C, n_sk = 41, 120
    x, v = make_init("ZigZag", dim, C, seed=9, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=51, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 51, diagnostics=True)
    assert_bitwise(got, ref, keys=("X", "V", "t", "is_active") + DIAG_KEYS)
    assert (ref["is_active"] == 0).any()


@pytest.mark.parametrize("dim,target,kap,seg", [(2, "iso", 0.5, 0), (3, "banana", 0.6, 0), (13, "diag", 0.3, 3), (20, "banana", 0.5, 0), (24, "iso", 0.4, 5),
                                                (50, "iso", 0.7, 4), (50, "banana", 0.5, 0), (100, "iso", 0.2, 0), (7, "iso", 0.6, 0), (112, "diag", 0.3, 2)])
def test_sticky_register_and_lds_forms_agree(engine, oracle, monkeypatch, dim, target, kap, seg):
    """Round 4: StickyZigZag in the register form (x, v and the active mask in VGPRs; exact-K kernels for K <= 7 slots per lane,
    inst_reg_sticky_k*.hip; forward-mode slopes, the default bound flags): the same bits as the LDS form and the checker, hand-off
    segments (the masks travel in the hand-off state) included."""
    if seg:
        monkeypatch.setenv("PDMP_FORCE_SEGMENTS", str(seg))
    outs = []
    for reg in ("1", "0"):
        monkeypatch.setenv("PDMP_REG", reg)
        s, cfg = pair(engine, oracle, "StickyZigZagAD", dim, target, kap)
        K = -(-dim // s.group_lanes())
        assert K <= 7
        x, v = make_init("ZigZag", dim, 37, seed=5, target=target)
        outs.append(engine.sample_skeleton_ensemble(s, 90, x, v, seed=13, diagnostics=True))
        assert s.last_launch_form() == (3 if reg == "1" else 2)
    assert_bitwise(outs[0], outs[1], keys=("X", "V", "t", "is_active") + DIAG_KEYS)
    assert_bitwise(outs[0], oracle.sample_skeleton(cfg, 90, x, v, 13, diagnostics=True), keys=("X", "V", "t", "is_active") + DIAG_KEYS)
    assert (outs[0]["is_active"] == 0).any()
