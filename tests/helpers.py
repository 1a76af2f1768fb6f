"""Shared helpers for parity tests: build matching (engine sampler, oracle config) pairs and inputs."""
import numpy as np

SAMPLER_IDS = {"ZigZag": 0, "BPS": 1, "Boomerang": 2}


def make_pair(P, O, kind, dim, target="iso", arith=None, **kw):
    """Returns (engine sampler, oracle Config) built from the same reference-style kwargs."""
    rs = np.random.default_rng(dim * 7 + 1)
    if target == "iso":
        tg, tid, params = P.GaussianIso(), O.T_GAUSS_ISO, None
    elif target == "diag":
        mu, prec = rs.normal(size=dim), np.exp(rs.uniform(-1, 1, size=dim))
        tg, tid, params = P.GaussianDiag(mu, prec), O.T_GAUSS_DIAG, np.concatenate([mu, prec])
    elif target == "banana":
        tg, tid, params = P.Banana(), O.T_BANANA, None
    elif target == "dense":                      # P = Q diag(lambda) Q', lambda log-uniform in [0.1, 10] (BASELINE config 4)
        Q, _ = np.linalg.qr(rs.normal(size=(dim, dim)))
        Pm = (Q * np.exp(rs.uniform(np.log(0.1), np.log(10.0), size=dim))) @ Q.T
        Pm = 0.5 * (Pm + Pm.T)
        tg, tid, params = P.GaussianPrecision(Pm), O.T_GAUSS_DENSE, np.asfortranarray(Pm).ravel(order="F")
    elif target in ("src_banana", "src_funnel", "src_ridged"):    # potentials given as source vs the checker's forward-mode restatement
        src, tid = {"src_banana": (P.U_BANANA_SOURCE, O.T_AD_BANANA), "src_funnel": (P.U_FUNNEL_SOURCE, O.T_AD_FUNNEL),
                    "src_ridged": (P.U_RIDGED_GAUSS_SOURCE, O.T_AD_RIDGED)}[target]
        tg, params = P.SourcePotential(src), None
    else:
        raise ValueError(target)
    ctor = {"ZigZag": P.ZigZag, "ZigZagAD": P.ZigZagAD, "BPS": P.BPS, "BPSAD": P.BPSAD,
            "Boomerang": P.Boomerang, "BoomerangAD": P.BoomerangAD, "ForwardECMC": P.ForwardECMC,
            "ForwardECMCAD": P.ForwardECMCAD, "SpeedUpZigZag": P.SpeedUpZigZag, "SpeedUpZigZagAD": P.SpeedUpZigZagAD,
            "RHMC": P.RHMC}[kind]
    s = ctor(dim, tg, **kw)
    group = 1
    if arith is None:                       # match the summation order the engine uses for this configuration
        G = s.group_lanes()
        arith, group = (O.ARITH_GROUP, G) if G > 0 else (O.ARITH_WAVE64, 1)
    cfg = O.Config(sampler=s.sampler_id, dim=dim, grid_size=s.grid_size, tmax=s.tmax, refresh_rate=s.refresh_rate,
                   vectorized_bound=s.vectorized_bound, signed_bound=s.signed_bound, adaptive=s.adaptive,
                   gaussian_velocity=s.gaussian_velocity, deriv_mode=s.deriv_mode, target_id=tid,
                   target_params=params, arith=arith, group_lanes=group,
                   mix_p=getattr(s, "mix_p", 0.5), ran_p=getattr(s, "ran_p", False), switch=getattr(s, "switch", True),
                   positive=getattr(s, "positive", True), step_size=getattr(s, "step_size", 0.05), phi=getattr(s, "phi", 1.5707963267948966), speed_factor=getattr(s, "speed_factor", 1.0),
                   rng_version=getattr(s, "rng_version", 0))
    return s, cfg


def make_init(kind, dim, C, seed=0, target="iso"):
    rs = np.random.default_rng(seed + 1000 * dim + C)
    x = rs.normal(size=(C, dim))
    if target in ("banana", "src_banana"):
        x = 1.0 + 0.1 * x
    if target == "src_funnel":                   # U_funnel lives on x[1] > 0 (log(v), test/test_config.jl:38-42)
        x[:, 0] = 2.0 + 0.2 * x[:, 0]
    if kind.startswith(("ZigZag", "SpeedUpZigZag")):
        v = rs.choice([-1.0, 1.0], size=(C, dim))
    elif kind.startswith("BPS") or kind.startswith("ForwardECMC"):
        v = rs.normal(size=(C, dim))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    else:
        v = rs.normal(size=(C, dim))
    return x, v


DIAG_KEYS = ("horizon", "ar", "errored_bound", "error_value_ar", "rejected", "hitting_horizon")


def assert_bitwise(got, ref, keys=("X", "V", "t") + DIAG_KEYS):
    for k in keys:
        a, b = np.asarray(got[k]), np.asarray(ref[k])
        assert a.shape == b.shape, k
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        if not same.all():
            idx = np.argwhere(~same)[0]
            raise AssertionError(f"{k}: first mismatch at {tuple(idx)}: {a[tuple(idx)]!r} vs {b[tuple(idx)]!r} "
                                 f"({(~same).sum()} of {same.size} differ)")
