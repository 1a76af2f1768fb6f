"""CPU checks of the checker's ForwardECMC (ForwardEventChainMonteCarlo.jl:127-175) and SpeedUpZigZag
(SpeedUpZigZagSamplers.jl:71-83) restatements and of its deterministic exp: properties the reference's own tests rely on
(unit speed, target moments, error messages) plus flow identities."""
import ctypes
import math

import numpy as np
import pytest


def test_exp_within_one_ulp(oracle):
    L = oracle.lib()
    L.orc_exp.restype = ctypes.c_double
    L.orc_exp.argtypes = [ctypes.c_double]
    rs = np.random.default_rng(0)
    xs = np.concatenate([rs.uniform(-700, 700, 4000), rs.uniform(-1, 1, 4000), [0.0, 709.7, -745.0, 0.3465, 0.35, 1.03, 1.04, -1.04]])
    for x in xs:
        a, b = L.orc_exp(float(x)), math.exp(float(x))
        assert abs(a - b) <= abs(b) * 2.3e-16 * 1.01 + 5e-324
    assert L.orc_exp(710.0) == math.inf and L.orc_exp(-746.0) == 0.0 and math.isnan(L.orc_exp(math.nan))


@pytest.mark.parametrize("kw", [{}, {"ran_p": True}, {"switch": False}, {"positive": False}])
def test_forward_ecmc_unit_speed_and_gaussian_moments(oracle, kw):
    d, C = 5, 24
    rs = np.random.default_rng(1)
    x = rs.normal(size=(C, d)); v = rs.normal(size=(C, d)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    cfg = oracle.Config(sampler=oracle.FECMC, dim=d, target_id=oracle.T_GAUSS_ISO, tmax=2.0, vectorized_bound=False, **kw)
    r = oracle.sample_skeleton(cfg, 1500, x, v, 7)
    assert np.abs(np.linalg.norm(r["V"], axis=2) - 1).max() < 1e-12                    # ForwardECMC keeps |v| = 1
    smp = oracle.sample_from_skeleton(cfg, 1500, r["X"], r["V"], r["t"]).reshape(-1, d)
    assert np.abs(smp.mean(0)).max() < 0.12 and np.abs(smp.var(0) - 1).max() < 0.2


def test_forward_ecmc_needs_two_dimensions(oracle):
    cfg = oracle.Config(sampler=oracle.FECMC, dim=1, target_id=oracle.T_GAUSS_ISO, vectorized_bound=False)
    with pytest.raises(oracle.OracleError, match="at least 2"):                         # ForwardEventChainMonteCarlo.jl:298-300
        oracle.sample_skeleton(cfg, 5, np.zeros((1, 1)), np.ones((1, 1)), 1)


def test_speed_up_zigzag_flow_is_a_flow(oracle):
    """sample_from_skeleton interpolates with the sampler's flow: on a two-point history (one segment) the closed form of
    SpeedUpZigZagSamplers.jl:71-79 must start at x, and composing it (restart from an interpolated point with the same
    velocity) must land on the same curve: phi_t(phi_s(x)) = phi_{s+t}(x)."""
    d = 4
    rs = np.random.default_rng(2)
    cfg = oracle.Config(sampler=oracle.SUZZ, dim=d, target_id=oracle.T_GAUSS_ISO)
    for _ in range(5):
        x, v = rs.normal(size=d), rs.choice([-1.0, 1.0], size=d)
        T = 0.8
        X = np.stack([x, x])[None]; V = np.stack([v, v])[None]; t = np.array([[0.0, T]])
        pts = oracle.sample_from_skeleton(cfg, 8, X, V, t)[0]                            # phi at T/8, ..., T  (the last one from x again with tau = 0 ... see below)
        tiny = oracle.sample_from_skeleton(cfg, 1, X, V, np.array([[0.0, 1e-12]]))[0, 0]
        np.testing.assert_allclose(tiny, x, atol=1e-10)                                  # phi_0 = identity
        mid = pts[3]                                                                     # phi_{T/2}(x)   (sample 4 of 8 lies on the first segment)
        X2 = np.stack([mid, mid])[None]
        far = oracle.sample_from_skeleton(cfg, 2, X2, V, np.array([[0.0, T / 2]]))[0]   # phi_{T/4}(mid) = phi_{3T/4}(x)
        np.testing.assert_allclose(far[0], pts[5], rtol=1e-9, atol=1e-11)


def test_speed_up_zigzag_samples_the_gaussian(oracle):
    d, C = 3, 24
    rs = np.random.default_rng(3)
    x, v = rs.normal(size=(C, d)), rs.choice([-1.0, 1.0], size=(C, d))
    cfg = oracle.Config(sampler=oracle.SUZZ, dim=d, target_id=oracle.T_GAUSS_ISO)
    r = oracle.sample_skeleton(cfg, 2500, x, v, 5)
    assert np.all(np.diff(r["t"], axis=1) > 0)
    smp = oracle.sample_from_skeleton(cfg, 2500, r["X"], r["V"], r["t"]).reshape(-1, d)
    assert np.abs(smp.mean(0)).max() < 0.1 and np.abs(smp.var(0) - 1).max() < 0.15


def test_sticky_zigzag_atom_weight_and_frozen_coordinates(oracle):
    """StickyZigZag (StickySamplingLoop.jl): frozen coordinates sit exactly at 0, time never runs backwards, and the fraction of time
    a coordinate spends frozen is the weight of the spike in the spike-and-slab target: pi(0)/kappa / (1 + pi(0)/kappa)."""
    d, C = 3, 24
    rs = np.random.default_rng(6)
    x, v = rs.normal(size=(C, d)), rs.choice([-1.0, 1.0], size=(C, d))
    for kap in (0.5, 2.0):
        cfg = oracle.Config(sampler=oracle.STICKY, dim=d, target_id=oracle.T_GAUSS_ISO, target_params=np.full(d, kap))
        r = oracle.sample_skeleton(cfg, 4000, x, v, 3)
        ia, X, t = r["is_active"], r["X"], r["t"]
        assert np.all(X[ia == 0] == 0.0) and np.all(np.diff(t, axis=1) >= 0)
        dt = np.diff(t, axis=1)
        frozen = ((1 - ia[:, :-1, :]) * dt[:, :, None]).sum(1) / t[:, -1][:, None]
        w = (0.3989422804 / kap) / (1 + 0.3989422804 / kap)
        assert abs(frozen.mean() - w) < 0.05        # (the reference's thaw drops the time already moved, :160-161: a small bias at large kappa)
