"""The checker's forward-mode restatement of create_gradient_function (src/ADBackend.jl:95-109) on the reference's own
test potentials (test/test_config.jl:33-46): gradients against central differences of the potential and against the
hand-derived banana gradient; slopes (the nested level) against differences of the gradient."""
import numpy as np
import pytest


@pytest.mark.parametrize("name", ["T_AD_BANANA", "T_AD_FUNNEL", "T_AD_RIDGED"])
def test_ad_gradient_matches_central_differences(oracle, name):
    d = 7
    rs = np.random.default_rng(3)
    cfg = oracle.Config(dim=d, target_id=getattr(oracle, name))
    for _ in range(5):
        x = rs.normal(size=d)
        x[0] = 1.5 + 0.3 * x[0]
        g = oracle.grad(cfg, x)
        fd = np.array([(oracle.potential(cfg, x + h) - oracle.potential(cfg, x - h)) / 2e-6 for h in 1e-6 * np.eye(d)])
        np.testing.assert_allclose(g, fd, rtol=2e-6, atol=2e-8)


def test_ad_banana_equals_hand_derived_gradient(oracle):
    """U_banana differentiated by forward mode reproduces the analytic gradient of the built-in banana target bit for bit
    here (same products, the zero partials add exact zeros)."""
    d = 9
    rs = np.random.default_rng(4)
    a, b = oracle.Config(dim=d, target_id=oracle.T_AD_BANANA), oracle.Config(dim=d, target_id=oracle.T_BANANA)
    for _ in range(10):
        x = rs.normal(size=d)
        np.testing.assert_allclose(oracle.grad(a, x), oracle.grad(b, x), rtol=1e-14, atol=1e-15)


@pytest.mark.parametrize("name", ["T_AD_BANANA", "T_AD_FUNNEL", "T_AD_RIDGED"])
def test_ad_chain_runs_and_time_increases(oracle, name):
    d = 6
    rs = np.random.default_rng(5)
    x = rs.normal(size=(3, d)); x[:, 0] = 2.0 + 0.1 * x[:, 0]
    for sampler, kw in ((oracle.ZIGZAG, {}), (oracle.BPS, dict(tmax=1.0, refresh_rate=0.1, vectorized_bound=False))):
        cfg = oracle.Config(sampler=sampler, dim=d, target_id=getattr(oracle, name), **kw)
        v = np.ones((3, d)) if sampler == oracle.ZIGZAG else rs.normal(size=(3, d))
        r = oracle.sample_skeleton(cfg, 60, x, v, 9)
        assert np.all(np.diff(r["t"], axis=1) > 0) and np.isfinite(r["X"]).all()
