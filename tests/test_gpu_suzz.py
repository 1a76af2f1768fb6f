"""SpeedUpZigZag (src/Samplers/SpeedUpZigZagSamplers.jl: the Zig-Zag loop with the closed-form non-linear flow :71-79 and
grad-U_effective = speed grad-U - grad speed :81-83) on the group kernel against the checker, bit for bit."""
import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu

CASES = [
    ("SpeedUpZigZagAD", 2, "iso", {}),
    ("SpeedUpZigZagAD", 5, "iso", {}),
    ("SpeedUpZigZagAD", 13, "diag", {}),
    ("SpeedUpZigZagAD", 50, "iso", {}),                              # G = 4
    ("SpeedUpZigZagAD", 20, "banana", {}),
    ("SpeedUpZigZag", 6, "iso", {}),                                 # default AD_backend = "FiniteDiff"
    ("SpeedUpZigZagAD", 30, "iso", {"signed_bound": False}),
    ("SpeedUpZigZagAD", 9, "diag", {"vectorized_bound": False}),
    ("SpeedUpZigZagAD", 100, "iso", {"tmax": 0.5, "adaptive": False}),
    ("SpeedUpZigZagAD", 7, "iso", {"grid_size": 0}),                 # Brent constant bound
]


@pytest.mark.parametrize("kind,dim,target,kw", CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}-{'-'.join(f'{k}={v}' for k, v in c[3].items())}" for c in CASES])
def test_speed_up_zigzag_matches_checker(engine, oracle, kind, dim, target, kw):
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C, n_sk = 17, 40
    x, v = make_init(kind, dim, C, seed=6, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=31, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 31, diagnostics=True)
    assert_bitwise(got, ref)


def test_speed_up_zigzag_samples_use_the_nonlinear_flow(engine, oracle):
    """sample_from_skeleton interpolates with the sampler's own flow (sample.jl:499): bit for bit against the checker, and the
    samples of a long run have the moments of the target N(0, I) -- which checks flow and effective gradient together."""
    d, C = 4, 96
    s, cfg = make_pair(engine, oracle, "SpeedUpZigZagAD", d, "iso")
    x, v = make_init("SpeedUpZigZagAD", d, C, seed=2)
    sk = engine.sample_skeleton_ensemble(s, 3000, x, v, seed=12, diagnostics=False)
    smp = engine.sample_from_skeleton_ensemble(s, 3000, sk["X"], sk["V"], sk["t"])
    ref = oracle.sample_from_skeleton(cfg, 3000, sk["X"], sk["V"], sk["t"])
    assert np.array_equal(smp, ref)
    flat = smp.reshape(-1, d)
    assert np.abs(flat.mean(0)).max() < 0.05 and np.abs(flat.var(0) - 1).max() < 0.1
    with pytest.raises(engine.EngineError):                      # the fused moment sums are not implemented for this flow
        import torch
        from pdmpflux.jl_amd import ensemble
        dev = torch.device("cuda:0")
        ensemble.sample_moments_device(s, 10, torch.from_numpy(sk["X"]).to(dev), torch.from_numpy(sk["V"]).to(dev), torch.from_numpy(sk["t"]).to(dev))
