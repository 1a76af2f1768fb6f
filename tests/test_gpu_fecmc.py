"""ForwardECMC (src/Samplers/ForwardEventChainMonteCarlo.jl: flow, rate and bound of BPS without refreshment, event-chain
velocity jump :127-175 with _orthogonal_switch / _full_refresh) on the group kernel against the checker, bit for bit."""
import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu

CASES = [
    ("ForwardECMC", 2, "iso", {}),                       # dim == 2: mix_p forced to 0 (:301-303)
    ("ForwardECMC", 3, "iso", {}),
    ("ForwardECMC", 10, "iso", {}),
    ("ForwardECMC", 50, "iso", {}),                      # G = 4
    ("ForwardECMC", 50, "banana", {}),
    ("ForwardECMC", 30, "diag", {"ran_p": True}),
    ("ForwardECMC", 20, "banana", {"switch": False}),
    ("ForwardECMC", 17, "iso", {"positive": False, "ran_p": True}),
    ("ForwardECMC", 40, "iso", {"mix_p": 1.0}),
    ("ForwardECMC", 40, "iso", {"mix_p": 0.0}),
    ("ForwardECMC", 12, "banana", {"AD_backend": "FiniteDiff", "signed_bound": False}),
    ("ForwardECMC", 100, "diag", {"tmax": 0.0}),         # G = 8
    ("ForwardECMCAD", 9, "iso", {"grid_size": 0}),       # Brent constant bound
    ("ForwardECMC", 300, "iso", {}),                     # K > 13: G = 32
    ("ForwardECMC", 11, "banana", {"speed_factor": 1.7}),  # _velocity_jump_event_chain_speed_up, :177-218
    ("ForwardECMC", 24, "iso", {"speed_factor": 0.6, "ran_p": True}),
]


@pytest.mark.parametrize("kind,dim,target,kw", CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}-{'-'.join(f'{k}={v}' for k, v in c[3].items())}" for c in CASES])
def test_forward_ecmc_matches_checker(engine, oracle, kind, dim, target, kw):
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C, n_sk = 19, 50
    x, v = make_init(kind, dim, C, seed=4, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=23, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 23, diagnostics=True)
    assert_bitwise(got, ref)


def test_forward_ecmc_keeps_unit_speed_and_samples_the_gaussian(engine):
    d, C = 6, 256
    s = engine.ForwardECMC(d, engine.GaussianIso())
    rs = np.random.default_rng(1)
    x = rs.normal(size=(C, d)); v = rs.normal(size=(C, d)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    sk = engine.sample_skeleton_ensemble(s, 1500, x, v, seed=3, diagnostics=False)
    assert np.abs(np.linalg.norm(sk["V"], axis=2) - 1).max() < 1e-12
    smp = engine.sample_from_skeleton_ensemble(s, 1500, sk["X"], sk["V"], sk["t"]).reshape(-1, d)
    assert np.abs(smp.mean(0)).max() < 0.05 and np.abs(smp.var(0) - 1).max() < 0.08


def test_forward_ecmc_argument_errors(engine):
    with pytest.raises(engine.ArgumentError):
        engine.ForwardECMC(1, engine.GaussianIso())                                  # ForwardEventChainMonteCarlo.jl:298-300
    with pytest.raises(engine.ArgumentError):
        engine.ForwardECMC(5, engine.GaussianIso(), normal=True)
