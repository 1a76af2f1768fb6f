"""RHMC (src/Samplers/RandomizedHamiltonianMonteCarlo.jl: velocity-Verlet flow :97-130, constant refresh clock, Horowitz
refreshment :144-149) on the group kernel against the checker, bit for bit."""
import math

import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu

CASES = [
    ("RHMC", 2, "iso", {}),
    ("RHMC", 10, "iso", {}),
    ("RHMC", 50, "iso", {}),                                          # G = 4
    ("RHMC", 13, "diag", {"phi": 0.7}),
    ("RHMC", 20, "banana", {"step_size": 0.013}),
    ("RHMC", 6, "banana", {"mean_duration": 0.37, "tmax": 0.5}),       # horizon hits: the flow to the horizon is integrated too
    ("RHMC", 100, "diag", {"adaptive": True, "tmax": 0.4, "refresh_rate": 2.5}),
    ("RHMC", 300, "iso", {"step_size": 0.2}),
]


@pytest.mark.parametrize("kind,dim,target,kw", CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}-{'-'.join(f'{k}={v}' for k, v in c[3].items())}" for c in CASES])
def test_rhmc_matches_checker(engine, oracle, kind, dim, target, kw):
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C, n_sk = 15, 30
    x, v = make_init(kind, dim, C, seed=8, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=41, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 41, diagnostics=True)
    assert_bitwise(got, ref)


def test_rhmc_event_states_have_the_target_moments(engine):
    """With phi = pi/2 every event draws a fresh velocity and the event times are a Poisson clock: the positions at the events
    of a long run are (up to the integrator's O(h^2) error) draws from the target, velocities are N(0, I)."""
    d, C = 3, 200
    s = engine.RHMC(d, engine.GaussianIso())
    rs = np.random.default_rng(0)
    sk = engine.sample_skeleton_ensemble(s, 600, rs.normal(size=(C, d)), rs.normal(size=(C, d)), seed=2, diagnostics=True)
    X, V = sk["X"][:, 50:].reshape(-1, d), sk["V"][:, 50:].reshape(-1, d)
    assert np.abs(X.mean(0)).max() < 0.03 and np.abs(X.var(0) - 1).max() < 0.05
    assert np.abs(V.mean(0)).max() < 0.03 and np.abs(V.var(0) - 1).max() < 0.05
    assert np.all(sk["ar"][:, 1:] == 1.0)                                 # rejection-free: the clock's rate is its own bound
    dt = np.diff(sk["t"], axis=1)
    assert abs(dt.mean() - 1.0) < 0.02                                    # Exp(refresh_rate = 1) waiting times


def test_rhmc_argument_errors(engine):
    for kw in ({"refresh_rate": 0.0}, {"phi": 0.0}, {"phi": 2.0}, {"step_size": -1.0}, {"mean_duration": 0.0}, {"tmax": math.inf}):
        with pytest.raises(engine.ArgumentError):                        # RandomizedHamiltonianMonteCarlo.jl:59-89
            engine.RHMC(3, engine.GaussianIso(), **kw)


@pytest.mark.parametrize("dim,target,kw", [(6, "iso", {}), (70, "diag", {"phi": 1.0}), (9, "banana", {"step_size": 0.02}), (200, "iso", {})])
def test_rhmc_samples_integrate_the_flow(engine, oracle, dim, target, kw):
    """sample_from_skeleton interpolates with the sampler's own flow (sample.jl:499): velocity-Verlet from the last event,
    bit for bit against the checker (positions and, with discard_vt = false, velocities and times)."""
    s, cfg = make_pair(engine, oracle, "RHMC", dim, target, **kw)
    C = 11
    x, v = make_init("RHMC", dim, C, seed=5, target=target)
    sk = engine.sample_skeleton_ensemble(s, 25, x, v, seed=6, diagnostics=False)
    for discard in (True, False):
        got = engine.sample_from_skeleton_ensemble(s, 40, sk["X"], sk["V"], sk["t"], discard_vt=discard)
        ref = oracle.sample_from_skeleton(cfg, 40, sk["X"], sk["V"], sk["t"], discard_vt=discard)
        assert np.array_equal(got, ref)


def test_rhmc_samples_have_the_target_moments(engine):
    d, C = 3, 128
    s = engine.RHMC(d, engine.Banana())
    rs = np.random.default_rng(4)
    sk = engine.sample_skeleton_ensemble(s, 800, 1 + 0.1 * rs.normal(size=(C, d)), rs.normal(size=(C, d)), seed=3, diagnostics=False)
    smp = engine.sample_from_skeleton_ensemble(s, 800, sk["X"], sk["V"], sk["t"])[:, 100:].reshape(-1, d)
    assert np.abs(smp.mean(0)).max() < 0.08                                  # banana: E x = 0, Var x1 = 1, Var x2 = 3, Var x3 = 1
    assert abs(smp[:, 0].var() - 1) < 0.08 and abs(smp[:, 1].var() - 3) < 0.35 and abs(smp[:, 2].var() - 1) < 0.08
