"""The reference's OWN pinned tests, run on the HIP engine (not on the checker) with the reference's literal workloads, seeds
and thresholds.  Each test names the Julia testset it restates (file:line under the reference tree); the calls read like the
Julia ones -- `ZigZagAD(1, U_Gauss_1D, grid_size=0)`, `sample_skeleton(sampler, 5000, 0.0, 1.0, seed=42)`,
`sample_from_skeleton(sampler, 5000, output)` -- with a target descriptor in place of the closure.  The random stream is the
engine's Philox stream, not MersenneTwister(seed), so these are the statistical / invariant pins of the reference, not replays:
what a Julia-produced fixture would add is tests/test_julia_fixtures.py.

`var` is Julia's (corrected, ddof = 1); `mean(samples)` / `var(samples)` of a matrix are over all entries, as in Julia."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def jvar(a, **kw):
    return np.var(a, ddof=1, **kw)


# ---------------------------------------------------------------- test/test_property_based.jl
def test_ergodicity(engine):
    """test/test_property_based.jl:87-100 ("Ergodicity")."""
    P = engine
    sampler = P.ZigZagAD(1, P.GaussianIso(), grid_size=0)
    output = P.sample_skeleton(sampler, 5000, 0.0, 1.0, seed=42)
    samples = P.sample_from_skeleton(sampler, 5000, output)
    assert abs(samples.mean()) < 0.2
    assert 0.8 < jvar(samples) < 1.2


def test_mixing_properties(engine):
    """test/test_property_based.jl:102-127 ("Mixing Properties"): |autocorr(1)| > |autocorr(10)|."""
    P = engine
    sampler = P.ZigZagAD(1, P.GaussianIso(), grid_size=0)
    output = P.sample_skeleton(sampler, 2000, 0.0, 1.0, seed=42)
    x = P.sample_from_skeleton(sampler, 2000, output).ravel()

    def autocorr(x, lag):
        n = len(x)
        m = x.mean()
        return ((x[:n - lag] - m) * (x[lag:] - m)).sum() / ((x - m) ** 2).sum()

    assert abs(autocorr(x, 1)) > abs(autocorr(x, 10))


def test_central_limit_theorem(engine):
    """test/test_property_based.jl:129-149 ("Central Limit Theorem"): 100 runs, seed = 42 + i."""
    P = engine
    sampler = P.ZigZagAD(1, P.GaussianIso(), grid_size=0)
    means = []
    for i in range(1, 101):
        output = P.sample_skeleton(sampler, 1000, 0.0, 1.0, seed=42 + i, verbose=False)
        means.append(P.sample_from_skeleton(sampler, 1000, output).mean())
    means = np.array(means)
    assert abs(means.mean()) < 0.1
    assert 0.0 < jvar(means) < 1.2


# ---------------------------------------------------------------- test/test_integration.jl
def test_bayesian_inference_workflow(engine):
    """test/test_integration.jl:11-39: U_Posterior(x) = x^2/2 + (x - 1)^2 / (2 * 0.5) = a Gaussian with precision 3 and mean 2/3."""
    P = engine
    sampler = P.ZigZagAD(1, P.GaussianDiag([2.0 / 3.0], [3.0]), grid_size=0)
    output = P.sample_skeleton(sampler, 2000, 0.0, 1.0, seed=42)
    samples = P.sample_from_skeleton(sampler, 2000, output)
    assert 0.5 < samples.mean() < 0.8
    assert 0.2 < jvar(samples) < 0.5


def test_bayesian_inference_workflow_from_source(engine):
    """The same testset with the potential given as the reference gives it -- a function, differentiated by forward mode
    (`create_gradient_function`, ADBackend.jl:95-109) -- through pdmp_create_from_source."""
    P = engine
    src = """
template <class X> __device__ auto U(const X &x, int d) {
    auto prior = -(sq(x[0])) / 2.0;                    // prior N(0, 1)
    auto lik = -(sq(x[0] - 1.0)) / (2.0 * 0.5);        // y_obs = 1.0, noise_var = 0.5
    return -(prior + lik);
}
"""
    sampler = P.ZigZagAD(1, P.SourcePotential(src), grid_size=0)
    output = P.sample_skeleton(sampler, 2000, 0.0, 1.0, seed=42)
    samples = P.sample_from_skeleton(sampler, 2000, output)
    assert 0.5 < samples.mean() < 0.8
    assert 0.2 < jvar(samples) < 0.5


def test_high_dimensional_sampling(engine):
    """test/test_integration.jl:41-65: ForwardECMCAD(20, U_HighDim, grid_size=0), xinit = zeros, vinit = ones (not normalised)."""
    P = engine
    dim = 20
    sampler = P.ForwardECMCAD(dim, P.GaussianIso(), grid_size=0)
    output = P.sample_skeleton(sampler, 1000, np.zeros(dim), np.ones(dim), seed=42)
    samples = P.sample_from_skeleton(sampler, 1000, output)
    assert samples.shape == (dim, 1000)
    assert np.isfinite(samples).all()
    assert (np.abs(samples.mean(axis=1)) < 0.5).all()
    v = jvar(samples, axis=1)
    assert ((0.5 < v) & (v < 2.0)).all()


# ---------------------------------------------------------------- test/test_comprehensive.jl
@pytest.mark.parametrize("name", ["ZigZag", "BPS", "ForwardECMC", "Boomerang"])
def test_all_sampler_types(engine, name):
    """test/test_comprehensive.jl:9-60 ("All Sampler Types"): dim = 3, N_sk = N = 500, xinit = randn(dim), vinit = ones(dim)
    (normalised for BPS and ForwardECMC), seed = 42; per coordinate |mean| < 0.5 and 0.2 < var < 2.0."""
    P = engine
    dim, N_sk, N = 3, 500, 500
    sampler = {"ZigZag": lambda: P.ZigZagAD(dim, P.GaussianIso(), grid_size=0),
               "BPS": lambda: P.BPSAD(dim, P.GaussianIso(), grid_size=0),
               "ForwardECMC": lambda: P.ForwardECMCAD(dim, P.GaussianIso(), grid_size=10),
               "Boomerang": lambda: P.BoomerangAD(dim, P.GaussianIso(), grid_size=0)}[name]()
    xinit = np.random.default_rng(42).normal(size=dim)
    vinit = np.ones(dim)
    if name in ("BPS", "ForwardECMC"):
        vinit = vinit / np.linalg.norm(vinit)
    output = P.sample_skeleton(sampler, N_sk, xinit, vinit, seed=42)
    assert len(output.t) > 0
    assert np.isfinite(output.t).all() and np.isfinite(output.X).all() and np.isfinite(output.V).all()
    samples = P.sample_from_skeleton(sampler, N, output)
    assert samples.shape == (dim, N)
    assert np.isfinite(samples).all()
    assert (np.abs(samples.mean(axis=1)) < 0.5).all()
    v = jvar(samples, axis=1)
    assert ((0.2 < v) & (v < 2.0)).all()


def test_reproducibility(engine):
    """test/test_comprehensive.jl:109-125 ("Reproducibility"): the same seed twice gives the same t, x, v (here: the same bits)."""
    P = engine
    sampler = P.ZigZagAD(1, P.GaussianIso(), grid_size=0)
    output1 = P.sample_skeleton(sampler, 100, 0.0, 1.0, seed=12345)
    output2 = P.sample_skeleton(sampler, 100, 0.0, 1.0, seed=12345)
    assert np.array_equal(output1.t, output2.t)
    assert np.array_equal(output1.X, output2.X)
    assert np.array_equal(output1.V, output2.V)
    other = P.sample_skeleton(sampler, 100, 0.0, 1.0, seed=12346)
    assert not np.array_equal(output1.t, other.t)


def test_long_run_stability(engine):
    """test/test_comprehensive.jl:133-148 ("Long Run Stability"): 10 000 events in 3-D, finite, strictly increasing times."""
    P = engine
    sampler = P.ZigZagAD(3, P.GaussianIso(), grid_size=0)
    output = P.sample_skeleton(sampler, 10000, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], seed=42)
    assert len(output.t) > 1000
    assert np.isfinite(output.t).all() and np.isfinite(output.X).all() and np.isfinite(output.V).all()
    assert (np.diff(output.t) > 0).all()                                   # :147


def test_memory_efficiency(engine):
    """test/test_comprehensive.jl:151-164: 5000 events -> 5000 samples, size (3, 5000), finite."""
    P = engine
    sampler = P.ZigZagAD(3, P.GaussianIso(), grid_size=0)
    output = P.sample_skeleton(sampler, 5000, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], seed=42)
    samples = P.sample_from_skeleton(sampler, 5000, output)
    assert samples.shape == (3, 5000)
    assert np.isfinite(samples).all()


def test_conservation_properties(engine):
    """test/test_comprehensive.jl:172-180 ("Conservation Properties"): |v| = sqrt(2) over the last 51 events."""
    P = engine
    sampler = P.ZigZagAD(2, P.GaussianIso(), grid_size=0)
    output = P.sample_skeleton(sampler, 1000, [0.0, 0.0], [1.0, 1.0], seed=42)
    v = output.v
    for i in range(len(v) - 51, len(v)):
        assert abs(np.linalg.norm(v[i]) - np.sqrt(2.0)) < 1e-5


def test_convergence_properties(engine):
    """test/test_comprehensive.jl:182-202 ("Convergence Properties"): 100 and 1000 samples of the same 1000-event skeleton."""
    P = engine
    sampler = P.ZigZagAD(1, P.GaussianIso(), grid_size=0)
    samples_100 = P.sample_from_skeleton(sampler, 100, P.sample_skeleton(sampler, 1000, 0.0, 1.0, seed=42))
    samples_1000 = P.sample_from_skeleton(sampler, 1000, P.sample_skeleton(sampler, 1000, 0.0, 1.0, seed=42))
    assert 0.5 < jvar(samples_100) < 2.0
    assert 0.5 < jvar(samples_1000) < 2.0


# ---------------------------------------------------------------- test/test_samplers.jl
def test_zigzag_sampler_testset(engine):
    """test/test_samplers.jl:9-62 ("ZigZag Sampler"): ZigZag(1, grad-U, grid_size=0) (FiniteDiff slopes: the constructor's default),
    1000 events, N = 1000 samples with |mean| < 0.5, 0.5 < var < 2.0; then the time-horizon method with T = 1.0,
    init_capacity = 8: t[end] == T exactly (`isapprox(...; atol=0, rtol=0)`), times non-decreasing, finite samples."""
    P = engine
    dim = 1
    sampler = P.ZigZag(dim, P.GaussianIso(), grid_size=0)
    assert sampler.dim == dim and sampler.grid_size == 0
    output = P.sample_skeleton(sampler, 1000, 0.0, 1.0, seed=42)
    assert len(output.t) > 0 and len(output.x) > 0 and len(output.v) > 0
    assert np.isfinite(output.t).all() and np.isfinite(output.X).all() and np.isfinite(output.V).all()
    N = 1000
    samples = P.sample_from_skeleton(sampler, N, output)
    assert samples.shape == (dim, N)
    assert np.isfinite(samples).all()
    assert abs(samples.mean()) < 0.5
    assert 0.5 < jvar(samples) < 2.0
    T = 1.0
    outputT = P.sample_skeleton(sampler, T, 0.0, 1.0, seed=42, verbose=False, init_capacity=8)
    assert outputT.t[-1] == T                                              # :59
    assert (np.diff(outputT.t) >= 0).all()                                 # :60
    samplesT = P.sample_from_skeleton(sampler, N, outputT)
    assert np.isfinite(samplesT).all()


def test_manual_gradient_forward_ecmc(engine):
    """test/test_samplers.jl:139-149: ForwardECMC(10, grad-U_strict; grid_size=10, tmax=1.0, mix_p=0.0, switch=true) with the
    user-defined gradient tanh.(x / 2), given as source (tanh by its exponential form); `length(hist.t) == 50`."""
    P = engine
    dim = 10
    src = """
template <class X> __device__ auto grad_i(int i, const X &x, int d) {
    auto e = exp_(x[i]);                                // tanh(x/2) = (e^x - 1) / (e^x + 1)
    return (e - 1.0) / (e + 1.0);
}
"""
    tgt = P.SourceGradient(src)
    sampler = P.ForwardECMC(dim, tgt, grid_size=10, tmax=1.0, mix_p=0.0, switch=True)
    xinit = np.random.default_rng(1).normal(size=dim)
    hist = P.sample_skeleton(sampler, 50, xinit, np.ones(dim) / np.sqrt(dim), seed=1, verbose=False)
    assert len(hist.t) == 50


def test_manual_gradient_zigzag_vectorized(engine):
    """test/test_samplers.jl:151-160: ZigZag(5, grad-U_strict = x; grid_size=10, tmax=1.0, vectorized_bound=true, signed_bound=true),
    FiniteDiff slopes (the constructor's default); 50 events."""
    P = engine
    dim = 5
    sampler = P.ZigZag(dim, P.GaussianIso(), grid_size=10, tmax=1.0, vectorized_bound=True, signed_bound=True)
    xinit = np.random.default_rng(2).normal(size=dim)
    hist = P.sample_skeleton(sampler, 50, xinit, np.ones(dim), seed=1, verbose=False)
    assert len(hist.t) == 50


# ---------------------------------------------------------------- test/test_diagnostics.jl
def test_three_point_history_known_answer(engine):
    """test/test_diagnostics.jl:100-124: the reference's one known-answer fixture of the path -- PDMPHistory(1, 3) with
    X = [0, .5, 1], V = 1, t = [0, .5, 1] -- through pdmp_sample_from_skeleton: RV_diagnostic(history, U; B = 4) interpolates the
    path at the B = 4 equispaced times and sums squared increments of U(x) = x^2 / 2; expected = sum_k (U(k/4) - U((k-1)/4))^2."""
    P = engine
    sampler = P.ZigZagAD(1, P.GaussianIso())
    hist = P.PDMPHistory(X=np.array([[0.0, 0.5, 1.0]]), V=np.ones((1, 3)), t=np.array([0.0, 0.5, 1.0]), is_active=np.ones((1, 3), bool),
                         horizon=np.zeros(3), ar=np.zeros(3), errored_bound=np.zeros(3, np.int32), error_value_ar=np.zeros((5, 3)),
                         rejected=np.zeros(3, np.int32), hitting_horizon=np.zeros(3, np.int32))
    B = 4
    s = P.sample_from_skeleton(sampler, B, hist)
    assert s.shape == (1, B)
    np.testing.assert_allclose(s[0], [0.25, 0.5, 0.75, 1.0], rtol=0, atol=1e-15)
    U = lambda x: x * x / 2
    path = np.concatenate([[hist.X[0, 0]], s[0]])
    rv = float(((U(path[1:]) - U(path[:-1])) ** 2).sum())
    expected = sum((U(k / 4) - U((k - 1) / 4)) ** 2 for k in range(1, 5))
    assert rv == pytest.approx(expected, rel=1e-12)
    full = P.sample_from_skeleton(sampler, B, hist, discard_vt=False)      # sample.jl:505-508: rows x, v, t
    assert full.shape == (3, B)
    np.testing.assert_allclose(full[2], [0.25, 0.5, 0.75, 1.0])
    np.testing.assert_allclose(full[1], 1.0)
