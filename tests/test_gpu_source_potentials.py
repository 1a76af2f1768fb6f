"""Potentials given as source (pdmp_create_from_source; the `U::Function` of the reference's *AD constructors,
ZigZagSamplers.jl:118-126, differentiated as in ADBackend.jl:95-109): the runtime-compiled kernel against the checker's
forward-mode restatement of the reference's own test potentials (test/test_config.jl:33-46), bit for bit."""
import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu

CASES = [
    ("ZigZagAD", 2, "src_banana", {}),
    ("ZigZagAD", 10, "src_banana", {}),
    ("ZigZagAD", 50, "src_banana", {}),                                   # G = 4: seeds and reads cross lanes
    ("ZigZagAD", 10, "src_banana", {"AD_backend": "FiniteDiff"}),
    ("ZigZagAD", 10, "src_banana", {"signed_bound": False}),
    ("ZigZagAD", 10, "src_banana", {"grid_size": 0}),                     # Brent constant bound
    ("ZigZagAD", 6, "src_funnel", {}),
    ("ZigZagAD", 12, "src_ridged", {}),
    ("BPSAD", 10, "src_banana", {}),
    ("BPSAD", 50, "src_banana", {}),
    ("BPSAD", 6, "src_funnel", {}),
    ("BPSAD", 12, "src_ridged", {"AD_backend": "FiniteDiff"}),
    ("BoomerangAD", 10, "src_banana", {}),
    ("BoomerangAD", 12, "src_ridged", {}),
    ("BoomerangAD", 40, "src_ridged", {"AD_backend": "ForwardDiff"}),
]


@pytest.mark.parametrize("kind,dim,target,kw", CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}-{'-'.join(f'{k}={v}' for k, v in c[3].items())}" for c in CASES])
def test_source_potential_matches_checker(engine, oracle, kind, dim, target, kw):
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C, n_sk = 9, 60
    x, v = make_init(kind, dim, C, seed=3, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=11, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 11, diagnostics=True)
    assert_bitwise(got, ref)


BANANA_GRADIENT_SOURCE = """
template <class X> __device__ auto grad_i(int i, const X &x, int d) {     // the engine's built-in banana gradient, as source
    if (i >= 2 || d < 2) return x[i];
    auto x0 = x[0];
    auto r = x[1] - (sq(x0) - 1.0);
    if (i == 1) return r;
    return x0 - (2.0 * x0) * r;
}
"""


@pytest.mark.parametrize("kind", ["ZigZag", "BPS", "Boomerang"])
def test_source_gradient_equals_builtin_target(engine, kind):
    """`ZigZag(dim, grad-U)` with the gradient given as source (ADBackend.jl:55-66) runs the same arithmetic as the
    built-in Banana target: identical skeletons.  (BPS with forward-mode slopes: the built-in target sums its affine
    coordinates once per state -- the "affine tail" order of DESIGN.md 2 -- which a gradient given as source cannot; the two
    agree to rounding, with identical decisions.)"""
    dim, C, n_sk = 10, 7, 80
    ctor = getattr(engine, kind)
    a = ctor(dim, engine.SourceGradient(BANANA_GRADIENT_SOURCE))
    b = ctor(dim, engine.Banana())
    x, v = make_init(kind, dim, C, seed=5, target="banana")
    ga = engine.sample_skeleton_ensemble(a, n_sk, x, v, seed=4, diagnostics=True)
    gb = engine.sample_skeleton_ensemble(b, n_sk, x, v, seed=4, diagnostics=True)
    if kind == "BPS":
        for k in ("rejected", "hitting_horizon", "errored_bound"):
            assert np.array_equal(ga[k], gb[k]), k
        for k in ("X", "V", "t"):       # BPS reflections amplify a rounding difference along a trajectory: most of it stays at rounding level
            close = np.isclose(ga[k], gb[k], rtol=1e-8, atol=1e-9)
            assert close.mean() > 0.9 and np.allclose(ga[k], gb[k], rtol=1e-3, atol=1e-4), (k, close.mean())
            assert np.allclose(ga[k][:, :20], gb[k][:, :20], rtol=1e-10, atol=1e-11), k
        return
    assert_bitwise(ga, gb)


def test_source_that_does_not_compile_is_an_argument_error(engine):
    s = engine.ZigZagAD(3, engine.SourcePotential("template <class X> __device__ auto U(const X &x, int d) { return undefined_name(x[0]); }"))
    with pytest.raises(engine.ArgumentError) as e:
        s.handle()
    assert "undefined_name" in str(e.value)


def test_source_potential_statistics(engine):
    """sample() end to end on a source potential: U = sum(x.^2)/2 gives N(0, I) (README.md:36-49)."""
    src = """template <class X> __device__ auto U(const X &x, int d) {
        auto s = sq(x[0]); for (int j = 1; j < d; ++j) s = s + sq(x[j]); return s / 2.0; }"""
    d, C = 4, 64
    s = engine.ZigZagAD(d, engine.SourcePotential(src))
    rs = np.random.default_rng(0)
    x, v = rs.normal(size=(C, d)), rs.choice([-1.0, 1.0], size=(C, d))
    sk = engine.sample_skeleton_ensemble(s, 2000, x, v, seed=1, diagnostics=False)
    smp = engine.sample_from_skeleton_ensemble(s, 2000, sk["X"], sk["V"], sk["t"]).reshape(-1, d)
    assert np.abs(smp.mean(0)).max() < 0.05 and np.abs(smp.var(0) - 1).max() < 0.1
    # and the same skeleton as the built-in isotropic Gaussian up to the rounding of the differentiated sum
    b = engine.ZigZagAD(d, engine.GaussianIso())
    ga = engine.sample_skeleton_ensemble(s, 50, x, v, seed=2)
    gb = engine.sample_skeleton_ensemble(b, 50, x, v, seed=2)
    np.testing.assert_allclose(ga["t"], gb["t"], rtol=1e-9)


@pytest.mark.parametrize("kind,kw", [("ForwardECMC", {}), ("RHMC", {"step_size": 0.1}), ("ForwardECMC", {"ran_p": True, "mix_p": 1.0})])
def test_source_potential_with_the_other_samplers(engine, oracle, kind, kw):
    """A potential given as source under ForwardECMC and RHMC (their jumps / flows read grad-U at the state itself)."""
    s, cfg = make_pair(engine, oracle, kind, 8, "src_banana", **kw)
    C, n_sk = 7, 30
    x, v = make_init(kind, 8, C, seed=3, target="src_banana")
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=19, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 19, diagnostics=True)
    assert_bitwise(got, ref)
