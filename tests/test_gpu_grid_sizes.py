"""`grid_size` as a real parameter (round 5): n_grid is a free argument of both bounds (src/UpperBound.jl:92,203) and the reference's own
tests pass 5, 8 and 20 -- `ZigZagAD(dim, U, grid_size in {0, 5, 10, 20})` (test/test_coverage.jl:153-160), `SpeedUpZigZagAD(...; grid_size=5)`,
`StickyZigZagAD(...; grid_size=5)` (:228-270), `ForwardECMCAD(dim, U_quad, grid_size=8, tmax=1.0, adaptive=false)` (test/test_diagnostics.jl:128).
Rounds 1-4 served 10 (and 0, Brent) on the group kernel and sent the rest to the wave-per-chain kernel (three samplers, built-in targets,
dim <= 256) or refused it; now every sampler, the potentials given as source and the dense-precision target run any 2 .. 64 on the
LDS-form group kernel, bit for bit against the checker in the same GROUP(G) order."""
import numpy as np
import pytest

from helpers import DIAG_KEYS, assert_bitwise, make_init, make_pair
from test_gpu_sticky import pair as sticky_pair

pytestmark = pytest.mark.gpu

FORM_GROUP, FORM_WAVE = 2, 1

U_QUAD_SOURCE = """
template <class X> __device__ auto U(const X &x, int d) {      // U_test / U_quad of the reference's tests: sum(x.^2) / 2
    auto s = sq(x[0]);
    for (int i = 1; i < d; ++i) s = s + sq(x[i]);
    return s / 2.0;
}
"""

GRIDS = [2, 3, 5, 8, 20, 33, 64]
CASES = [(kind, dim, target, kw, g)
         for kind, dim, target, kw in [
             ("ZigZagAD", 50, "iso", {}), ("ZigZagAD", 50, "banana", {}), ("ZigZagAD", 13, "diag", {"signed_bound": False}),
             ("ZigZag", 20, "banana", {}),                                        # finite-difference slopes
             ("ZigZagAD", 9, "banana", {"vectorized_bound": False}),
             ("ZigZagAD", 300, "iso", {}),                                        # beyond the wave kernel's 256 coordinates
             ("BPS", 50, "banana", {}), ("BPS", 20, "diag", {"AD_backend": "FiniteDiff"}), ("BPSAD", 9, "iso", {"signed_bound": False}),
             ("Boomerang", 100, "iso", {}), ("BoomerangAD", 24, "diag", {"refresh_rate": 0.1}), ("Boomerang", 10, "banana", {}),
             ("ForwardECMC", 50, "banana", {}), ("ForwardECMCAD", 9, "diag", {"ran_p": True}),
             ("SpeedUpZigZagAD", 12, "iso", {}), ("SpeedUpZigZag", 6, "banana", {}),
             ("ZigZagAD", 10, "src_banana", {}), ("BPSAD", 6, "src_funnel", {}),
             ("ZigZagAD", 40, "dense", {}), ("BPS", 33, "dense", {"refresh_rate": 0.5}), ("BoomerangAD", 20, "dense", {}),
         ]
         for g in GRIDS]


@pytest.mark.parametrize("kind,dim,target,kw,grid", CASES,
                         ids=[f"{c[0]}-d{c[1]}-{c[2]}-n{c[4]}-{'-'.join(f'{k}={v}' for k, v in c[3].items())}" for c in CASES])
def test_any_grid_size_matches_checker(engine, oracle, kind, dim, target, kw, grid):
    s, cfg = make_pair(engine, oracle, kind, dim, target, grid_size=grid, **kw)
    assert s.group_lanes() > 0 and cfg.grid_size == grid
    C, n_sk = 11, 50
    x, v = make_init(kind, dim, C, seed=grid, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=100 + grid, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 100 + grid, diagnostics=True)
    assert_bitwise(got, ref)
    assert s.last_launch_form() in (FORM_GROUP, 6)                       # the LDS-form group kernel (6: a potential given as source)


@pytest.mark.parametrize("grid", GRIDS)
@pytest.mark.parametrize("dim,target,kap,kw", [(2, "iso", 0.5, {}), (13, "diag", 0.3, {}), (50, "iso", 0.7, {}), (20, "banana", 0.5, {"vectorized_bound": False}),
                                               (6, "iso", 0.4, {"AD_backend": "FiniteDiff"})])
def test_sticky_any_grid_size(engine, oracle, dim, target, kap, kw, grid):
    s, cfg = sticky_pair(engine, oracle, "StickyZigZagAD", dim, target, kap, grid_size=grid, **kw)
    x, v = make_init("ZigZag", dim, 13, seed=9, target=target)
    got = engine.sample_skeleton_ensemble(s, 70, x, v, seed=51, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, 70, x, v, 51, diagnostics=True)
    assert_bitwise(got, ref, keys=("X", "V", "t", "is_active") + DIAG_KEYS)


@pytest.mark.parametrize("grid", [0, 5, 10, 20])
def test_reference_call_zigzagad_source_potential_grid_sizes(engine, oracle, grid):
    """test/test_coverage.jl:139-161 verbatim: `ZigZagAD(2, U_test, grid_size=grid_size)`, `sample_skeleton(sampler, 100, [0, 0], [1, 1], seed=42)`,
    U given as a function -- here as source, differentiated by forward mode on the device; the same assertions, and the checker's
    forward-mode restatement of the same potential bit for bit (ORC_T_GAUSS_ISO has the same gradient: x)."""
    P = engine
    s = P.ZigZagAD(2, P.SourcePotential(U_QUAD_SOURCE), grid_size=grid)
    out = P.sample_skeleton(s, 100, np.zeros(2), np.ones(2), seed=42)
    assert len(out.t) == 100 and np.isfinite(out.t).all() and np.isfinite(out.X).all()
    assert np.all(np.diff(out.t) > 0)
    # the built-in isotropic Gaussian has the same gradient (d/dx_i of sum(x^2) / 2 by forward mode is (2 x_i) / 2 = x_i exactly): the checker
    # on that target, in the layout the source-given potential runs in and WITHOUT the engine order's algebraic identities (a potential given
    # as source runs the reference's full tangent-intersection formula: the engine cannot know that its gradient is affine), bit for bit
    cfg = oracle.Config(sampler=oracle.ZIGZAG, dim=2, grid_size=grid, tmax=2.0, arith=oracle.ARITH_GROUP, group_lanes=s.group_lanes(), no_identities=True)
    ref = oracle.sample_skeleton(cfg, 100, np.zeros((1, 2)), np.ones((1, 2)), 42)
    assert np.array_equal(out.t, ref["t"][0]) and np.array_equal(out.X, ref["X"][0].T)


def test_reference_call_speedup_zigzag_grid5(engine, oracle):
    """test/test_coverage.jl:228-252 verbatim: SpeedUpZigZagAD(2, U_quad; grid_size=5, tmax=0, vectorized_bound=false, signed_bound=true,
    AD_backend="Zygote") warns, ends up adaptive with an unsigned bound; 80 events from [0, 0] / [1, 1], 50 samples, all finite."""
    P = engine
    with pytest.warns(UserWarning):
        s = P.SpeedUpZigZagAD(2, P.GaussianIso(), grid_size=5, tmax=0, vectorized_bound=False, signed_bound=True, AD_backend="Zygote")
    assert s.adaptive is True and s.signed_bound is False
    h = P.sample_skeleton(s, 80, np.zeros(2), np.ones(2), seed=42)
    assert len(h.t) == 80 and np.isfinite(h.t).all()
    smp = P.sample_from_skeleton(s, 50, h)
    assert smp.shape == (2, 50) and np.isfinite(smp).all()
    cfg = oracle.Config(sampler=oracle.SUZZ, dim=2, grid_size=5, tmax=s.tmax, refresh_rate=0.0, vectorized_bound=False, signed_bound=False,
                        adaptive=True, deriv_mode=oracle.DERIV_FORWARD, arith=oracle.ARITH_GROUP, group_lanes=s.group_lanes())
    ref = oracle.sample_skeleton(cfg, 80, np.zeros((1, 2)), np.ones((1, 2)), 42)
    assert np.array_equal(h.t, ref["t"][0]) and np.array_equal(h.X, ref["X"][0].T)


def test_reference_call_sticky_zigzag_grid5(engine, oracle):
    """test/test_coverage.jl:254-285 verbatim: StickyZigZagAD(2, U_quad, fill(100.0, 2); grid_size=5, tmax=0, vectorized_bound=false,
    signed_bound=true, AD_backend="Zygote"), xinit = [1e-6, 0.2], vinit = [-1, -1], 120 events: the first coordinate crosses its axis at once."""
    P = engine
    with pytest.warns(UserWarning):
        s = P.StickyZigZagAD(2, P.GaussianIso(), np.full(2, 100.0), grid_size=5, tmax=0, vectorized_bound=False, signed_bound=True, AD_backend="Zygote")
    assert s.adaptive is True and s.signed_bound is False
    h = P.sample_skeleton(s, 120, np.array([1e-6, 0.2]), np.array([-1.0, -1.0]), seed=42)
    assert len(h.t) == 120 and np.isfinite(h.t).all() and np.isfinite(h.X).all()
    assert (~h.is_active).any()                                           # it did stick
    cfg = oracle.Config(sampler=oracle.STICKY, dim=2, grid_size=5, tmax=s.tmax, refresh_rate=0.0, vectorized_bound=False, signed_bound=False,
                        adaptive=True, deriv_mode=oracle.DERIV_FORWARD, target_params=np.full(2, 100.0), arith=oracle.ARITH_GROUP,
                        group_lanes=s.group_lanes())
    ref = oracle.sample_skeleton(cfg, 120, np.array([[1e-6, 0.2]]), np.array([[-1.0, -1.0]]), 42)
    assert np.array_equal(h.t, ref["t"][0]) and np.array_equal(h.X, ref["X"][0].T)


def test_reference_call_forward_ecmc_grid8_time_horizon(engine, oracle):
    """test/test_diagnostics.jl:126-141: ForwardECMCAD(3, U_quad, grid_size=8, tmax=1.0, adaptive=false), xinit = zeros, vinit = ones / sqrt(3),
    run to T = 3.0 (sample_skeleton_with_diagnostic drives sample_skeleton(sampler, T, ...)): the time-horizon variant on the engine
    against the checker's, bit for bit."""
    P = engine
    s = P.ForwardECMCAD(3, P.GaussianIso(), grid_size=8, tmax=1.0, adaptive=False)
    assert s.adaptive is False and s.grid_size == 8
    x0, v0 = np.zeros((1, 3)), np.ones((1, 3)) / np.sqrt(3.0)
    cfg = oracle.Config(sampler=oracle.FECMC, dim=3, grid_size=8, tmax=1.0, refresh_rate=0.0, vectorized_bound=False, signed_bound=True,
                        adaptive=False, deriv_mode=oracle.DERIV_FORWARD, arith=oracle.ARITH_GROUP, group_lanes=s.group_lanes())
    got = P.sample_skeleton_ensemble(s, 60, x0, v0, seed=42, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, 60, x0, v0, 42, diagnostics=True)
    assert_bitwise(got, ref)
    h = P.sample_skeleton(s, 3.0, x0[0], v0[0], seed=42)
    r = oracle.sample_skeleton_until(cfg, 3.0, x0, v0, 42)
    n = int(r["count"][0])
    assert len(h.t) == n and h.t[-1] == 3.0
    assert np.array_equal(h.t, r["t"][0, :n]) and np.array_equal(h.X, r["X"][0, :n].T)


@pytest.mark.parametrize("grid", [5, 20])
def test_zigzag_d50_other_grids_leave_the_wave_kernel(engine, oracle, monkeypatch, grid):
    """Zig-Zag d = 50 at grid_size 5 / 20: the LDS-form group kernel (G = 4), no longer skeleton_kernel -- and, asked for by
    PDMP_GROUP_LANES=0, still the wave-per-chain kernel in its own WAVE64 order."""
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso", grid_size=grid)
    x, v = make_init("ZigZagAD", 50, 70)
    got = engine.sample_skeleton_ensemble(s, 40, x, v, seed=3)
    assert s.last_launch_form() == FORM_GROUP and s.group_lanes() == 4
    assert_bitwise(got, oracle.sample_skeleton(cfg, 40, x, v, 3), keys=("X", "V", "t"))
    monkeypatch.setenv("PDMP_GROUP_LANES", "0")
    s2, cfg2 = make_pair(engine, oracle, "ZigZagAD", 50, "iso", grid_size=grid)
    got2 = engine.sample_skeleton_ensemble(s2, 40, x, v, seed=3)
    assert s2.last_launch_form() == FORM_WAVE and s2.group_lanes() == 0
    assert_bitwise(got2, oracle.sample_skeleton(cfg2, 40, x, v, 3), keys=("X", "V", "t"))


def test_time_horizon_with_another_grid(engine, oracle):
    """The in-kernel time horizon (chunked resume) on a grid other than 10."""
    from test_gpu_time_horizon import _compare_until
    s, cfg = make_pair(engine, oracle, "BPS", 20, "banana", grid_size=7)
    x, v = make_init("BPS", 20, 21, seed=2, target="banana")
    _compare_until(engine, oracle, s, cfg, x, v, 2.5, 9, 8, ("X", "V", "t") + DIAG_KEYS)
