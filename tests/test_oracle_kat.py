"""Pins for the CPU oracle (the checker): known-answer vectors and hand-derived values.  CPU only."""
import math

import numpy as np
import pytest


# Random123 v1.14 kat_vectors: philox4x32 10 rounds
PHILOX_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,out", PHILOX_KAT)
def test_philox_known_answers(oracle, ctr, key, out):
    assert oracle.philox(ctr, key) == out


# Random123 v1.14 kat_vectors: philox4x32 7 rounds -- the generator of arithmetic contract v2 (the default since round 5)
PHILOX7_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]),
]


@pytest.mark.parametrize("ctr,key,out", PHILOX7_KAT)
def test_philox7_known_answers(oracle, ctr, key, out):
    assert oracle.philox(ctr, key, rounds=7) == out


@pytest.mark.parametrize("ver", [1, 2])
def test_stream_layout_from_the_block_function(oracle, ver):
    """DESIGN.md "Random stream": variate n of chain c under key `seed` = 64-bit half n & 1 of Philox block n >> 1 with counter
    (block, chain); u = (top 52 bits + 0.5) 2^-52; Exp(1) = -log u.  v1: ten rounds, v2: seven."""
    seed, chain = 0x1234567811223344, 77
    u = oracle.rng_dump(seed, chain, 0, 8, rng_version=ver)
    e = oracle.rng_dump(seed, chain, 1, 8, rng_version=ver)
    for n in range(8):
        w = oracle.philox([n >> 1, 0, chain, 0], [seed & 0xffffffff, seed >> 32], rounds=10 if ver == 1 else 7)
        bits = w[2 * (n & 1)] | (w[2 * (n & 1) + 1] << 32)
        assert u[n] == ((bits >> 12) + 0.5) * 2.0 ** -52
        assert e[n] == -oracle.log(u[n], rng_version=ver)


def test_box_muller_v2_from_its_definition(oracle):
    """Contract v2's normal pair from one block (oracle/pdmp_oracle.c: box_muller): radius sqrt(-2 log u0) from the first half; angle =
    (m - 3/2) pi/2 with m in [1, 2) from the top 52 bits of the second half, bit 0 swaps sin and cos, bit 1 flips both signs."""
    import mpmath
    mpmath.mp.prec = 100
    seed, chain, d = 99, 5, 64
    z = oracle.rng_dump(seed, chain, 2, d, rng_version=2)
    for p in range(d // 2):
        w = oracle.philox([p, 0, chain, 0], [seed, 0], rounds=7)
        b0, b1 = w[0] | (w[1] << 32), w[2] | (w[3] << 32)
        u0 = mpmath.mpf((b0 >> 12) + 0.5) * mpmath.mpf(2) ** -52
        rad = mpmath.sqrt(-2 * mpmath.log(u0))
        a = (mpmath.mpf(1) + mpmath.mpf(b1 >> 12) * mpmath.mpf(2) ** -52 - mpmath.mpf(3) / 2) * mpmath.pi / 2
        sa, ca = mpmath.sin(a), mpmath.cos(a)
        if b1 & 1:
            sa, ca = ca, sa
        if b1 & 2:
            rad = -rad
        assert abs(z[2 * p] - float(rad * ca)) < 1e-14 and abs(z[2 * p + 1] - float(rad * sa)) < 1e-14


@pytest.mark.parametrize("ver", [1, 2])
def test_elementary_functions_against_mpmath(oracle, ver):
    """log / exp / sin / cos of both contracts within 2 ulp of the true values (fdlibm's series; v2 evaluates them by Horner on fma)."""
    import mpmath
    mpmath.mp.prec = 120
    rs = np.random.default_rng(ver)

    def ulps(got, ref):
        rf = float(ref)
        return float(abs(mpmath.mpf(got) - ref) / (math.ulp(rf) if rf != 0 else 5e-324))
    for x in list(rs.random(2000)) + list(np.exp(rs.uniform(-40, 40, 2000))) + [1.0, 0.5, 2.0, 0.7071067811865476, 0.7071067811865475, 2.0 ** -53]:
        assert ulps(oracle.log(x, ver), mpmath.log(mpmath.mpf(x))) < 1.0
    for x in list(rs.uniform(-700, 700, 2000)) + list(rs.uniform(-1, 1, 2000)) + [0.0]:
        assert ulps(oracle.exp(x, ver), mpmath.exp(mpmath.mpf(x))) < 1.0
    for x in list(rs.uniform(-50, 50, 2000)) + [0.0, math.pi / 4, math.pi / 2, math.pi]:
        s, c = oracle.sincos(x, ver)
        assert abs(s - float(mpmath.sin(mpmath.mpf(x)))) < 1e-15 and abs(c - float(mpmath.cos(mpmath.mpf(x)))) < 1e-15


def test_log_within_one_ulp_of_libm(oracle):
    rs = np.random.default_rng(0)
    xs = np.concatenate([rs.random(5000), 2.0 ** rs.uniform(-60, 0, 5000), 1 - 2.0 ** rs.uniform(-52, -1, 1000),
                         [0.5, 0.25, 1 - 2.0 ** -53, 2.0 ** -53]])
    for x in xs:
        if 0 < x < 1:
            ref = math.log(x)
            assert abs(oracle.log(x) - ref) <= 2 * np.spacing(abs(ref)) + 1e-300


def test_sincos_accuracy(oracle):
    rs = np.random.default_rng(1)
    for x in np.concatenate([rs.uniform(0, 60, 4000), [0.0, 1e-9, math.pi / 4, math.pi / 2, math.pi, 10 * math.pi]]):
        s, c = oracle.sincos(x)
        assert abs(s - math.sin(x)) < 5e-15 and abs(c - math.cos(x)) < 5e-15
    for u in rs.random(4000):
        s, c = oracle.sincos_2pi(u)
        assert abs(s - math.sin(2 * math.pi * u)) < 2e-15 and abs(c - math.cos(2 * math.pi * u)) < 2e-15


def test_stream_moments(oracle):
    u = oracle.rng_dump(1, 0, 0, 200000)
    e = oracle.rng_dump(1, 0, 1, 200000)
    z = oracle.rng_dump(1, 0, 2, 200001)
    assert 0 < u.min() and u.max() < 1
    assert abs(u.mean() - 0.5) < 3e-3 and abs(u.var() - 1 / 12) < 1e-3
    assert abs(e.mean() - 1) < 1e-2 and abs(e.var() - 1) < 3e-2 and e.min() > 0
    assert abs(z.mean()) < 1e-2 and abs(z.var() - 1) < 1e-2 and abs((z ** 4).mean() - 3) < 0.1
    # streams are keyed by (seed, chain): different keys differ, same key repeats
    assert np.array_equal(u, oracle.rng_dump(1, 0, 0, 200000))
    assert not np.array_equal(u[:100], oracle.rng_dump(1, 1, 0, 100))
    assert not np.array_equal(u[:100], oracle.rng_dump(2, 0, 0, 100))
    # exponential n is -log of uniform n
    assert np.array_equal(e[:50], np.array([-oracle.log(x) for x in u[:50]]))


def test_vectorised_bound_worked_example(oracle):
    """SURVEY.md 3.3a, hand-evaluated from UpperBound.jl:203-247,264-273: Gaussian, x=[.5,-1], v=[1,1], horizon 2, n=10."""
    cfg = oracle.Config(sampler=oracle.ZIGZAG, dim=2, grid_size=10, tmax=2.0)
    grid, box, cum, step = oracle.upper_bound(cfg, [0.5, -1.0], [1.0, 1.0], 2.0)
    assert step == pytest.approx(2 / 9)
    assert grid[0] == 0.0 and grid[-1] == 2.0
    np.testing.assert_allclose(box, [0.722222, 0.944444, 1.166667, 1.388889, 1.722222, 2.166667, 2.611111, 3.055556, 3.5], atol=1e-6)
    np.testing.assert_allclose(cum, [0, 0.160494, 0.370370, 0.629630, 0.938272, 1.320988, 1.802469, 2.382716, 3.061728, 3.839506], atol=1e-6)
    tp, lb = oracle.next_event(grid, box, cum, step, 0.1)
    assert tp == pytest.approx(0.138462, abs=1e-6) and lb == pytest.approx(0.722222, abs=1e-6)
    tp, lb = oracle.next_event(grid, box, cum, step, 1.0)
    assert tp == pytest.approx(0.924731, abs=1e-6) and lb == pytest.approx(1.722222, abs=1e-6)
    tp, lb = oracle.next_event(grid, box, cum, step, 5.0)
    assert math.isinf(tp) and lb == pytest.approx(3.5)


def test_scalar_bound_bps_refresh_counted_twice(oracle):
    """Q2: the signed BPS rate already contains refresh_rate and upper_bound_grid adds it again
    (BouncyParticleSamplers.jl:44-47 + UpperBound.jl:131).  Gaussian: s(t) = x.v + |v|^2 t + rho is linear."""
    rho = 0.1
    cfg = oracle.Config(sampler=oracle.BPS, dim=2, grid_size=5, tmax=1.0, refresh_rate=rho, vectorized_bound=False)
    x, v = np.array([0.3, -0.2]), np.array([0.6, 0.8])
    grid, box, cum, step = oracle.upper_bound(cfg, x, v, 1.0)
    s = lambda t: x @ v + (v @ v) * t + rho
    expect = np.array([max(s(grid[j]), s(grid[j + 1]), 0.0) + rho for j in range(4)])
    np.testing.assert_allclose(box, expect, rtol=1e-14)
    np.testing.assert_allclose(cum[1:], np.cumsum(expect) * step, rtol=1e-14)


@pytest.mark.parametrize("target", ["iso", "diag", "banana", "dense"])
def test_gradient_matches_potential(oracle, target):
    """AD consistency in the spirit of test/test_ad_backends.jl:172-215: grad-U against central differences of U."""
    d = 6
    rs = np.random.default_rng(3)
    params, tid = None, oracle.T_GAUSS_ISO
    if target == "diag":
        params, tid = np.concatenate([rs.normal(size=d), np.exp(rs.normal(size=d))]), oracle.T_GAUSS_DIAG
    elif target == "banana":
        tid = oracle.T_BANANA
    elif target == "dense":
        A = rs.normal(size=(d, d)); P = A @ A.T + d * np.eye(d)
        params, tid = P.T.ravel(), oracle.T_GAUSS_DENSE
    cfg = oracle.Config(dim=d, target_id=tid, target_params=params)
    x = rs.normal(size=d)
    g = oracle.grad(cfg, x)
    h = 1e-6
    fd = np.array([(oracle.potential(cfg, x + h * e) - oracle.potential(cfg, x - h * e)) / (2 * h) for e in np.eye(d)])
    np.testing.assert_allclose(g, fd, rtol=1e-6, atol=1e-7)


def test_forward_and_finite_difference_slopes_agree(oracle):
    """Q13: FiniteDiff slopes carry ~1e-8 error that only enters through the tangent intersection."""
    d = 8
    rs = np.random.default_rng(5)
    x, v = 1 + 0.1 * rs.normal(size=d), rs.choice([-1.0, 1.0], size=d)
    a = oracle.Config(dim=d, target_id=oracle.T_BANANA, deriv_mode=oracle.DERIV_FORWARD)
    b = oracle.Config(dim=d, target_id=oracle.T_BANANA, deriv_mode=oracle.DERIV_FINITE_DIFF)
    ga, ba, ca, _ = oracle.upper_bound(a, x, v, 0.7)
    gb, bb, cb, _ = oracle.upper_bound(b, x, v, 0.7)
    np.testing.assert_allclose(ba, bb, rtol=1e-5)
    np.testing.assert_allclose(ca, cb, rtol=1e-5)


def test_sample_from_skeleton_known_answer(oracle):
    """The reference's one known-answer fixture for linear interpolation along a skeleton
    (test/test_diagnostics.jl:100-124): X=[0,.5,1], V=1, t=[0,.5,1]."""
    cfg = oracle.Config(dim=1)
    X = np.array([[[0.0], [0.5], [1.0]]]); V = np.ones((1, 3, 1)); t = np.array([[0.0, 0.5, 1.0]])
    out = oracle.sample_from_skeleton(cfg, 4, X, V, t)
    np.testing.assert_allclose(out[0, :, 0], [0.25, 0.5, 0.75, 1.0], rtol=0, atol=1e-15)
    full = oracle.sample_from_skeleton(cfg, 4, X, V, t, discard_vt=False)
    assert full.shape == (1, 4, 3)
    np.testing.assert_allclose(full[0, :, 2], [0.25, 0.5, 0.75, 1.0])


def test_summation_orders_agree_to_rounding(oracle):
    """REFERENCE, WAVE64 and GROUP(G) orders differ only in the last bits."""
    d = 50
    rs = np.random.default_rng(0)
    x, v = rs.normal(size=(3, d)), rs.choice([-1.0, 1.0], size=(3, d))
    ref = oracle.sample_skeleton(oracle.Config(dim=d), 200, x, v, 5)
    for cfg in (oracle.Config(dim=d, arith=oracle.ARITH_WAVE64), oracle.Config(dim=d, arith=oracle.ARITH_GROUP, group_lanes=1), oracle.Config(dim=d, arith=oracle.ARITH_GROUP, group_lanes=4),
                oracle.Config(dim=d, arith=oracle.ARITH_GROUP, group_lanes=64)):
        o = oracle.sample_skeleton(cfg, 200, x, v, 5)
        np.testing.assert_allclose(o["t"], ref["t"], rtol=1e-11)
        np.testing.assert_allclose(o["X"], ref["X"], rtol=1e-9, atol=1e-11)
        assert np.array_equal(o["V"], ref["V"])
