"""The lane-PAIR-per-chain kernel (csrc/lane_kernel.hpp, GL = 2) -- large Zig-Zag ensembles at d = 53..104 on the isotropic Gaussian
and the banana -- against the checker in the GROUP(8) order and against the G = 8 group kernel it replaces: bit for bit, every array,
every diagnostic; hand-off between segments, ragged batches, bound instalments, unaligned outputs, the in-kernel time horizon, the
fall-back for speeds that are not +-1."""
import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu


@pytest.fixture
def lane(monkeypatch):
    monkeypatch.setenv("PDMP_LANE", "1")      # take the lane kernel whatever the number of chains
    return monkeypatch


def _run(engine, oracle, dim, target, C, n_sk, seed=3, init_seed=0, diagnostics=True, **kw):
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, target, **kw)
    assert s.group_lanes() == 8
    x, v = make_init("ZigZagAD", dim, C, seed=init_seed, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=seed, diagnostics=diagnostics)
    import os
    if os.environ.get("PDMP_LANE") == "1":
        assert s.last_launch_form() == 7, "this case is meant to run in the lane kernel"
    return s, cfg, x, v, got


@pytest.mark.parametrize("dim", [53, 55, 56, 57, 60, 61, 64, 65, 68, 72, 73, 77, 80, 81, 88, 89, 92, 96, 97, 99, 100, 101, 103, 104])
@pytest.mark.parametrize("target", ["iso", "banana"])
def test_pair_kernel_matches_checker(engine, oracle, lane, dim, target):
    s, cfg, x, v, got = _run(engine, oracle, dim, target, 75, 90)            # 75 chains: two full batches of 32 and a ragged one
    ref = oracle.sample_skeleton(cfg, 90, x, v, 3, diagnostics=True)
    assert_bitwise(got, ref)


@pytest.mark.parametrize("target,dim", [("iso", 100), ("banana", 100), ("banana", 64), ("iso", 57)])
def test_pair_kernel_equals_group_kernel(engine, oracle, monkeypatch, target, dim):
    """The choice between the two kernels -- made per launch from the number of chains -- cannot be seen in the results."""
    monkeypatch.setenv("PDMP_LANE", "0")
    _, _, _, _, a = _run(engine, oracle, dim, target, 167, 150, seed=11)
    monkeypatch.setenv("PDMP_LANE", "1")
    _, _, _, _, b = _run(engine, oracle, dim, target, 167, 150, seed=11)
    assert_bitwise(b, a)


@pytest.mark.parametrize("segments", [2, 7, 1000])
@pytest.mark.parametrize("target,dim", [("iso", 100), ("banana", 100), ("banana", 61), ("iso", 86)])
def test_pair_kernel_segment_handoff(engine, oracle, lane, segments, target, dim):
    lane.setenv("PDMP_FORCE_SEGMENTS", str(segments))
    s, cfg, x, v, got = _run(engine, oracle, dim, target, 40, 96)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 96, x, v, 3, diagnostics=True))


@pytest.mark.parametrize("dim", [53, 66, 79, 100, 104])
@pytest.mark.parametrize("target", ["iso", "banana"])
def test_pair_kernel_without_diagnostics(engine, oracle, lane, dim, target):
    lane.setenv("PDMP_FORCE_SEGMENTS", "4")
    s, cfg, x, v, got = _run(engine, oracle, dim, target, 67, 72, seed=17, diagnostics=False)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 72, x, v, 17, diagnostics=False), keys=("X", "V", "t"))


@pytest.mark.parametrize("budget", ["1", "2", "5", "9"])
def test_pair_kernel_bound_instalments(engine, oracle, lane, budget):
    lane.setenv("PDMP_CELL_BUDGET", budget)
    s, cfg, x, v, got = _run(engine, oracle, 100, "banana", 40, 60)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 60, x, v, 3, diagnostics=True))


@pytest.mark.parametrize("kw", [{"tmax": 0.0}, {"tmax": 0.05}, {"tmax": 20.0}, {"adaptive": False, "tmax": 0.4}])
def test_pair_kernel_horizons(engine, oracle, lane, kw):
    s, cfg, x, v, got = _run(engine, oracle, 100, "banana", 40, 60, **kw)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 60, x, v, 3, diagnostics=True))


def test_pair_kernel_zero_start(engine, oracle, lane):
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 100, "iso")
    x = np.zeros((33, 100))
    v = np.where(np.random.default_rng(5).random((33, 100)) < 0.5, -1.0, 1.0)
    got = engine.sample_skeleton_ensemble(s, 40, x, v, seed=9, diagnostics=True)
    assert s.last_launch_form() == 7
    assert_bitwise(got, oracle.sample_skeleton(cfg, 40, x, v, 9, diagnostics=True))


def test_pair_kernel_leaves_other_speeds_to_the_group_kernel(engine, oracle, lane):
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 100, "iso")
    x, v = make_init("ZigZagAD", 100, 70)
    v[37, 13] = 0.5
    got = engine.sample_skeleton_ensemble(s, 50, x, v, seed=2, diagnostics=True)
    assert s.last_launch_form() == 3
    assert_bitwise(got, oracle.sample_skeleton(cfg, 50, x, v, 2, diagnostics=True))


def test_pair_kernel_full_machine(engine, oracle, monkeypatch):
    """Default selection (no environment): 33 000 chains at d = 100 take the pair kernel, more batches than resident waves, several
    segments per chain; a strided sample of chains bit for bit against the checker, global chain ids, invariants on every chain."""
    monkeypatch.delenv("PDMP_LANE", raising=False)
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 100, "banana")
    C = 33000
    x, v = make_init("ZigZagAD", 100, C, target="banana")
    got = engine.sample_skeleton_ensemble(s, 64, x, v, seed=77, diagnostics=False)
    assert s.last_launch_form() == 7
    for c in np.arange(0, C, 1291):
        r1 = oracle.sample_skeleton(cfg, 64, x[c:c + 1], v[c:c + 1], 77, chain_id_offset=int(c), diagnostics=False)
        for k in ("X", "V", "t"):
            assert np.array_equal(got[k][c], r1[k][0]), (k, c)
    assert (np.diff(got["t"], axis=1) > 0).all()
    assert (np.abs(got["V"]) == 1.0).all()


@pytest.mark.parametrize("offs", [(0, 5, 2), (8, 3, 7), (0, 0, 1), (3, 0, 0)])
@pytest.mark.parametrize("n_sk,dim", [(40, 100), (40, 101), (37, 100), (48, 53), (24, 75), (31, 90), (32, 104)])
def test_pair_kernel_unaligned_outputs(engine, oracle, lane, offs, n_sk, dim):
    """As tests/test_gpu_lane_kernel.py: outputs that start anywhere inside a 64-byte line, rows that do not fill whole lines; the guard
    words around the arrays stay untouched."""
    import torch
    lane.setenv("PDMP_FORCE_SEGMENTS", "3")
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, "banana")
    C = 40
    x, v = make_init("ZigZagAD", dim, C, seed=4, target="banana")
    dev = torch.device("cuda:0")
    guard = -12345.678
    bufs, out = {}, {}
    for name, shape, off in (("X", (C, n_sk, dim), offs[0]), ("V", (C, n_sk, dim), offs[1]), ("t", (C, n_sk), offs[2])):
        n = int(np.prod(shape))
        b = torch.full((n + 16,), guard, dtype=torch.float64, device=dev)
        bufs[name] = (b, off, n)
        out[name] = b[off:off + n].view(shape)
    from pdmpflux.jl_amd import ensemble
    got = ensemble.sample_skeleton_device(s, n_sk, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=6, out=out)
    assert s.last_launch_form() == 7
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 6, diagnostics=False)
    for name in ("X", "V", "t"):
        b, off, n = bufs[name]
        assert np.array_equal(got[name].cpu().numpy(), ref[name]), name
        assert bool((b[:off] == guard).all()) and bool((b[off + n:] == guard).all()), name


@pytest.mark.parametrize("target,dim", [("iso", 100), ("banana", 77)])
def test_pair_kernel_time_horizon_chunks(engine, oracle, lane, target, dim):
    """The in-kernel time horizon on the pair kernel: the first chunk and every resumed chunk run in the SAME form (the hand-off state
    has that kernel's layout), every chain's packed history equals the checker's (tests/test_gpu_time_horizon_chunks.py)."""
    from test_gpu_time_horizon_chunks import _run as run_until
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, target)
    C, T = 75, 1.2
    x, v = make_init("ZigZagAD", dim, C, seed=5, target=target)
    stats = {}
    run_until(engine, oracle, s, cfg, x, v, T, 4, 16, stats, lo=7)
    assert s.last_launch_form() == 7 and stats["chunks"] >= 2


# ---- the pair form at d = 25 .. 52 (two coordinates of a slot per lane, GROUP(4)): mid-sized launches; PDMP_LANE=2 forces it

@pytest.fixture
def quad(monkeypatch):
    monkeypatch.setenv("PDMP_LANE", "2")
    return monkeypatch


def _runq(engine, oracle, dim, target, C, n_sk, seed=3, init_seed=0, diagnostics=True, **kw):
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, target, **kw)
    assert s.group_lanes() == 4
    x, v = make_init("ZigZagAD", dim, C, seed=init_seed, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=seed, diagnostics=diagnostics)
    assert s.last_launch_form() == 7, "this case is meant to run in the pair form of the lane kernel"
    return s, cfg, x, v, got


@pytest.mark.parametrize("dim", [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 20, 23, 24, 25, 26, 27, 28, 29, 32, 33, 36, 37, 40, 41, 44, 45, 47, 48, 49, 50, 51, 52])   # (9 .. 26: round 4, G = 4 by default there too)
@pytest.mark.parametrize("target", ["iso", "banana"])
def test_quad_pair_matches_checker(engine, oracle, quad, dim, target):
    s, cfg, x, v, got = _runq(engine, oracle, dim, target, 75, 100)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 100, x, v, 3, diagnostics=True))


@pytest.mark.parametrize("target", ["iso", "banana"])
def test_quad_pair_equals_the_other_forms(engine, oracle, monkeypatch, target):
    """Group kernel (G = 4), one lane per chain, a pair of lanes per chain: one result."""
    outs = []
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("PDMP_LANE", mode)
        s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, target)
        x, v = make_init("ZigZagAD", 50, 167, seed=1, target=target)
        outs.append(engine.sample_skeleton_ensemble(s, 150, x, v, seed=11, diagnostics=True))
        assert s.last_launch_form() == {"0": 3, "1": 4, "2": 7}[mode]
    assert_bitwise(outs[1], outs[0])
    assert_bitwise(outs[2], outs[0])


@pytest.mark.parametrize("segments", [2, 7, 1000])
@pytest.mark.parametrize("target,dim", [("iso", 50), ("banana", 50), ("banana", 31), ("iso", 46), ("iso", 9), ("banana", 13), ("iso", 22),
                                         ("iso", 1), ("banana", 2), ("iso", 3), ("banana", 5), ("iso", 5), ("banana", 6), ("iso", 7), ("banana", 8)])
def test_quad_pair_segment_handoff(engine, oracle, quad, segments, target, dim):
    quad.setenv("PDMP_FORCE_SEGMENTS", str(segments))
    s, cfg, x, v, got = _runq(engine, oracle, dim, target, 40, 96)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 96, x, v, 3, diagnostics=True))


@pytest.mark.parametrize("dim", [2, 3, 5, 6, 7, 8, 27, 34, 41, 50, 52])
@pytest.mark.parametrize("target", ["iso", "banana"])
def test_quad_pair_without_diagnostics(engine, oracle, quad, dim, target):
    """The throughput instantiations (two waves per SIMD, no diagnostic arrays)."""
    quad.setenv("PDMP_FORCE_SEGMENTS", "4")
    s, cfg, x, v, got = _runq(engine, oracle, dim, target, 67, 72, seed=17, diagnostics=False)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 72, x, v, 17, diagnostics=False), keys=("X", "V", "t"))


@pytest.mark.parametrize("kw", [{"tmax": 0.0}, {"tmax": 0.05}, {"tmax": 20.0}, {"adaptive": False, "tmax": 0.4}])
def test_quad_pair_horizons(engine, oracle, quad, kw):
    s, cfg, x, v, got = _runq(engine, oracle, 50, "banana", 40, 60, **kw)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 60, x, v, 3, diagnostics=True))


@pytest.mark.parametrize("offs", [(0, 5, 2), (8, 3, 7), (3, 0, 0)])
@pytest.mark.parametrize("n_sk,dim", [(40, 50), (40, 51), (37, 50), (48, 27), (31, 45), (37, 3), (33, 5), (35, 6), (31, 7)])
def test_quad_pair_unaligned_outputs(engine, oracle, quad, offs, n_sk, dim):
    import torch
    quad.setenv("PDMP_FORCE_SEGMENTS", "3")
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, "banana")
    C = 40
    x, v = make_init("ZigZagAD", dim, C, seed=4, target="banana")
    dev = torch.device("cuda:0")
    guard = -12345.678
    bufs, out = {}, {}
    for name, shape, off in (("X", (C, n_sk, dim), offs[0]), ("V", (C, n_sk, dim), offs[1]), ("t", (C, n_sk), offs[2])):
        n = int(np.prod(shape))
        b = torch.full((n + 16,), guard, dtype=torch.float64, device=dev)
        bufs[name] = (b, off, n)
        out[name] = b[off:off + n].view(shape)
    from pdmpflux.jl_amd import ensemble
    got = ensemble.sample_skeleton_device(s, n_sk, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=6, out=out)
    assert s.last_launch_form() == 7
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 6, diagnostics=False)
    for name in ("X", "V", "t"):
        b, off, n = bufs[name]
        assert np.array_equal(got[name].cpu().numpy(), ref[name]), name
        assert bool((b[:off] == guard).all()) and bool((b[off + n:] == guard).all()), name


@pytest.mark.parametrize("target", ["iso", "banana"])
def test_quad_pair_time_horizon_chunks(engine, oracle, quad, target):
    from test_gpu_time_horizon_chunks import _run as run_until
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, target)
    x, v = make_init("ZigZagAD", 50, 75, seed=5, target=target)
    stats = {}
    run_until(engine, oracle, s, cfg, x, v, 1.6, 4, 16, stats, lo=7)
    assert s.last_launch_form() == 7 and stats["chunks"] >= 2


def test_quad_pair_default_selection(engine, oracle, monkeypatch):
    """No environment: 25 000 chains (the 4-GPU shard of the headline ensemble) take the pair form, 12 500 the group kernel, 43 000 one
    lane per chain; a strided sample of the 25 000 bit for bit against the checker."""
    monkeypatch.delenv("PDMP_LANE", raising=False)
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    for C, form in ((12500, 3), (25000, 7), (43000, 4)):
        x, v = make_init("ZigZagAD", 50, C)
        got = engine.sample_skeleton_ensemble(s, 48, x, v, seed=5, diagnostics=False)
        assert s.last_launch_form() == form, (C, s.last_launch_form())
        if C == 25000:
            for c in np.arange(0, C, 997):
                r1 = oracle.sample_skeleton(cfg, 48, x[c:c + 1], v[c:c + 1], 5, chain_id_offset=int(c), diagnostics=False)
                for k in ("X", "V", "t"):
                    assert np.array_equal(got[k][c], r1[k][0]), (k, c)


def _random_lane_cases(n, seed=20261004):
    rs = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        dim = int(rs.integers(27, 105))
        mode = "1" if dim > 52 else str(rs.choice(["1", "2"]))
        cases.append(dict(dim=dim, mode=mode, target=str(rs.choice(["iso", "banana"])), C=int(rs.integers(1, 150)),
                          n_sk=int(rs.integers(2, 90)), segments=int(rs.choice([0, 1, 2, 3, 7])), budget=int(rs.choice([0, 1, 2, 5, 9])),
                          diag=bool(rs.integers(0, 2)), tmax=float(rs.choice([0.0, 0.05, 0.5, 2.0, 20.0])), adaptive=bool(rs.integers(0, 4)),
                          lo=int(rs.integers(0, 1000)), seed=int(rs.integers(0, 10**6))))
    return cases


@pytest.mark.parametrize("case", _random_lane_cases(60), ids=lambda c: f"d{c['dim']}-m{c['mode']}-{c['target']}-C{c['C']}-n{c['n_sk']}-s{c['segments']}")
def test_random_lane_launches_match_checker(engine, oracle, monkeypatch, case):
    """Sixty random launches of the lane kernel's three forms (one lane per chain, pair at d <= 52, pair at d >= 53): dimension,
    target, number of chains (ragged batches, fewer chains than a batch), skeleton length, hand-off segments, bound instalments,
    horizon settings, diagnostic arrays or not, chain id offset -- bit for bit against the checker."""
    monkeypatch.setenv("PDMP_LANE", case["mode"])
    if case["segments"]:
        monkeypatch.setenv("PDMP_FORCE_SEGMENTS", str(case["segments"]))
    if case["budget"]:
        monkeypatch.setenv("PDMP_CELL_BUDGET", str(case["budget"]))
    kw = {"tmax": case["tmax"]}
    if not case["adaptive"] and case["tmax"] > 0:
        kw["adaptive"] = False
    s, cfg = make_pair(engine, oracle, "ZigZagAD", case["dim"], case["target"], **kw)
    x, v = make_init("ZigZagAD", case["dim"], case["C"], seed=case["seed"] % 997, target=case["target"])
    got = engine.sample_skeleton_ensemble(s, case["n_sk"], x, v, seed=case["seed"], diagnostics=case["diag"], chain_id_offset=case["lo"])
    assert s.last_launch_form() == (4 if (case["dim"] <= 52 and case["mode"] == "1") else 7)
    ref = oracle.sample_skeleton(cfg, case["n_sk"], x, v, case["seed"], diagnostics=case["diag"], chain_id_offset=case["lo"])
    if case["diag"]:
        assert_bitwise(got, ref)
    else:
        assert_bitwise(got, ref, keys=("X", "V", "t"))
