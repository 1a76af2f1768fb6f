"""The one-lane-per-chain kernel (csrc/lane_kernel.hpp) -- large Zig-Zag ensembles at d = 1..52 on the isotropic Gaussian and the
banana -- against the checker in the GROUP(4) order and against the group kernel it replaces: bit for bit, every array, every
diagnostic; hand-off between segments, ragged batches, bound instalments, the fall-back for speeds that are not +-1."""
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
    assert s.group_lanes() == 4
    x, v = make_init("ZigZagAD", dim, C, seed=init_seed, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=seed, diagnostics=diagnostics)
    import os
    if os.environ.get("PDMP_LANE") == "1":
        assert (n_sk * dim) % 8 == 0 and s.last_launch_form() == 4, "this case is meant to run in the lane kernel"
    return s, cfg, x, v, got


@pytest.mark.parametrize("dim", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 20, 21, 24, 25, 26, 27, 28, 29, 32, 33, 36, 37, 40, 41, 44, 45, 47, 48, 49, 50, 51, 52])   # (1 .. 26: round 4)
@pytest.mark.parametrize("target", ["iso", "banana"])
def test_lane_kernel_matches_checker(engine, oracle, lane, dim, target):
    if target == "banana" and dim < 2:
        pytest.skip("the banana has two twisted coordinates")
    s, cfg, x, v, got = _run(engine, oracle, dim, target, 150, 120)          # 150 chains: two full batches and a ragged one
    assert s.last_launch_form() == 4
    ref = oracle.sample_skeleton(cfg, 120, x, v, 3, diagnostics=True)
    assert_bitwise(got, ref)


@pytest.mark.parametrize("target", ["iso", "banana"])
def test_lane_kernel_equals_group_kernel(engine, oracle, monkeypatch, target):
    """The choice between the two kernels -- made per launch from the number of chains -- cannot be seen in the results."""
    monkeypatch.setenv("PDMP_LANE", "0")
    _, _, _, _, a = _run(engine, oracle, 50, target, 333, 200, seed=11)
    monkeypatch.setenv("PDMP_LANE", "1")
    _, _, _, _, b = _run(engine, oracle, 50, target, 333, 200, seed=11)
    assert_bitwise(b, a)


@pytest.mark.parametrize("segments", [2, 7, 1000])
@pytest.mark.parametrize("target,dim", [("iso", 50), ("banana", 50), ("banana", 31), ("iso", 46), ("iso", 9), ("banana", 12), ("iso", 17), ("banana", 24),
                                         ("iso", 1), ("banana", 2), ("iso", 3), ("banana", 4), ("iso", 5), ("banana", 5), ("iso", 6), ("banana", 7), ("iso", 8)])
def test_lane_kernel_segment_handoff(engine, oracle, lane, segments, target, dim):
    lane.setenv("PDMP_FORCE_SEGMENTS", str(segments))
    s, cfg, x, v, got = _run(engine, oracle, dim, target, 70, 96)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 96, x, v, 3, diagnostics=True))


@pytest.mark.parametrize("dim", [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 16, 18, 21, 24, 26, 27, 34, 41, 50, 52])
@pytest.mark.parametrize("target", ["iso", "banana"])
def test_lane_kernel_without_diagnostics(engine, oracle, lane, dim, target):
    """Launches that record no diagnostic arrays run another instantiation of the kernel (counters and error ring compiled out)."""
    lane.setenv("PDMP_FORCE_SEGMENTS", "4")
    s, cfg, x, v, got = _run(engine, oracle, dim, target, 131, 72, seed=17, diagnostics=False)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 72, x, v, 17, diagnostics=False), keys=("X", "V", "t"))


@pytest.mark.parametrize("budget", ["1", "2", "5", "9"])
def test_lane_kernel_bound_instalments(engine, oracle, lane, budget):
    lane.setenv("PDMP_CELL_BUDGET", budget)
    s, cfg, x, v, got = _run(engine, oracle, 50, "banana", 70, 60)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 60, x, v, 3, diagnostics=True))


@pytest.mark.parametrize("kw", [{"tmax": 0.0}, {"tmax": 0.05}, {"tmax": 20.0}, {"adaptive": False, "tmax": 0.4}])
def test_lane_kernel_horizons(engine, oracle, lane, kw):
    """Fixed and extreme horizons: bounds that run to the horizon (all nine cells), rejections."""
    s, cfg, x, v, got = _run(engine, oracle, 50, "banana", 70, 60, **kw)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 60, x, v, 3, diagnostics=True))
    # Which branches these cases reach (counted by the checker, equal in the engine by the comparison above): rejections in all of
    # them, horizon hits with the short horizons.  The `ar > 1` branch (erroneous_acceptance_rate!, SamplingLoopInplace.jl:131-151)
    # does not fire at d = 50 (the checker counts no erroneous bound in any of 70 x 59 events, here or with horizons up to 100: the 48
    # Gaussian coordinates dominate the rate); it does at d = 2, 3 with a fixed horizon of 20: test_lane_kernel_error_path below.
    assert got["rejected"].sum() > 0
    assert got["errored_bound"].sum() == 0
    if kw.get("tmax") in (0.05, 0.4):
        assert got["hitting_horizon"].sum() > 0


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("form", ["1", "2"])
def test_lane_kernel_error_path(engine, oracle, monkeypatch, dim, form):
    """The `ar > 1` branch of the lane kernels (erroneous_acceptance_rate!, SamplingLoopInplace.jl:131-151: the bound is rebuilt over half the
    horizon with a fresh exponential, the ratio goes into the five-slot ring) DOES run where the banana's two twisted coordinates carry
    most of the rate: d = 2, 3 with a fixed horizon of 20 (the checker counts 126 / 46 erroneous bounds in 70 x 199 events; none at d >= 5,
    which is why round 4's d = 50 case above saw none) -- on the one-lane form and on the pair form, bit for bit incl. errored_bound and
    error_value_ar."""
    monkeypatch.setenv("PDMP_LANE", form)
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, "banana", tmax=20.0, adaptive=False)
    x, v = make_init("ZigZagAD", dim, 70, target="banana")
    got = engine.sample_skeleton_ensemble(s, 200, x, v, seed=3, diagnostics=True)
    assert s.last_launch_form() == (4 if form == "1" else 7)
    ref = oracle.sample_skeleton(cfg, 200, x, v, 3, diagnostics=True)
    assert ref["errored_bound"].sum() > 10 and (ref["error_value_ar"] > 1.0).any()
    assert_bitwise(got, ref)


def test_lane_kernel_zero_start(engine, oracle, lane):
    """x = 0 exactly with both signs of v: y = x v starts at +-0."""
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    x = np.zeros((64, 50))
    v = np.where(np.random.default_rng(5).random((64, 50)) < 0.5, -1.0, 1.0)
    got = engine.sample_skeleton_ensemble(s, 40, x, v, seed=9, diagnostics=True)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 40, x, v, 9, diagnostics=True))


def test_lane_kernel_leaves_other_speeds_to_the_group_kernel(engine, oracle, lane):
    """|v_i| != 1 somewhere in the ensemble: the check kernel raises the flag and the group kernel does the launch's work."""
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    x, v = make_init("ZigZagAD", 50, 130)
    v[77, 13] = 0.5
    got = engine.sample_skeleton_ensemble(s, 50, x, v, seed=2, diagnostics=True)
    assert_bitwise(got, oracle.sample_skeleton(cfg, 50, x, v, 2, diagnostics=True))


def test_lane_kernel_full_machine(engine, oracle, monkeypatch):
    """Default selection (no environment): 65 600 chains take the lane kernel, more batches than resident waves, several segments per
    chain; a strided sample of chains bit for bit against the checker, global chain ids, invariants on every chain."""
    monkeypatch.delenv("PDMP_LANE", raising=False)
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "banana")
    C = 65600
    x, v = make_init("ZigZagAD", 50, C, target="banana")
    got = engine.sample_skeleton_ensemble(s, 96, x, v, seed=77, diagnostics=False)
    for c in np.arange(0, C, 1291):
        r1 = oracle.sample_skeleton(cfg, 96, x[c:c + 1], v[c:c + 1], 77, chain_id_offset=int(c), diagnostics=False)
        for k in ("X", "V", "t"):
            assert np.array_equal(got[k][c], r1[k][0]), (k, c)
    assert (np.diff(got["t"], axis=1) > 0).all()
    assert (np.abs(got["V"]) == 1.0).all()


@pytest.mark.parametrize("offs", [(0, 5, 2), (8, 3, 7), (0, 0, 1), (3, 0, 0), (9, 12, 3), (15, 1, 0)])     # (the last two: phases beyond 8 doubles -- the 128-byte lines of round 5)
@pytest.mark.parametrize("n_sk,dim", [(40, 50), (40, 51), (37, 50), (48, 27), (24, 38), (32, 45), (31, 45),
                                      (40, 2), (37, 3), (33, 5), (40, 5), (35, 6), (31, 7), (40, 8)])     # (rows shorter than a line: round 4)
def test_lane_kernel_unaligned_outputs(engine, oracle, lane, offs, n_sk, dim):
    """Output arrays that start anywhere inside a 64-byte line (views into larger buffers, as a sharded run's in-place exchange
    produces them) and chains whose rows do not fill whole lines (n_sk d not a multiple of 8).  Round 3: the lane kernel takes all of
    them -- the line phase of a chain's rows is per chain, column 0 is written row-exact, every later line is written whole from the
    chain's own bytes, the ends of every chain and segment go out as partial lines.  It must write exactly the bytes of the arrays:
    the guard words around them stay untouched, and (chains being contiguous) no chain's partial lines may clobber its neighbour's."""
    import torch
    lane.setenv("PDMP_FORCE_SEGMENTS", "3")
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, "banana")
    C = 70
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
    assert s.last_launch_form() == 4
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 6, diagnostics=False)
    for name in ("X", "V", "t"):
        b, off, n = bufs[name]
        assert got[name].data_ptr() == b.data_ptr() + 8 * off                    # the engine wrote into the view it was given
        assert np.array_equal(got[name].cpu().numpy(), ref[name]), name
        assert bool((b[:off] == guard).all()) and bool((b[off + n:] == guard).all()), name


@pytest.mark.parametrize("offs", [(0, 0, 0), (9, 12, 3), (15, 1, 0), (8, 8, 1)])
@pytest.mark.parametrize("n_sk,dim", [(24, 100), (23, 61), (25, 53), (21, 104)])
def test_pair_form_unaligned_outputs(engine, oracle, lane, offs, n_sk, dim):
    """The same for the pair form at d = 53 .. 104, which writes whole 128-byte lines since round 5 (carries of up to 15 doubles, rows of up to
    15 + 104 positions in 128 sign bits): every phase of X inside a line, V at another phase, rows that are no whole number of lines, three
    segments per chain; exactly the bytes of the arrays are written."""
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


def test_lane_kernel_is_what_ran(engine, oracle, monkeypatch):
    """pdmp_last_launch_form: 4 (one lane per chain) for a large unit-speed ensemble, 3 (group kernel, registers) when the check
    kernel finds another speed or the ensemble is small."""
    monkeypatch.delenv("PDMP_LANE", raising=False)
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    x, v = make_init("ZigZagAD", 50, 43000)
    engine.sample_skeleton_ensemble(s, 8, x, v, seed=1, diagnostics=False)
    assert s.last_launch_form() == 4
    v[5, 5] = 2.0
    engine.sample_skeleton_ensemble(s, 8, x, v, seed=1, diagnostics=False)
    assert s.last_launch_form() == 3
    engine.sample_skeleton_ensemble(s, 8, x[:100], np.sign(v[:100]), seed=1, diagnostics=False)
    assert s.last_launch_form() == 3
