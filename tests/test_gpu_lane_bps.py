"""The one-lane-per-chain form of the Bouncy Particle Sampler (csrc/lane_bps_kernel.hpp) -- built and MEASURED in round 3 (config 3:
58.7 ms against the group kernel's 40.3 ms: fewer instructions per event, but one wave per SIMD cannot hide the dependent chains of
the per-chain scalar work), therefore off unless PDMP_LANE_BPS=1.  It stays in the tree, tested, as the measured alternative: bit for bit
against the checker in the GROUP(4) order and against the group kernel it stands in for -- every dimension class it serves (7 .. 13
slots of four coordinates, d = 25 .. 52), both targets, signed / unsigned bound, Gaussian velocities, refresh rates from "never" to
"mostly", the fixed far-too-long horizon that exercises the ar > 1 path, hand-off between segments, ragged batches, global chain
ids, the time-horizon chunks.  pdmp_last_launch_form says which kernel ran."""
import numpy as np
import pytest

from helpers import DIAG_KEYS, assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu


@pytest.fixture
def lane(monkeypatch):
    monkeypatch.setenv("PDMP_LANE_BPS", "1")
    return monkeypatch


CASES = [
    ("BPS", 50, "banana", {}),                                          # config 3
    ("BPS", 50, "iso", {}),
    ("BPS", 27, "banana", {}), ("BPS", 28, "iso", {}), ("BPS", 29, "banana", {}), ("BPS", 33, "iso", {}), ("BPS", 36, "banana", {}),
    ("BPS", 37, "iso", {}), ("BPS", 41, "banana", {}), ("BPS", 44, "iso", {}), ("BPS", 48, "banana", {}),
    ("BPS", 49, "banana", {}), ("BPS", 51, "iso", {}), ("BPS", 52, "banana", {}),
    ("BPS", 50, "banana", {"signed_bound": False}),
    ("BPS", 50, "iso", {"Gaussian_velocity": True, "refresh_rate": 0.7}),
    ("BPS", 50, "banana", {"refresh_rate": 0.0}),                       # never refreshes (Q9: prob = 0 / 0 when the bounce rate is 0 too)
    ("BPS", 50, "banana", {"refresh_rate": 5.0}),                       # mostly refreshes: several refreshing chains per pass
    ("BPS", 45, "banana", {"tmax": 20.0, "adaptive": False}),           # the error path (ar > 1), the error ring in LDS
    ("BPSAD", 50, "banana", {}),                                        # BPSAD's defaults: refresh_rate = 0, tmax = 2
    ("BPS", 50, "iso", {"tmax": 0.0}),
]


@pytest.mark.parametrize("kind,dim,target,kw", CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}-{'-'.join(f'{k}={v}' for k, v in c[3].items()) or 'default'}" for c in CASES])
def test_bps_lane_kernel_matches_checker_and_group_kernel(engine, oracle, lane, kind, dim, target, kw):
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    assert s.group_lanes() == 4
    C, n_sk = 70, 90                                                     # two batches of 64, the second ragged
    x, v = make_init(kind, dim, C, seed=7, target=target)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=31, chain_id_offset=11, diagnostics=True)
    assert s.last_launch_form() == 4
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 31, chain_id_offset=11, diagnostics=True)
    assert_bitwise(got, ref)
    plain = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=31, chain_id_offset=11, diagnostics=False)   # the kernels without the diagnostic arrays
    assert_bitwise(plain, ref, keys=("X", "V", "t"))
    lane.setenv("PDMP_LANE_BPS", "0")
    grp = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=31, chain_id_offset=11, diagnostics=True)
    assert s.last_launch_form() == 3
    assert_bitwise(grp, ref)
    if kw.get("adaptive") is False and ref["errored_bound"].sum() > 0:       # (the ring of offending ar values: slot errored % 5)
        assert (ref["error_value_ar"][ref["errored_bound"] > 0][:, 1] > 1.0).all()


@pytest.mark.parametrize("segments", [2, 7])
@pytest.mark.parametrize("target", ["iso", "banana"])
def test_bps_lane_kernel_segment_handoff(engine, oracle, lane, segments, target):
    lane.setenv("PDMP_FORCE_SEGMENTS", str(segments))
    s, cfg = make_pair(engine, oracle, "BPS", 50, target)
    x, v = make_init("BPS", 50, 130, seed=3, target=target)
    got = engine.sample_skeleton_ensemble(s, 64, x, v, seed=5, diagnostics=True)
    assert s.last_launch_form() == 4
    assert_bitwise(got, oracle.sample_skeleton(cfg, 64, x, v, 5, diagnostics=True))


def test_bps_lane_kernel_is_opt_in(engine, oracle, monkeypatch):
    """Without PDMP_LANE_BPS=1 every BPS launch -- whatever its size -- runs on the group kernel (the faster form, measured)."""
    monkeypatch.delenv("PDMP_LANE_BPS", raising=False)
    s, cfg = make_pair(engine, oracle, "BPS", 50, "banana")
    C = 43000
    x, v = make_init("BPS", 50, C, target="banana")
    got = engine.sample_skeleton_ensemble(s, 12, x, v, seed=77, diagnostics=False)
    assert s.last_launch_form() == 3
    monkeypatch.setenv("PDMP_LANE_BPS", "1")
    lane = engine.sample_skeleton_ensemble(s, 12, x, v, seed=77, diagnostics=False)
    assert s.last_launch_form() == 4
    assert_bitwise(lane, got, keys=("X", "V", "t"))                       # 43 000 chains: more batches than resident waves
    fd = engine.BPS(50, engine.Banana(), AD_backend="FiniteDiff")
    engine.sample_skeleton_ensemble(fd, 8, x[:200], v[:200], seed=1, diagnostics=False)
    assert fd.last_launch_form() in (2, 3)                                # finite-difference slopes: not this kernel


def test_bps_lane_kernel_time_horizon_chunks(engine, oracle, lane):
    """The time horizon inside the kernel: resumed chunks continue in the lane form (its hand-off state has its own layout)."""
    import torch
    from pdmpflux.jl_amd import ensemble
    s, cfg = make_pair(engine, oracle, "BPS", 50, "banana")
    C, T = 150, 25.0
    x, v = make_init("BPS", 50, C, seed=5, target="banana")
    dev = torch.device("cuda:0")
    stats = {}
    got = ensemble.sample_skeleton_until_chunked_device(s, T, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=4, diagnostics=True,
                                                        init_capacity=8, stats=stats, chain_id_offset=3)
    assert s.last_launch_form() == 4 and stats["chunks"] >= 2
    ref = oracle.sample_skeleton_until(cfg, T, x, v, 4, init_capacity=64, chain_id_offset=3)
    cnt, off = got["count"].cpu().numpy(), got["offsets"].cpu().numpy()
    assert np.array_equal(cnt, ref["count"])
    for key in ("X", "V", "t") + DIAG_KEYS:
        a = got[key].cpu().numpy()
        for c in range(C):
            assert np.array_equal(a[off[c]:off[c + 1]], ref[key][c, :cnt[c]], equal_nan=True), (key, c)
