"""The time horizon as a KERNEL feature (ABI 8: pdmp_sample_skeleton_until_device + pdmp_close_chunk_device +
pdmp_until_advance_device + pdmp_compact_chunk_device; host loop: ensemble.sample_skeleton_until_chunked_device) against the
checker's restatement of `sample_skeleton(sampler, T::Float64, ...)` (src/sample.jl:323-439, orc_sample_skeleton_until): chains stop
inside the kernel at their first event with t >= T, the others resume from their hand-off state in the next chunk, finished
chains cost nothing, memory is the sum of the chunks.  Every chain's packed history must equal the checker's, bit for bit,
whatever the chunking (tiny capacities force many chunks and many resumes), on the group kernels and on the lane kernel."""
import numpy as np
import pytest

from helpers import DIAG_KEYS, make_init, make_pair

pytestmark = pytest.mark.gpu

CASES = [
    ("ZigZagAD", 10, "iso", {}, 3.7), ("ZigZagAD", 50, "banana", {}, 2.2), ("ZigZag", 6, "diag", {}, 5.0),
    ("BPS", 8, "banana", {}, 3.1), ("BPSAD", 50, "iso", {}, 4.0),
    ("Boomerang", 12, "iso", {"refresh_rate": 0.5}, 6.3), ("BoomerangAD", 20, "diag", {"refresh_rate": 0.3}, 4.4),
    ("SpeedUpZigZagAD", 5, "iso", {}, 2.0), ("RHMC", 6, "banana", {}, 4.5), ("ForwardECMC", 9, "diag", {}, 3.3),
    ("ZigZagAD", 8, "iso", {"grid_size": 0}, 3.0),                       # Brent constant bound
    ("ZigZagAD", 300, "iso", {}, 0.4),                                   # LDS form, G = 32
]


def _run(engine, oracle, s, cfg, x, v, T, seed, cap, stats=None, lo=0, stream=None):
    import torch
    from pdmpflux.jl_amd import ensemble
    dev = torch.device("cuda:0")
    got = ensemble.sample_skeleton_until_chunked_device(s, T, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=seed,
                                                        diagnostics=True, init_capacity=cap, stats=stats, chain_id_offset=lo,
                                                        stream=stream)
    if stream is not None:
        stream.synchronize()
    ref = oracle.sample_skeleton_until(cfg, T, x, v, seed, init_capacity=max(cap, 64), chain_id_offset=lo)
    off = got["offsets"].cpu().numpy()
    cnt = got["count"].cpu().numpy()
    assert np.array_equal(cnt, ref["count"]), (cnt[:8], ref["count"][:8])
    assert off[0] == 0 and np.array_equal(off[1:], np.cumsum(cnt))
    host = {k: got[k].cpu().numpy() for k in ("X", "V", "t") + DIAG_KEYS}
    for c in range(x.shape[0]):
        k = int(cnt[c])
        for key in ("X", "V", "t") + DIAG_KEYS:
            a, b = host[key][off[c]:off[c + 1]], ref[key][c, :k]
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            assert same.all(), f"chain {c} {key}: first mismatch at {np.argwhere(~same)[0]} of {k} columns"
        assert host["t"][off[c + 1] - 1] == T or T == 0.0                # t[end] == T exactly (test/test_samplers.jl:56-62)
    return got, ref, cnt


@pytest.mark.parametrize("cap", [8, 64])
@pytest.mark.parametrize("kind,dim,target,kw,T", CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}-T{c[4]}" for c in CASES])
def test_chunked_time_horizon_matches_checker(engine, oracle, kind, dim, target, kw, T, cap):
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C = 21
    x, v = make_init(kind, dim, C, seed=2, target=target)
    stats = {}
    got, ref, cnt = _run(engine, oracle, s, cfg, x, v, T, 9, cap, stats)
    assert len(set(cnt.tolist())) > 1                                    # ragged
    if cap == 8 and cnt.max() > 9:
        assert stats["chunks"] >= 2 and stats["chains"][-1] <= C         # several resumes, fewer and fewer chains
    assert stats["chains"] == sorted(stats["chains"], reverse=True)


@pytest.mark.parametrize("dim", [50, 12, 5, 3])                          # (12, 5, 3: the lane kernels of round 4 -- 5 and 3: rows shorter than a 64-byte line)
@pytest.mark.parametrize("target", ["iso", "banana"])
def test_chunked_time_horizon_on_the_lane_kernel(engine, oracle, monkeypatch, target, dim):
    """PDMP_LANE=1: the first chunk runs one lane per chain; the resumed chunks -- a handful of chains -- must continue in the SAME
    form (the hand-off state has that kernel's layout: signed positions, sign words, batch-major)."""
    monkeypatch.setenv("PDMP_LANE", "1")
    s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, target)
    C, T = 150, 80.0 / dim                                               # three batches of 64 chains, the last one ragged; ~30 events per chain
    x, v = make_init("ZigZagAD", dim, C, seed=5, target=target)
    stats = {}
    _run(engine, oracle, s, cfg, x, v, T, 4, 16, stats, lo=7)            # chain_id_offset: streams keyed by the global chain id
    assert s.last_launch_form() == 4 and stats["chunks"] >= 2


def test_chunked_time_horizon_on_a_side_stream(engine, oracle):
    """A caller's stream that is NOT torch's current stream: the host loop's own torch ops (zero fills, the clone of the list of
    running chains, the `.item()` reads between chunks) must be ordered with the engine's kernels on that stream.  The current
    stream is kept busy with a long fill while the side stream runs the chunks, so an op left on the wrong stream shows."""
    import torch
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    x, v = make_init("ZigZagAD", 50, 300, seed=11)
    side = torch.cuda.Stream()
    busy = torch.empty((1 << 28,), dtype=torch.float64, device="cuda")
    for rep in range(3):
        busy.fill_(float(rep))                                           # ~2 GB fill queued on the current stream
        stats = {}
        _run(engine, oracle, s, cfg, x, v, 1.4, 6 + rep, 8, stats, stream=side)
        assert stats["chunks"] >= 2
    torch.cuda.synchronize()


def test_chunked_time_horizon_segment_handoff(engine, oracle, monkeypatch):
    """Several segments per chunk (PDMP_FORCE_SEGMENTS): a chain that reaches T in an early segment records nothing in the later ones."""
    monkeypatch.setenv("PDMP_FORCE_SEGMENTS", "5")
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 50, "iso")
    x, v = make_init("ZigZagAD", 50, 40, seed=8)
    stats = {}
    _run(engine, oracle, s, cfg, x, v, 1.1, 3, 200, stats)               # ~25 events per chain in a capacity of 200: one chunk, five segments
    assert stats["chunks"] == 1


def test_chunked_time_horizon_edge_cases(engine, oracle):
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 3, "iso")
    x, v = np.zeros((5, 3)), np.ones((5, 3))
    got, ref, cnt = _run(engine, oracle, s, cfg, x, v, 0.0, 1, 8)        # T == 0: the initial point alone (sample.jl:350-353)
    assert (cnt == 1).all()
    # an event at exactly t == T ends the loop without a closing column: replay a run with T = one of its event times
    fixed = engine.sample_skeleton_ensemble(s, 30, x[:1] + 0.1, v[:1], seed=5)
    T = float(fixed["t"][0, 17])
    got, ref, cnt = _run(engine, oracle, s, cfg, x[:1] + 0.1, v[:1], T, 5, 8)
    assert cnt[0] == 18 and got["t"][-1].item() == T
    import torch
    from pdmpflux.jl_amd import ensemble
    with pytest.raises(engine.ArgumentError):
        ensemble.sample_skeleton_until_chunked_device(s, -1.0, torch.zeros((2, 3), dtype=torch.float64, device="cuda"),
                                                      torch.ones((2, 3), dtype=torch.float64, device="cuda"), seed=1)


def test_chunked_path_memory_and_fallback(engine, oracle):
    """Memory is the sum of the chunks: 12 000 chains x T for ~130 events each from a first capacity of 64 -- the padded re-run needs
    chains x 256 columns, the chunks a fraction more than the histories themselves.  The dense-precision target is not served in-kernel: the
    device entry says UNSUPPORTED and the host falls back to the re-run."""
    import torch
    from pdmpflux.jl_amd import ensemble, _lib
    s, cfg = make_pair(engine, oracle, "ZigZagAD", 10, "iso")
    C, T = 12000, 33.0
    x, v = make_init("ZigZagAD", 10, C, seed=1)
    stats = {}
    dev = torch.device("cuda:0")
    got = ensemble.sample_skeleton_until_chunked_device(s, T, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=2, init_capacity=64, stats=stats)
    cnt = got["count"].cpu().numpy()
    ref = oracle.sample_skeleton_until(cfg, T, x[:30], v[:30], 2, init_capacity=64)
    assert np.array_equal(cnt[:30], ref["count"])
    assert stats["chunk_bytes"] < 1.4 * stats["dense_bytes"], stats      # (the re-run's last attempt alone: C x 256 columns = 2 x the histories)
    assert stats["chains"][0] == C and stats["chunks"] <= 5, stats      # the later chunks are sized from the chains' own event rates
    dn, _ = make_pair(engine, oracle, "ZigZagAD", 70, "dense")           # the dense-precision target is not served in-kernel
    xs, vs = make_init("ZigZagAD", 70, 4, seed=1)
    with pytest.raises(_lib.EngineError) as e:
        ensemble.sample_skeleton_until_chunked_device(dn, 0.5, torch.from_numpy(xs).to(dev), torch.from_numpy(vs).to(dev), seed=1)
    assert e.value.code == _lib.ERR_UNSUPPORTED
    packed = ensemble.sample_skeleton_until_device(dn, 0.5, torch.from_numpy(xs).to(dev), torch.from_numpy(vs).to(dev), seed=1, compact=True)
    assert packed["offsets"].shape[0] == 5 and packed["t"][-1].item() == 0.5


@pytest.mark.parametrize("dim,target,kap,T,cap", [(4, "iso", 0.5, 6.0, 8), (13, "diag", 0.3, 4.0, 16), (50, "iso", 0.7, 2.5, 24), (20, "banana", 0.5, 3.0, 8)])
def test_chunked_time_horizon_sticky(engine, oracle, dim, target, kap, T, cap):
    """StickyZigZag in chunks (round 3: the hand-off state carries the active masks): the resumed chunk starts with the frozen
    coordinates still frozen; is_active of every column and the closing column's masked flow (sample.jl:386-401) as the checker's."""
    import torch
    from pdmpflux.jl_amd import ensemble
    from test_gpu_time_horizon import sticky_pair
    s, cfg = sticky_pair(engine, oracle, dim, target, kap)
    C = 33
    x, v = make_init("ZigZag", dim, C, seed=4, target=target)
    dev = torch.device("cuda:0")
    stats = {}
    got = ensemble.sample_skeleton_until_chunked_device(s, T, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=17, diagnostics=True,
                                                        init_capacity=cap, stats=stats)
    ref = oracle.sample_skeleton_until(cfg, T, x, v, 17, init_capacity=64)
    cnt, off = got["count"].cpu().numpy(), got["offsets"].cpu().numpy()
    assert np.array_equal(cnt, ref["count"]) and stats["chunks"] >= 2
    for key in ("X", "V", "t", "is_active") + DIAG_KEYS:
        a = got[key].cpu().numpy()
        for c in range(C):
            assert np.array_equal(a[off[c]:off[c + 1]], ref[key][c, :cnt[c]], equal_nan=True), (key, c)
    assert (got["is_active"] == 0).any()


@pytest.mark.parametrize("cap", [8, 24, 40])
@pytest.mark.parametrize("kind,dim,T", [("ZigZagAD", 40, 1.5), ("ZigZagAD", 300, 0.3), ("BPS", 40, 4.0), ("BoomerangAD", 40, 6.0), ("BPS", 130, 12.0)])
def test_chunked_time_horizon_on_the_batched_dense_forms(engine, oracle, kind, dim, T, cap):
    """Round 4: the time horizon inside the batched dense-precision kernels (dense_batched.hip: `t_stop`, `resume`, `chain_map`, `count` --
    the workspace holds every chain's state between chunks, a resumed chunk's slots are the chains still running).  Several chunks with
    compaction from a small first capacity; against the checker's reference-order restatement: history lengths identical, times 1e-9
    (the batched forms are held to a tolerance: P x is carried by recurrences between re-basing products)."""
    import torch
    from pdmpflux.jl_amd import ensemble, _lib
    C_ = 70                                                              # (two ragged blocks of four chains at the end)
    s, cfg = make_pair(engine, oracle, kind, dim, "dense", arith=oracle.ARITH_REFERENCE)
    x, v = make_init(kind, dim, C_, seed=3)
    dev = torch.device("cuda:0")
    stats = {}
    got = ensemble.sample_skeleton_until_chunked_device(s, T, torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev), seed=11,
                                                        init_capacity=cap, stats=stats)
    assert s.last_launch_form() == 5 and _lib.lib().pdmp_dense_batched_last_mode() in (2, 3)
    ref = oracle.sample_skeleton_until(cfg, T, x, v, 11, init_capacity=64)
    cnt, off = got["count"].cpu().numpy(), got["offsets"].cpu().numpy()
    assert np.array_equal(cnt, ref["count"])
    assert len(set(cnt.tolist())) > 1
    if cnt.max() > cap + 1:
        assert stats["chunks"] >= 2, stats                               # resumed chunks
    print(kind, dim, cap, stats["capacities"], stats["chains"])          # (-s: which cases compact -- chains[-1] < 70)
    tt, X, V = got["t"].cpu().numpy(), got["X"].cpu().numpy(), got["V"].cpu().numpy()
    for c in range(C_):
        k = int(cnt[c])
        assert np.allclose(tt[off[c]:off[c + 1]], ref["t"][c, :k], rtol=1e-9, atol=0), c
        assert np.allclose(X[off[c]:off[c + 1]], ref["X"][c, :k], rtol=1e-8, atol=1e-10), c
        assert np.allclose(V[off[c]:off[c + 1]], ref["V"][c, :k], rtol=1e-8, atol=1e-10), c
        assert tt[off[c + 1] - 1] == T
