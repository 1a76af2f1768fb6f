"""oracle/julia_rng.c -- Julia's MersenneTwister(seed) with rand / randexp / randn, restated from the published algorithms (dSFMT-19937,
Marsaglia-Tsang ziggurat with Julia's tables, SHA-256 seeding of 1.11) as a TAPE PRODUCER for the checker.
Pinned here by every known answer that could be recalled without a Julia at hand -- values printed in Julia's own documentation and
countless forum posts for the pre-1.11 seeding, entries of the literal tables of Random/src/normal.jl, the FIPS 180 vectors -- and by
the round trip through the tape replay.  Unverified against a running Julia: the SHA-256 seeding of 1.11 and the draws inside the
ziggurat's unlikely branches."""
import ctypes as C

import numpy as np
import pytest


def test_mersenne_twister_known_answers(oracle):
    O = oracle
    # rand(MersenneTwister(1234), 5) of Julia 0.4 - 1.10 (docs of Random.seed!, the manual's "Random numbers" page)
    assert O.julia_draws(1234, 5, "rand").tolist() == [0.5908446386657102, 0.7667970365022592, 0.5662374165061859, 0.4600853424625171,
                                                       0.7940257103317943]
    assert O.julia_draws(0, 1, "rand")[0] == 0.8236475079774124
    assert O.julia_draws(1, 2, "rand").tolist() == [0.23603334566204692, 0.34651701419196046]
    # randn(MersenneTwister(1234), 4)
    assert O.julia_draws(1234, 4, "randn").tolist() == [0.8673472019512456, -0.9017438158568171, -0.4944787535042339, -0.9029142938652416]


def test_dsfmt_published_vector(oracle):
    """The known-answer file that ships with the dSFMT sources (dSFMT.19937.out.txt: `init_gen_rand(0)`, "generated randoms [1, 2)",
    printed with %.15f): the first eight values.  Pins the recursion, the masks, the initial mask and the period certification of
    oracle/julia_rng.c independently of Julia's seeding."""
    import ctypes as C
    import numpy as np
    L = oracle.lib()
    out = np.empty(8)
    L.jl_dsfmt_close1_open2(C.c_uint32(0), 8, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert ["%.15f" % x for x in out] == ["1.030581026769374", "1.213140320067012", "1.299002525016001", "1.381138853044628",
                                          "1.863488397063594", "1.133443440024236", "1.073465290616508", "1.924735636004535"]


def test_ziggurat_tables_match_normal_jl_entries(oracle):
    L = oracle.lib()
    u64p, dp = C.POINTER(C.c_uint64), C.POINTER(C.c_double)
    ki, wi, fi, ke, we, fe = u64p(), dp(), dp(), u64p(), dp(), dp()
    L.jl_zig_tables(C.byref(ki), C.byref(wi), C.byref(fi), C.byref(ke), C.byref(we), C.byref(fe))
    # 1-based in Julia: ki[1], ki[2], ki[3], ki[4]; ke[1..3]; the first widths
    assert [ki[i] for i in range(4)] == [0x0007799EC012F7B2, 0, 0x0006045F4C7DE363, 0x0006D1AA7D5EC0A5]
    assert [ke[i] for i in range(3)] == [0x000E290A13924BE3, 0, 0x0009BEADEBCE18BF]
    assert (wi[0], wi[1]) == (1.7367254121602630e-15, 9.5586603514556339e-17)
    assert (we[0], we[1]) == (1.9311480126418366e-15, 1.4178028487910829e-17)
    assert fi[0] == 1.0 and fe[0] == 1.0
    kin = np.array([ki[i] for i in range(256)], dtype=np.uint64)
    assert (kin[2:] < 2 ** 51).all() and (np.diff(np.array([fi[i] for i in range(256)])) < 0).all()   # strips: f decreasing from 1


def test_sha256_vectors(oracle):
    L = oracle.lib()
    out = (C.c_uint8 * 32)()
    L.jl_sha256(b"abc", 3, out)
    assert bytes(out).hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    L.jl_sha256(b"", 0, out)
    assert bytes(out).hex() == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"
    msg = b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq"
    L.jl_sha256(msg, len(msg), out)
    assert bytes(out).hex() == "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"


def test_variate_distributions(oracle):
    O = oracle
    for seeding in (O.JL_SEED_LIMBS, O.JL_SEED_SHA256):
        e = O.julia_draws(7, 200000, "randexp", seeding)
        z = O.julia_draws(7, 200000, "randn", seeding)
        u = O.julia_draws(7, 200000, "rand", seeding)
        assert abs(e.mean() - 1.0) < 0.01 and abs(e.var() - 1.0) < 0.03 and e.min() > 0
        assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02 and abs((z ** 4).mean() - 3.0) < 0.1
        assert abs(u.mean() - 0.5) < 0.005 and 0.0 <= u.min() and u.max() < 1.0
        assert (e > 7.69711747013105).sum() > 20 and (np.abs(z) > 3.6541528853610088).sum() > 20     # the tails of both ziggurats are reached
    assert not np.array_equal(O.julia_draws(7, 8, "rand", O.JL_SEED_LIMBS), O.julia_draws(7, 8, "rand", O.JL_SEED_SHA256))


@pytest.mark.parametrize("kind,dim,target,kw", [("zigzag", 10, "iso", {}), ("bps", 8, "banana", {"refresh_rate": 0.5, "tmax": 1.0}),
                                                ("boomerang", 6, "iso", {"refresh_rate": 0.3, "tmax": 1.0})])
def test_julia_source_run_replays_from_its_tapes(oracle, kind, dim, target, kw):
    """A run of the checker on the Julia source logs the variates it consumed; replayed from those tapes (the path a dump of a real Julia
    run takes, julia/dump_tape.jl) the checker reproduces the run bit for bit -- one interleaved stream in, three tapes out."""
    O = oracle
    sid = {"zigzag": O.ZIGZAG, "bps": O.BPS, "boomerang": O.BOOMERANG}[kind]
    tid = {"iso": O.T_GAUSS_ISO, "banana": O.T_BANANA}[target]
    cfg = O.Config(sampler=sid, dim=dim, target_id=tid, vectorized_bound=(sid == O.ZIGZAG), **kw)
    rs = np.random.default_rng(3)
    x, v = rs.normal(size=(4, dim)), (rs.choice([-1.0, 1.0], size=(4, dim)) if sid == O.ZIGZAG else rs.normal(size=(4, dim)))
    a = O.sample_skeleton_julia(cfg, 60, x, v, 42)
    cnt = a["tape_counts"]
    assert (cnt[:, 0] >= 59).all() and (cnt[:, 1] >= 59).all()
    if sid != O.ZIGZAG:
        assert (cnt[:, 2] > 0).any() and (cnt[:, 2] % dim == 0).all()          # refreshes draw whole vectors
    kwargs = {k: getattr(cfg, k) for k in cfg.__dataclass_fields__ if k != "_keep"}
    kwargs["tape"] = tuple(t[:, :max(int(cnt[:, q].max()), 1)] for q, t in enumerate(a["tape"]))
    b = O.sample_skeleton(O.Config(**kwargs), 60, x, v, 0)
    for k in ("X", "V", "t", "rejected", "hitting_horizon", "errored_bound"):
        assert np.array_equal(a[k], b[k]), k
    # chain c draws from MersenneTwister(seed + c): chain 1 of seed 42 is chain 0 of seed 43
    c = O.sample_skeleton_julia(cfg, 60, x[1:2], v[1:2], 43)
    assert np.array_equal(c["t"][0], a["t"][1])


def test_randn_fill_bulk_method(oracle):
    """`randn!(rng::MersenneTwister, A::Array{Float64})` (Random/src/normal.jl, recalled): fewer than 13 elements are scalar draws; from 13 on the
    array is first filled with the next n raw doubles and every element then goes through one ziggurat attempt, rejections drawing AFTER the
    fill.  Until the first rejected element the two orders give the same values; the first n raw draws decide which element that is."""
    for seed in range(1, 30):
        assert np.array_equal(oracle.julia_randn_fill(seed, 12), oracle.julia_randn_fill(seed, 12, scalar=True))
        assert np.array_equal(oracle.julia_randn_fill(seed, 12), oracle.julia_draws(seed, 12, "randn"))
    differ = 0
    for seed in range(1, 200):
        n = 100
        bulk, scal = oracle.julia_randn_fill(seed, n), oracle.julia_randn_fill(seed, n, scalar=True)
        if np.array_equal(bulk, scal):
            continue                                       # no element of the hundred was rejected (about half of the seeds)
        differ += 1
        k = int(np.argmax(bulk != scal))                   # the first rejected element: everything before it is the same draw in both orders
        assert np.array_equal(bulk[:k], scal[:k])
        # in the bulk order the elements AFTER the rejected one still come from raw draws k+1 .. n-1 of the fill: the scalar order has
        # shifted them by the draws the rejection consumed, so its element j > k equals a later bulk-order raw draw only by accident
        assert not np.array_equal(bulk[k + 1:], scal[k + 1:]) or k == n - 1
    assert 50 < differ < 150                               # P(no rejection in 100 draws) = 0.993^100 = 0.5
    z = np.concatenate([oracle.julia_randn_fill(s, 300) for s in range(300)])
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and abs((z ** 4).mean() - 3) < 0.15
