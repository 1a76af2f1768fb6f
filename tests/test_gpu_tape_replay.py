"""Tape replay (pdmp_set_tape) on EVERY kernel form of the engine -- LDS-form group kernel, register-form group
kernel, one lane per chain, wave per chain -- and for every sampler, ForwardECMC included: engine and checker replay the same
externally generated Exp(1) / U(0,1) / N(0,1) tapes (numpy's generators here; a dump of Julia's MersenneTwister made by
julia/dump_tape.jl in the intended use) and must agree bit for bit, every PDMPHistory field.  A reference-held fixture therefore
pins the kernels that produce the numbers, not a fallback (tests/test_julia_fixtures.py runs the default dispatch)."""
import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu

FORM_WAVE, FORM_GROUP, FORM_GROUP_REG, FORM_LANE = 1, 2, 3, 4

# (kind, dim, target, kwargs, env, expected kernel form)
CASES = [
    # the headline workloads on the forms that serve them with the Philox stream
    ("ZigZagAD", 50, "iso", {}, {}, FORM_GROUP_REG),                         # config 2 on small ensembles: G = 4, K = 13, FULL register kernel
    ("ZigZagAD", 50, "iso", {}, {"PDMP_LANE": "1"}, FORM_LANE),              # config 2 at full size: one lane per chain
    ("ZigZagAD", 50, "banana", {}, {"PDMP_LANE": "1"}, FORM_LANE),           # the metric's second target
    ("ZigZagAD", 50, "banana", {}, {}, FORM_GROUP_REG),
    ("BPS", 50, "banana", {}, {}, FORM_GROUP_REG),                           # config 3
    ("Boomerang", 100, "iso", {"refresh_rate": 0.5}, {}, FORM_GROUP_REG),    # config 5 (finite-difference slopes), G = 8
    ("BoomerangAD", 100, "iso", {"refresh_rate": 0.5}, {}, FORM_GROUP_REG),
    # the shapes of julia/dump_tape.jl
    ("ZigZagAD", 6, "iso", {}, {}, FORM_GROUP_REG),                          # G = 1: 13-slot kernel with the per-slot skips
    ("ZigZagAD", 6, "banana", {}, {}, FORM_GROUP_REG),
    ("ZigZagAD", 50, "diag", {}, {}, FORM_GROUP_REG),
    ("ZigZag", 6, "iso", {}, {}, FORM_GROUP),                                # finite-difference Zig-Zag runs in the LDS form
    ("BPS", 6, "banana", {}, {}, FORM_GROUP_REG),
    ("BPS", 2, "banana", {"tmax": 20.0, "adaptive": False}, {}, FORM_GROUP_REG),   # the error path (ar > 1)
    ("Boomerang", 6, "diag", {"refresh_rate": 0.0}, {}, FORM_GROUP_REG),
    ("ForwardECMC", 6, "iso", {}, {}, FORM_GROUP),
    ("ForwardECMCAD", 6, "banana", {}, {}, FORM_GROUP),
    # every other sampler / form
    ("ForwardECMC", 50, "banana", {"ran_p": True}, {}, FORM_GROUP),
    ("ForwardECMC", 20, "diag", {"switch": False}, {}, FORM_GROUP),
    ("ForwardECMC", 3, "iso", {"mix_p": 1.0}, {}, FORM_GROUP),
    ("SpeedUpZigZagAD", 12, "iso", {}, {}, FORM_GROUP),
    ("RHMC", 10, "banana", {}, {}, FORM_GROUP),
    ("ZigZagAD", 50, "iso", {}, {"PDMP_REG": "0"}, FORM_GROUP),              # LDS form of the headline workload
    ("BPS", 20, "banana", {"refresh_rate": 0.5}, {"PDMP_REG": "0"}, FORM_GROUP),
    ("Boomerang", 9, "diag", {}, {"PDMP_REG": "0"}, FORM_GROUP),
    ("ZigZagAD", 300, "iso", {}, {}, FORM_GROUP),                            # K > 13: LDS form, G = 32
    ("ZigZagAD", 8, "iso", {"grid_size": 0}, {}, FORM_GROUP),                # Brent constant bound
    ("ZigZagAD", 50, "iso", {}, {"PDMP_TAPE_WAVE": "1"}, FORM_WAVE),         # round 1's replaying form
    ("ZigZagAD", 7, "iso", {"grid_size": 6}, {}, FORM_GROUP),                # grid sizes other than 10: the LDS-form group kernel (a run-time grid_size since round 5)
    ("ZigZagAD", 7, "iso", {"grid_size": 6}, {"PDMP_GROUP_LANES": "0"}, FORM_WAVE),   # ... or, asked for, the wave-per-chain kernel as in rounds 1-4
    ("ZigZagAD", 40, "dense", {}, {}, FORM_GROUP),                           # round 4: the dense-precision target on the per-chain mat-vec kernel
    ("BPS", 33, "dense", {"refresh_rate": 0.5}, {}, FORM_GROUP),
    ("BoomerangAD", 33, "dense", {}, {}, FORM_GROUP),
]


def _tapes(rs, C, n_sk, dim):
    # (64 draws per event and kind: the fixed, far too long horizon of the error-path case rejects dozens of proposals per event)
    return (rs.exponential(size=(C, 64 * n_sk)), rs.random(size=(C, 64 * n_sk)), rs.normal(size=(C, 3 * n_sk * max(dim, 2))))


@pytest.mark.parametrize("kind,dim,target,kw,env,form", CASES,
                         ids=[f"{c[0]}-d{c[1]}-{c[2]}-{'-'.join(f'{k}={v}' for k, v in {**c[3], **c[4]}.items()) or 'default'}" for c in CASES])
def test_tape_replay_on_every_kernel_form(engine, oracle, monkeypatch, kind, dim, target, kw, env, form):
    for k, val in env.items():
        monkeypatch.setenv(k, val)
    rs = np.random.default_rng(11)
    C, n_sk = 70, 120                                            # 70 chains: two batches of the lane kernel, one of them ragged
    tape = _tapes(rs, C, n_sk, dim)
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    G = s.group_lanes()
    s.set_tape(*tape)
    if form == FORM_WAVE:                                        # the wave-per-chain kernel sums in the WAVE64 order
        cfg.arith, cfg.group_lanes = oracle.ARITH_WAVE64, 1
        assert s.group_lanes() == 0
    else:
        assert s.group_lanes() == G > 0                          # a replay keeps the layout (and the summation order) of the Philox run
    cfg.tape = tape
    x, v = make_init(kind, dim, C, seed=3, target=target)
    lo = 3                                                       # tapes follow the GLOBAL chain id
    got = engine.sample_skeleton_ensemble(s, n_sk, x[lo:], v[lo:], seed=0, chain_id_offset=lo, diagnostics=True)
    assert s.last_launch_form() == form
    ref = oracle.sample_skeleton(cfg, n_sk, x[lo:], v[lo:], 0, chain_id_offset=lo, diagnostics=True)
    assert_bitwise(got, ref)
    # the tapes really are the random source: another seed changes nothing, other tapes change everything
    again = engine.sample_skeleton_ensemble(s, n_sk, x[lo:], v[lo:], seed=12345, chain_id_offset=lo, diagnostics=True)
    assert_bitwise(again, ref)
    s.clear_tape()
    philox = engine.sample_skeleton_ensemble(s, n_sk, x[lo:], v[lo:], seed=0, chain_id_offset=lo, diagnostics=False)
    assert not np.array_equal(philox["t"], ref["t"])


def test_sticky_tape_replay(engine, oracle):
    """StickyZigZag (src/StickySamplingLoop.jl): the thaw clock's exponential and the thawed coordinate's uniform come from the tapes too."""
    from test_gpu_sticky import pair                             # the pair builder of the Sticky parity tests
    d, C, n_sk = 6, 9, 150
    s, cfg = pair(engine, oracle, "StickyZigZagAD", d, "iso", 0.5)
    x, v = make_init("ZigZag", d, C, seed=9)
    rs = np.random.default_rng(2)
    tape = _tapes(rs, C, n_sk, d)
    s.set_tape(*tape)
    cfg.tape = tape
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=0, diagnostics=True)
    ref = oracle.sample_skeleton(cfg, n_sk, x, v, 0, diagnostics=True)
    assert_bitwise(got, ref, keys=("X", "V", "t", "is_active", "horizon", "ar", "errored_bound", "error_value_ar", "rejected", "hitting_horizon"))
    assert (ref["is_active"] == 0).any()


@pytest.mark.parametrize("env", [{}, {"PDMP_LANE": "1"}, {"PDMP_REG": "0"}], ids=["register", "lane", "lds"])
def test_exhausted_tape_is_reported_on_every_form(engine, oracle, monkeypatch, env):
    for k, val in env.items():
        monkeypatch.setenv(k, val)
    rs = np.random.default_rng(4)
    C, d = 5, 50
    s, _ = make_pair(engine, oracle, "ZigZagAD", d, "iso")
    s.set_tape(rs.exponential(size=(C, 40)), rs.random(size=(C, 40)), rs.normal(size=(C, 4)))
    x, v = make_init("ZigZagAD", d, C)
    with pytest.raises(engine.ArgumentError):                    # 40 variates cannot feed 400 events
        engine.sample_skeleton_ensemble(s, 400, x, v, seed=0)
    with pytest.raises(engine.ArgumentError):                    # more chains than the tape holds
        engine.sample_skeleton_ensemble(s, 10, np.zeros((C + 1, d)), np.ones((C + 1, d)), seed=0)
    s.clear_tape()
    ok = engine.sample_skeleton_ensemble(s, 30, x, v, seed=3)
    assert np.isfinite(ok["t"]).all()
