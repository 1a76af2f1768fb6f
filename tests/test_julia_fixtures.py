"""Reference-held fixtures: skeletons produced by PDMPFlux.jl itself, with the MersenneTwister variates it consumed
logged in call order (julia/dump_tape.jl), replayed draw for draw through the checker (orc_config_t.tape_*) and the engine
(pdmp_set_tape) and compared at the north-star tolerance.

This pipeline has no Julia toolchain, so tests/golden/julia/ is empty and these tests report `skipped: no reference-held fixture`;
a maintainer with Julia and PDMPFlux.jl v0.4.6 runs

    julia --project=/path/to/PDMPFlux.jl julia/dump_tape.jl tests/golden/julia

once and commits the output -- from then on the parity chain is pinned to the real reference instead of the restatement."""
import glob
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "julia", "*", "meta.json")))

if not CASES:
    pytest.skip("no reference-held fixture under tests/golden/julia (run julia/dump_tape.jl with PDMPFlux.jl v0.4.6)", allow_module_level=True)


from julia_fixture_io import compare, load, oracle_config   # noqa: E402


@pytest.mark.parametrize("meta", CASES, ids=[os.path.basename(os.path.dirname(p)) for p in CASES])
def test_oracle_replays_julia_fixture(oracle, meta):
    fx = load(meta)
    cfg = oracle_config(oracle, fx, oracle.ARITH_REFERENCE)
    got = oracle.sample_skeleton(cfg, int(fx["meta"]["n_sk"]), fx["xinit"][None], fx["vinit"][None], 0)
    compare(got, fx, "checker (reference order)")


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["default", "lane", "lds", "wave"])
@pytest.mark.parametrize("meta", CASES, ids=[os.path.basename(os.path.dirname(p)) for p in CASES])
def test_engine_replays_julia_fixture(engine, monkeypatch, meta, form):
    """The replay runs on the kernel form the same launch uses with the Philox stream (`default`: register-form group kernel where
    one exists), and on every other form that can serve the case: one lane per chain (Zig-Zag, d = 49..52), LDS-form group
    kernel, wave per chain."""
    env = {"lane": ("PDMP_LANE", "1"), "lds": ("PDMP_REG", "0"), "wave": ("PDMP_TAPE_WAVE", "1")}.get(form)
    if env:
        monkeypatch.setenv(*env)
    fx = load(meta)
    m = fx["meta"]
    P = engine
    d = int(m["dim"])
    tp = np.asarray(m["target_params"], dtype=np.float64)
    target = {"iso": P.GaussianIso, "banana": P.Banana}.get(m["target"], lambda: P.GaussianDiag(tp[:d], tp[d:]))()
    s = getattr(P, m["ctor"])(d, target, **m["kwargs"])
    if form == "lane" and not (m["ctor"] == "ZigZagAD" and 49 <= d <= 52 and m["target"] in ("iso", "banana")):
        pytest.skip("the lane kernel's replay serves Zig-Zag at d = 49..52 on the isotropic Gaussian / banana")
    if form == "wave" and m["ctor"].startswith("ForwardECMC"):
        pytest.skip("the wave-per-chain kernel serves Zig-Zag / BPS / Boomerang")
    s.set_tape(*fx["tape"])
    got = P.sample_skeleton_ensemble(s, int(m["n_sk"]), fx["xinit"][None], fx["vinit"][None], seed=0, diagnostics=True)
    kernel = {1: "wave per chain", 2: "group kernel, LDS form", 3: "group kernel, register form", 4: "one lane per chain"}[s.last_launch_form()]
    if form == "lane":
        assert s.last_launch_form() == 4
    compare(got, fx, f"engine (tape replay, {kernel})")
