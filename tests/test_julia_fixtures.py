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
@pytest.mark.parametrize("meta", CASES, ids=[os.path.basename(os.path.dirname(p)) for p in CASES])
def test_engine_replays_julia_fixture(engine, meta):
    fx = load(meta)
    m = fx["meta"]
    P = engine
    d = int(m["dim"])
    tp = np.asarray(m["target_params"], dtype=np.float64)
    target = {"iso": P.GaussianIso, "banana": P.Banana}.get(m["target"], lambda: P.GaussianDiag(tp[:d], tp[d:]))()
    s = getattr(P, m["ctor"])(d, target, **m["kwargs"])
    s.set_tape(*fx["tape"])
    got = P.sample_skeleton_ensemble(s, int(m["n_sk"]), fx["xinit"][None], fx["vinit"][None], seed=0, diagnostics=True)
    compare(got, fx, "engine (tape replay, wave-per-chain kernel)")
