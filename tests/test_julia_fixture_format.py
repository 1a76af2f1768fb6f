"""The fixture format of julia/dump_tape.jl, exercised without Julia: a run of the checker on numpy-generated tapes is written in
exactly that format (raw little-endian arrays, column-major X / V, meta.json) and read back through the same reader / comparer
tests/test_julia_fixtures.py uses -- so the day real fixtures arrive the harness is known to work."""
import json
import os

import numpy as np

from julia_fixture_io import compare, load, oracle_config


def test_fixture_format_roundtrip(oracle, tmp_path):
    O = oracle
    d, n_sk = 6, 120
    rs = np.random.default_rng(3)
    tape = (rs.exponential(size=(1, 900)), rs.random(size=(1, 900)), rs.normal(size=(1, 10)))
    x, v = 1 + 0.1 * rs.normal(size=(1, d)), rs.choice([-1.0, 1.0], size=(1, d))
    cfg = O.Config(sampler=O.ZIGZAG, dim=d, target_id=O.T_BANANA, arith=O.ARITH_REFERENCE, deriv_mode=O.DERIV_FORWARD, tape=tape)
    ref = O.sample_skeleton(cfg, n_sk, x, v, 0)
    case = tmp_path / "zigzagad_banana_d6"
    case.mkdir()
    put = lambda f, a, dt: np.asarray(a, dtype=dt).tofile(case / f)
    put("xinit.f64", x[0], "<f8"); put("vinit.f64", v[0], "<f8")
    for f, a in zip(("exp.f64", "uni.f64", "nrm.f64"), tape):
        put(f, a[0], "<f8")
    put("X.f64", ref["X"][0], "<f8"); put("V.f64", ref["V"][0], "<f8")        # (n_sk, d) C-order = Julia's d x n_sk column-major
    for f in ("t", "horizon", "ar"):
        put(f + ".f64", ref[f][0], "<f8")
    for f in ("rejected", "hitting_horizon", "errored_bound"):
        put(f + ".i32", ref[f][0], "<i4")
    json.dump({"name": "zigzagad_banana_d6", "ctor": "ZigZagAD", "kwargs": {}, "dim": d, "target": "banana", "target_params": [],
               "seed": 1, "n_sk": n_sk}, open(case / "meta.json", "w"))
    fx = load(str(case / "meta.json"))
    assert fx["X"].shape == (n_sk, d) and np.array_equal(fx["X"], ref["X"][0]) and fx["tape"][0].shape == (1, 901)
    got = O.sample_skeleton(oracle_config(O, fx, O.ARITH_REFERENCE), n_sk, fx["xinit"][None], fx["vinit"][None], 0)
    compare(got, fx, "round trip")
    wave = O.sample_skeleton(oracle_config(O, fx, O.ARITH_WAVE64), n_sk, fx["xinit"][None], fx["vinit"][None], 0)   # the engine's replay order
    compare(wave, fx, "wave order")
