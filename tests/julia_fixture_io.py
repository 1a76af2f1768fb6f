"""Reader / comparer of the raw little-endian fixtures julia/dump_tape.jl writes (tests/test_julia_fixtures.py; the format itself is
exercised without Julia by tests/test_julia_fixture_format.py)."""
import json
import os

import numpy as np

RTOL = 1e-5


def load(meta_path):
    d = os.path.dirname(meta_path)
    meta = json.load(open(meta_path))
    rd = lambda f, dt: np.fromfile(os.path.join(d, f), dtype=dt)
    dim, n = int(meta["dim"]), int(meta["n_sk"])
    fx = {"meta": meta, "xinit": rd("xinit.f64", "<f8"), "vinit": rd("vinit.f64", "<f8"),
          "tape": tuple(np.atleast_2d(np.concatenate([rd(f, "<f8"), np.zeros(1)])) for f in ("exp.f64", "uni.f64", "nrm.f64")),   # never empty
          "X": rd("X.f64", "<f8").reshape(n, dim), "V": rd("V.f64", "<f8").reshape(n, dim), "t": rd("t.f64", "<f8"),
          "horizon": rd("horizon.f64", "<f8"), "ar": rd("ar.f64", "<f8"),
          "rejected": rd("rejected.i32", "<i4"), "hitting_horizon": rd("hitting_horizon.i32", "<i4"), "errored_bound": rd("errored_bound.i32", "<i4")}
    return fx


def oracle_config(O, fx, arith):
    m = fx["meta"]
    sid = {"ZigZag": O.ZIGZAG, "ZigZagAD": O.ZIGZAG, "BPS": O.BPS, "BPSAD": O.BPS, "Boomerang": O.BOOMERANG, "BoomerangAD": O.BOOMERANG,
           "ForwardECMC": O.FECMC, "ForwardECMCAD": O.FECMC}[m["ctor"]]
    dflt = {"ZigZag": dict(tmax=2.0, refresh_rate=0.0, vectorized_bound=True, deriv_mode=O.DERIV_FINITE_DIFF),
            "ZigZagAD": dict(tmax=2.0, refresh_rate=0.0, vectorized_bound=True, deriv_mode=O.DERIV_FORWARD),
            "BPS": dict(tmax=1.0, refresh_rate=0.1, vectorized_bound=False, deriv_mode=O.DERIV_FORWARD),
            "Boomerang": dict(tmax=1.0, refresh_rate=0.1, vectorized_bound=False, deriv_mode=O.DERIV_FINITE_DIFF),
            "ForwardECMC": dict(tmax=2.0, refresh_rate=0.0, vectorized_bound=False, deriv_mode=O.DERIV_FORWARD),       # ForwardEventChainMonteCarlo.jl:290-292
            "ForwardECMCAD": dict(tmax=2.0, refresh_rate=0.0, vectorized_bound=False, deriv_mode=O.DERIV_FORWARD)}[m["ctor"]]
    dflt.update({k: v for k, v in m["kwargs"].items()})
    tid = {"iso": O.T_GAUSS_ISO, "diag": O.T_GAUSS_DIAG, "banana": O.T_BANANA}[m["target"]]
    tp = np.asarray(m["target_params"], dtype=np.float64)
    return O.Config(sampler=sid, dim=int(m["dim"]), target_id=tid, target_params=tp if tp.size else None, arith=arith, tape=fx["tape"], **dflt)


def compare(got, fx, what):
    """Within 1e-5 up to the first discrete fork (none is expected when the draws are replayed); reports where and why otherwise."""
    n = fx["t"].shape[0]
    close = lambda a, b: np.abs(a - b) <= RTOL * np.maximum(np.abs(b), 1.0)
    disc = (got["rejected"][0] != fx["rejected"]) | (got["hitting_horizon"][0] != fx["hitting_horizon"]) | (got["errored_bound"][0] != fx["errored_bound"])
    cont = ~(close(got["t"][0], fx["t"]) & close(got["X"][0], fx["X"]).all(1) & close(got["V"][0], fx["V"]).all(1))
    bad = disc | cont
    k = int(np.argmax(bad)) if bad.any() else n
    assert k == n, f"{what}: differs from the Julia run at event {k} of {n} ({'decision' if disc[k] else 'value'})"


