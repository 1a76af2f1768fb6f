#!/usr/bin/env python3
"""Generates tests/golden/reference_order/*.npz -- the FROZEN reference-order fixtures.

Each holds a short skeleton computed by the checker in ARITH_REFERENCE: the order of operations of the Julia code itself (left-to-right
sums, per-dimension cumsum, the full tangent-intersection formula with its NaN -> 0 + clamp, the normalised Categorical walk, no fused
multiply-add, none of the engine order's algebraic identities).  That code path does not move when a kernel is optimised: the engine
orders (GROUP / WAVE64) and their fixtures (tests/golden/*.npz) do.

RULES
  * This set is written ONCE.  The script refuses to overwrite an existing file; there is no --force.
  * It is never regenerated in a commit that touches a kernel, the engine-order code of the checker or the other fixtures.  If a
    fixture here stops reproducing, the reference-order restatement has changed: that is a finding, not a reason to regenerate.
  * Round 3 generated it from the checker of that round AND, byte for byte identically, from the checker as it was before the first
    engine-order identity was introduced (commit c1d3439; scripts/verify_reference_golden.py) -- i.e. these bytes predate every
    shortcut they are meant to guard against.
Like every fixture of this repository they are outputs of the RESTATEMENT, not of PDMPFlux.jl (no Julia toolchain in this pipeline).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_order")
OUT_V2 = os.path.join(HERE, "reference_order_v2")   # round 5: the same nineteen cases under arithmetic contract v2 (`--rng-version 2`), frozen once in the same way

# name, sampler, kwargs (the reference constructors' defaults spelled out), target, dim
def cases(O, rng_version=1):
    zzad = dict(sampler=O.ZIGZAG, grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=True, deriv_mode=O.DERIV_FORWARD)
    zzfd = dict(zzad, deriv_mode=O.DERIV_FINITE_DIFF)
    bps = dict(sampler=O.BPS, grid_size=10, tmax=1.0, refresh_rate=0.1, vectorized_bound=False, deriv_mode=O.DERIV_FORWARD)
    boo = dict(sampler=O.BOOMERANG, grid_size=10, tmax=1.0, refresh_rate=0.1, vectorized_bound=False, deriv_mode=O.DERIV_FINITE_DIFF)
    booad = dict(sampler=O.BOOMERANG, grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=False, deriv_mode=O.DERIV_FORWARD)
    fe = dict(sampler=O.FECMC, grid_size=10, tmax=2.0, refresh_rate=0.0, vectorized_bound=False, deriv_mode=O.DERIV_FORWARD)
    return [
        ("zigzagad_iso_d6", zzad, "iso", 6),
        ("zigzagad_banana_d6", zzad, "banana", 6),
        ("zigzagad_iso_d50", zzad, "iso", 50),                  # BASELINE config 2
        ("zigzagad_banana_d50", zzad, "banana", 50),            # the metric's second target
        ("zigzagad_diag_d50", zzad, "diag", 50),
        ("zigzagad_iso_d10", zzad, "iso", 10),                  # BASELINE config 1
        ("zigzag_fd_iso_d6", zzfd, "iso", 6),
        ("zigzagad_unsigned_banana_d6", dict(zzad, signed_bound=False), "banana", 6),
        ("zigzagad_brent_iso_d3", dict(zzad, grid_size=0), "iso", 3),
        ("bps_banana_d6", bps, "banana", 6),
        ("bps_banana_d50", bps, "banana", 50),                  # BASELINE config 3
        ("bps_errorpath_d2", dict(bps, tmax=20.0, adaptive=False), "banana", 2),
        ("bps_gaussian_velocity_diag_d9", dict(bps, gaussian_velocity=True, refresh_rate=0.5), "diag", 9),
        ("boomerang_fd_iso_d100", dict(boo, refresh_rate=0.5), "iso", 100),      # BASELINE config 5
        ("boomerangad_iso_d100", dict(booad, refresh_rate=0.5), "iso", 100),
        ("boomerang_fd_diag_d6", boo, "diag", 6),
        ("forwardecmc_banana_d6", fe, "banana", 6),
        ("forwardecmc_ranp_iso_d20", dict(fe, ran_p=True), "iso", 20),
        ("forwardecmc_fullrefresh_diag_d9", dict(fe, switch=False), "diag", 9),
    ] + ([] if rng_version == 1 else [
        ("boomerang_fd_iso_d100_baseline", boo, "iso", 100),                     # BASELINE config 5 at the constructor's own refresh_rate = 0.1 (the v1 set holds 0.5 only)
    ])


def build_case(O, name, kw, target, d, C=2, n_sk=40, seed=20260104, rng_version=1):
    rs = np.random.default_rng(sum(map(ord, name)) + 17)
    params = None
    tid = {"iso": O.T_GAUSS_ISO, "diag": O.T_GAUSS_DIAG, "banana": O.T_BANANA}[target]
    if target == "diag":
        params = np.concatenate([rs.normal(size=d), np.exp(rs.uniform(-1, 1, size=d))])
    cfg = O.Config(dim=d, target_id=tid, target_params=params, arith=O.ARITH_REFERENCE, rng_version=rng_version, **kw)
    x = rs.normal(size=(C, d))
    if target == "banana":
        x = 1 + 0.1 * x
    if kw["sampler"] == O.ZIGZAG:
        v = rs.choice([-1.0, 1.0], size=(C, d))
    else:
        v = rs.normal(size=(C, d))
        if kw["sampler"] in (O.BPS, O.FECMC):
            v /= np.linalg.norm(v, axis=1, keepdims=True)
    out = O.sample_skeleton(cfg, n_sk, x, v, seed, chain_id_offset=3)
    smp = O.sample_from_skeleton(cfg, 13, out["X"], out["V"], out["t"])
    keys = ("X", "V", "t", "horizon", "ar", "errored_bound", "error_value_ar", "rejected", "hitting_horizon")
    return dict(name=name, kwargs=repr(kw), target=target, dim=d, seed=seed, chain_id_offset=3, n_sk=n_sk, xinit=x, vinit=v,
                params=np.zeros(0) if params is None else params, samples=smp, rng_version=rng_version, **{k: out[k] for k in keys})


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import oracle as O
    ver = int(sys.argv[sys.argv.index("--rng-version") + 1]) if "--rng-version" in sys.argv else 1
    out_dir = OUT if ver == 1 else OUT_V2
    os.makedirs(out_dir, exist_ok=True)
    for name, kw, target, d in cases(O, ver):
        path = os.path.join(out_dir, name + ".npz")
        if os.path.exists(path):
            print(name, "exists: kept (this set is frozen)")
            continue
        np.savez_compressed(path, **build_case(O, name, kw, target, d, rng_version=ver))
        print(name, "written")
