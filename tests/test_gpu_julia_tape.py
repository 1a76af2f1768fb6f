"""The reference's own random source through the engine (second half): the checker runs on the restated Julia
MersenneTwister (oracle/julia_rng.c: chain c <- MersenneTwister(seed + c), randexp / rand / randn interleaved on ONE stream as in the
reference), logs the variates each chain consumed, and the engine replays those tapes (pdmp_set_tape) on the kernel form the launch
uses -- register-form group kernels, one lane per chain.  Engine and checker must agree bit for bit in the engine's summation order,
and the engine must stay within 1e-5 of the checker's REFERENCE-order run of the same source (the north star's tolerance).
Unverified against a running Julia: see the header of oracle/julia_rng.c for what is pinned and what is not."""
import numpy as np
import pytest

from helpers import assert_bitwise, make_init, make_pair

pytestmark = pytest.mark.gpu

CASES = [("ZigZagAD", 10, "iso", {}, {}, 3),                       # BASELINE config 1's sampler (d = 10), register form G = 1
         ("ZigZagAD", 50, "iso", {}, {"PDMP_LANE": "1"}, 4),       # config 2's kernel: one lane per chain
         ("ZigZagAD", 50, "banana", {}, {}, 3),
         ("BPS", 50, "banana", {}, {}, 3),                         # config 3
         ("Boomerang", 100, "iso", {"refresh_rate": 0.1}, {}, 3)]  # config 5 at its BASELINE parameters


@pytest.mark.parametrize("seeding", ["limbs", "sha256"])
@pytest.mark.parametrize("kind,dim,target,kw,env,form", CASES, ids=[f"{c[0]}-d{c[1]}-{c[2]}" for c in CASES])
def test_engine_replays_the_julia_source(engine, oracle, monkeypatch, kind, dim, target, kw, env, form, seeding):
    for k, val in env.items():
        monkeypatch.setenv(k, val)
    O = oracle
    mode = O.JL_SEED_LIMBS if seeding == "limbs" else O.JL_SEED_SHA256
    s, cfg = make_pair(engine, oracle, kind, dim, target, **kw)
    C, n_sk = 66, 100
    x, v = make_init(kind, dim, C, seed=9, target=target)
    ref = O.sample_skeleton_julia(cfg, n_sk, x, v, 42, seeding=mode)             # the engine's summation order, Julia's variates
    cnt = ref["tape_counts"]
    tape = tuple(t[:, :max(int(cnt[:, q].max()), 1)] for q, t in enumerate(ref["tape"]))
    s.set_tape(*tape)
    got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=0, diagnostics=True)
    assert s.last_launch_form() == form
    assert_bitwise(got, {k: ref[k] for k in got})
    # the same source in the reference's own order of operations: within the north star's 1e-5, decision for decision
    kw_ref = {k: getattr(cfg, k) for k in cfg.__dataclass_fields__ if k != "_keep"}
    kw_ref.update(arith=O.ARITH_REFERENCE, group_lanes=1)
    r0 = O.sample_skeleton_julia(O.Config(**kw_ref), n_sk, x, v, 42, seeding=mode)
    # per chain: the first event where a recorded decision or a value (1e-5 relative) differs between the two orders -- a fork: a 1-ulp
    # difference landing on the other side of a comparison, after which the two runs are different, equally valid realisations
    # (reflections amplify rounding: BPS / Boomerang share a long prefix, Zig-Zag chains do not fork at this length)
    dec = (r0["rejected"] != got["rejected"]) | (r0["hitting_horizon"] != got["hitting_horizon"]) | (r0["errored_bound"] != got["errored_bound"])
    val = ~np.isclose(got["t"], r0["t"], rtol=1e-5, atol=1e-12) | ~np.isclose(got["X"], r0["X"], rtol=1e-5, atol=1e-9).all(axis=2)
    bad = dec | val
    first = np.where(bad.any(axis=1), bad.argmax(axis=1), n_sk)
    if kind.startswith("ZigZag"):
        assert (first == n_sk).mean() >= 0.95, first
    else:
        assert np.median(first) >= 40, first
