"""One-off wider random sweep (not part of the suite): the generator of test_random_configurations_match_checker with
other seeds.  Usage (GPU box): python scripts/random_sweep.py <n_cases> <seed> ; prints mismatches, exit code 1 on any."""
import sys, traceback
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import pdmpflux.jl_amd as P
from oracle import oracle as O
from helpers import make_pair, make_init, assert_bitwise
from test_gpu_parity import _random_cases

n, seed0 = int(sys.argv[1]), int(sys.argv[2])
bad = done = 0
for (kind, dim, target, kw, C, n_sk, seed) in _random_cases(n, seed0):
    # (round 5: grid_size 5 / 33 run on the group kernel for every sampler and dimension -- nothing is skipped any more)
    try:
        s, cfg = make_pair(P, O, kind, dim, target, **kw)
        x, v = make_init(kind, dim, C, seed=seed % 1000, target=target)
        got = P.sample_skeleton_ensemble(s, n_sk, x, v, seed=seed, diagnostics=True)
        ref = O.sample_skeleton(cfg, n_sk, x, v, seed, diagnostics=True)
        assert_bitwise(got, ref)
        done += 1
    except Exception as e:                      # keep going: report every failing configuration
        bad += 1
        print("MISMATCH", kind, dim, target, kw, C, n_sk, seed, "->", repr(e)[:200], flush=True)
    if (done + bad) % 100 == 0:
        print("progress", done, bad, flush=True)
print("checked", done, "failed", bad)
sys.exit(1 if bad else 0)
