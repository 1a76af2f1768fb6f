"""One-off random sweep of StickyZigZag (not part of the suite): random dimension, target, kappa, bound flags, hand-off segments; every case
against the checker bit for bit, `is_active` included -- the register form (round 4) serves the default flags with forward-mode slopes,
the LDS form the rest.  Usage (GPU box): python scripts/sticky_random_sweep.py <n_cases> <seed>; exit code 1 on any mismatch."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import pdmpflux.jl_amd as P
from oracle import oracle as O
from helpers import DIAG_KEYS, make_init, assert_bitwise
from test_gpu_sticky import pair

n, seed0 = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.default_rng(seed0)
bad = done = 0
forms = {}
for q in range(n):
    dim = int(rs.choice([2, 3, 5, 7, 9, 13, 14, 20, 27, 40, 50, 52, 53, 77, 100, 112, 130]))
    target = ["iso", "diag", "banana"][rs.integers(3)]
    kind = ["StickyZigZagAD", "StickyZigZagAD", "StickyZigZag"][rs.integers(3)]
    kw = {}
    if rs.random() < 0.15: kw["signed_bound"] = False
    if rs.random() < 0.15: kw["adaptive"] = False
    if rs.random() < 0.3: kw["tmax"] = float(rs.choice([0.3, 1.0, 5.0]))
    seg = int(rs.choice([0, 0, 2, 3, 7]))
    if seg: os.environ["PDMP_FORCE_SEGMENTS"] = str(seg)
    else: os.environ.pop("PDMP_FORCE_SEGMENTS", None)
    kap = float(rs.choice([0.05, 0.3, 0.7, 2.0]))
    C, n_sk, seed = int(rs.integers(1, 70)), int(rs.integers(2, 60)), int(rs.integers(1 << 30))
    try:
        s, cfg = pair(P, O, kind, dim, target, kap, **kw)
        x, v = make_init("ZigZag", dim, C, seed=seed % 1000, target=target)
        got = P.sample_skeleton_ensemble(s, n_sk, x, v, seed=seed, diagnostics=True)
        forms[s.last_launch_form()] = forms.get(s.last_launch_form(), 0) + 1
        assert_bitwise(got, O.sample_skeleton(cfg, n_sk, x, v, seed, diagnostics=True), keys=("X", "V", "t", "is_active") + DIAG_KEYS)
        done += 1
    except Exception as e:
        bad += 1
        print("MISMATCH", kind, dim, target, kap, kw, C, n_sk, seg, seed, "->", repr(e)[:300], flush=True)
    if (done + bad) % 50 == 0:
        print("progress", done, bad, flush=True)
print("checked", done, "failed", bad, "forms", forms)
sys.exit(1 if bad else 0)
