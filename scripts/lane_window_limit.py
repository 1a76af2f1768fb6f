"""One-off check of the lane kernel at the edge of its 31-bit window offsets (64 n_sk d 8 < 2^31): d = 50, n_sk = 83 880, 70 chains, bit for bit
against the checker; and just beyond it (the group kernel takes the launch).  Usage (GPU box): python scripts/lane_window_limit.py"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ["PDMP_LANE"] = "1"
import numpy as np
import pdmpflux.jl_amd as P
from oracle import oracle as O
from helpers import make_pair, make_init
for n_sk, form in ((83880, 4), (83888, 3)):
    s, cfg = make_pair(P, O, "ZigZagAD", 50, "banana")
    x, v = make_init("ZigZagAD", 50, 70, seed=3, target="banana")
    got = P.sample_skeleton_ensemble(s, n_sk, x, v, seed=5, diagnostics=False)
    assert s.last_launch_form() == form, (n_sk, s.last_launch_form())
    ref = O.sample_skeleton(cfg, n_sk, x, v, 5, diagnostics=False)
    for k in ("X", "V", "t"):
        assert np.array_equal(got[k], ref[k]), (n_sk, k)
    print("n_sk", n_sk, "form", form, "bit-identical", flush=True)
