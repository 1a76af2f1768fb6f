"""One-off random sweep of the one-lane-per-chain kernel (not part of the suite): random dimension 9..52, target, horizon settings, number
of chains / events, hand-off segments and bound instalments, every case against the checker bit for bit (PDMP_LANE=1 forces the lane
kernel).  Usage (GPU box): python scripts/lane_random_sweep.py <n_cases> <seed> [dmin dmax [lane mode: 1 one lane, 2 pair form]];
prints mismatches, exit code 1 on any."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
DMIN, DMAX = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (9, 52)
MODE = sys.argv[5] if len(sys.argv) > 5 else "1"
os.environ["PDMP_LANE"] = MODE
import numpy as np
import pdmpflux.jl_amd as P
from oracle import oracle as O
from helpers import make_pair, make_init, assert_bitwise

n, seed0 = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.default_rng(seed0)
bad = done = 0
for q in range(n):
    dim = int(rs.integers(DMIN, DMAX + 1))
    target = ["iso", "banana"][rs.integers(2)] if dim >= 2 else "iso"
    kw = {}
    if rs.random() < 0.4:
        kw["tmax"] = float(rs.choice([0.0, 0.02, 0.3, 1.0, 5.0, 50.0]))
    if rs.random() < 0.25:
        kw["adaptive"] = False
    C = int(rs.choice([1, 5, 63, 64, 65, 100, 129, 200, 300]))
    n_sk = int(rs.choice([1, 2, 3, 9, 17, 40, 64, 100, 150]))
    seg = int(rs.choice([0, 0, 1, 2, 3, 5, 1000]))
    budget = int(rs.choice([0, 0, 1, 2, 3, 9]))
    for k, val in (("PDMP_FORCE_SEGMENTS", seg), ("PDMP_CELL_BUDGET", budget)):
        if val: os.environ[k] = str(val)
        else: os.environ.pop(k, None)
    seed = int(rs.integers(1, 10**6))
    try:
        s, cfg = make_pair(P, O, "ZigZagAD", dim, target, **kw)
        x, v = make_init("ZigZagAD", dim, C, seed=seed % 1000, target=target)
        if rs.random() < 0.2:
            x[:] = 0.0                                  # exact zeros: y = x v starts at +-0
        diag = bool(rs.random() < 0.5)              # without diagnostics the kernel instantiation is another one (DIAG = false)
        got = P.sample_skeleton_ensemble(s, n_sk, x, v, seed=seed, diagnostics=diag)
        assert s.last_launch_form() == (4 if MODE == "1" else 7), "unexpected kernel form"      # (round 3: no alignment preconditions any more)
        ref = O.sample_skeleton(cfg, n_sk, x, v, seed, diagnostics=diag)
        assert_bitwise(got, ref) if diag else assert_bitwise(got, ref, keys=("X", "V", "t"))
        done += 1
    except Exception as e:
        bad += 1
        print("MISMATCH", dim, target, kw, C, n_sk, seg, budget, seed, "->", repr(e)[:300], flush=True)
    if (done + bad) % 100 == 0:
        print("progress", done, bad, flush=True)
print("checked", done, "failed", bad)
sys.exit(1 if bad else 0)
