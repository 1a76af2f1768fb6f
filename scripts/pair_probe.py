"""Probe one launch of the lane-pair kernel against the checker: python scripts/pair_probe.py <dim> <target> <diag 0|1> [C] [n_sk]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import importlib
engine = importlib.import_module("pdmpflux.jl_amd")
from oracle import oracle
oracle.build()
from helpers import make_pair, make_init, assert_bitwise
dim, target, diag = int(sys.argv[1]), sys.argv[2], bool(int(sys.argv[3]))
C = int(sys.argv[4]) if len(sys.argv) > 4 else 75
n_sk = int(sys.argv[5]) if len(sys.argv) > 5 else 90
s, cfg = make_pair(engine, oracle, "ZigZagAD", dim, target)
x, v = make_init("ZigZagAD", dim, C, seed=0, target=target)
got = engine.sample_skeleton_ensemble(s, n_sk, x, v, seed=3, diagnostics=diag)
print("form", s.last_launch_form(), flush=True)
ref = oracle.sample_skeleton(cfg, n_sk, x, v, 3, diagnostics=diag)
assert_bitwise(got, ref) if diag else assert_bitwise(got, ref, keys=("X", "V", "t"))
print("OK", dim, target, diag, flush=True)
