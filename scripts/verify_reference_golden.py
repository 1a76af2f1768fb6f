#!/usr/bin/env python3
"""One-off provenance check of tests/golden/reference_order/*.npz: builds the checker AS IT WAS at a given commit (default c1d3439,
the last one before the first engine-order identity -- "affine cells", f812efd) into a scratch directory, generates the same cases
with it, and compares with the committed files byte for byte.  Needs the git history (not part of any test run).

    python scripts/verify_reference_golden.py [commit]
"""
import importlib.util
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1] if len(sys.argv) > 1 else "c1d3439"
tmp = tempfile.mkdtemp(prefix="orc_old_")
subprocess.check_call(f"git -C {ROOT} archive {commit} oracle | tar -x -C {tmp}", shell=True)
subprocess.check_call(["make", "-C", os.path.join(tmp, "oracle"), "-s"], stderr=subprocess.DEVNULL)
spec = importlib.util.spec_from_file_location("old_oracle", os.path.join(tmp, "oracle", "oracle.py"))
O = importlib.util.module_from_spec(spec)
sys.modules["old_oracle"] = O                               # (dataclasses look the module up by name)
spec.loader.exec_module(O)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_reference_golden as M

bad = 0
for name, kw, target, d in M.cases(O):
    ref = np.load(os.path.join(M.OUT, name + ".npz"), allow_pickle=False)
    try:
        got = M.build_case(O, name, kw, target, d)
    except Exception as e:                                   # a sampler / keyword the old checker did not have yet
        print(f"{name}: not reproducible at {commit}: {type(e).__name__}: {e}")
        continue
    same = all(np.array_equal(np.asarray(got[k]), ref[k], equal_nan=True) for k in ("X", "V", "t", "horizon", "ar", "errored_bound",
                                                                                  "error_value_ar", "rejected", "hitting_horizon", "samples"))
    print(f"{name}: {'identical' if same else 'DIFFERENT'} at {commit}")
    bad += not same
sys.exit(1 if bad else 0)
