"""Build hygiene: no kernel of libpdmphip.so may lose a wave per SIMD against the frozen table (profiles/occupancy.json, written by
`python3 scripts/occupancy_table.py --write` from the compiler's resource-usage remarks of a build whose kernels were measured).  The
register-form and lane kernels sit a few registers below an occupancy threshold (128 / 168 / 256 VGPRs) by construction; an unrelated
change can push one over it, nothing fails, and the launch runs a third slower (round 5: 96.6 -> 132 ms, found by accident).  Runs on
the CPU right after `__graft_entry__.build()`; skipped when the build's remarks are not there."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("occupancy_table", os.path.join(ROOT, "scripts", "occupancy_table.py"))
occ = importlib.util.module_from_spec(spec)
spec.loader.exec_module(occ)


def test_no_kernel_lost_a_wave_per_simd():
    cur = occ.current()
    if len(cur) < 100:
        pytest.skip("no resource-usage remarks beside the sources (the library was not built here)")
    ref = json.load(open(occ.TABLE))
    assert len(ref) > 1000
    lost = [(k, ref[k]["occupancy"], cur[k]["occupancy"], ref[k]["vgprs"], cur[k]["vgprs"]) for k in ref if k in cur and cur[k]["occupancy"] < ref[k]["occupancy"]]
    assert not lost, "kernels that lost a wave per SIMD (name, occupancy before / now, VGPRs before / now): " + "; ".join(map(str, lost[:8]))
    missing = [k for k in ref if k not in cur]
    assert len(missing) < 0.05 * len(ref), f"{len(missing)} kernels of the table are no longer built: regenerate it on purpose (scripts/occupancy_table.py --write)"
    # the other build rule, restated on the remarks: no kernel spills a vector register
    import glob
    import re
    for f in glob.glob(os.path.join(occ.CSRC, "*.remarks")):
        assert not re.search(r"VGPRs Spill: [1-9]", open(f, errors="replace").read()), f"{os.path.basename(f)}: a kernel spills VGPRs"
