#!/usr/bin/env python3
"""Occupancy (waves per SIMD, as the register counts allow) of every kernel of libpdmphip.so, from the compiler's resource-usage remarks the
build keeps (pdmpflux.jl_amd/csrc/*.remarks).  `python3 scripts/occupancy_table.py --write` freezes the table in profiles/occupancy.json;
tests/test_occupancy_guard.py holds every later build to it: a kernel that loses a wave per SIMD (a register-count threshold crossed by a
few registers: 128 / 168 / 256) fails the suite instead of shipping 36 % slower (round 5: Zig-Zag with finite-difference slopes on the banana,
96.6 -> 132 ms, found by accident)."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pdmpflux.jl_amd", "csrc")
TABLE = os.path.join(ROOT, "profiles", "occupancy.json")


def current():
    out = {}
    for f in sorted(glob.glob(os.path.join(CSRC, "*.remarks"))):
        name = None
        for ln in open(f, errors="replace"):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                name = m.group(1)
                continue
            m = re.search(r"(VGPRs|AGPRs|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)", ln)
            if m and name:
                out.setdefault(name, {})[{"VGPRs": "vgprs", "AGPRs": "agprs", "Occupancy [waves/SIMD]": "occupancy", "ScratchSize [bytes/lane]": "scratch"}[m.group(1)]] = int(m.group(2))
    return {k: v for k, v in out.items() if "occupancy" in v}


if __name__ == "__main__":
    cur = current()
    if "--write" in sys.argv:
        json.dump({k: cur[k] for k in sorted(cur)}, open(TABLE, "w"), indent=0, sort_keys=True)
        print(f"{len(cur)} kernels written to {TABLE}")
    else:
        ref = json.load(open(TABLE))
        lost = [(k, ref[k]["occupancy"], cur[k]["occupancy"]) for k in ref if k in cur and cur[k]["occupancy"] < ref[k]["occupancy"]]
        for k, a, b in lost:
            print(f"LOST  {a} -> {b}  {k}")
        print(f"{len(cur)} kernels, {len(lost)} lost a wave per SIMD, {sum(1 for k in ref if k not in cur)} of the table not built, {sum(1 for k in cur if k not in ref)} new")
        sys.exit(1 if lost else 0)
