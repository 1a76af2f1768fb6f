#!/bin/bash
# gpurun helper: kernel trace of bench.py, prints grid / LDS / VGPR / duration of the group kernels. Usage: scripts/trace_grid.sh <tag> [bench args]
TAG=$1; shift
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --pmc off "$@" > $OUT/bench.json 2> $OUT/log.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "group_kernel" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X"), "lds", r.get("LDS_Block_Size"), "vgpr", r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), "ms", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
PY
