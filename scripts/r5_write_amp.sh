#!/bin/bash
# scripts/probes/oob_probe.hip under the write-request counters
mkdir -p gpurun_out/prof_oob
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/prof_oob/p -- scripts/probes/oob_probe > gpurun_out/prof_oob/run.log 2>&1 || { tail -5 gpurun_out/prof_oob/run.log; exit 1; }
python3 - <<'PY'
import glob, csv, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob("gpurun_out/prof_oob/p/*/*_counter_collection.csv"))[-1:]:
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(per.items()):
    a, b = v["TCC_EA0_WRREQ_sum"], v["TCC_EA0_WRREQ_64B_sum"]
    print(k, "WRREQ %.5g (x %.3f of %d lines)  64B %.5g" % (a, a / (1 << 24), 1 << 24, b), flush=True)
PY
