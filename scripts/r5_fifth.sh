#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/r5_tests7.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r5_tests7.log
if [ $rc -ge 124 ]; then exit $rc; fi
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off --steps 4 --warmup 1"
$B --target diag > gpurun_out/r5/bench_r5_zz_diag.json 2>/dev/null
$B --target diag --dim 100 > gpurun_out/r5/bench_r5_zz_diag_d100.json 2>/dev/null
$B --target diag --dim 20 > gpurun_out/r5/bench_r5_zz_diag_d20.json 2>/dev/null
PDMP_REG=0 $B --target diag > gpurun_out/r5/bench_r5_zz_diag_lds.json 2>/dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/bench_r5_zz_diag*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms  frac %.3f" % d["roofline"]["frac"], d["roofline"]["kernel"][:60])
PY
