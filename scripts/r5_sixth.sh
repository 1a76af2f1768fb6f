#!/bin/bash
# round 5, sixth GPU visit: the whole GPU suite on the library with one run-time-grid instantiation of the LDS form, then the
# benches whose kernels that change touched (LDS forms) and the headline configurations
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/r5_tests8.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r5_tests8.log
if [ $rc -ge 124 ]; then exit $rc; fi
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off --steps 4 --warmup 1"
PDMP_REG=0 $B --sampler ZigZag --target banana > gpurun_out/r5/bench_r5b_zzfd_banana_lds.json 2>/dev/null
PDMP_REG=0 $B --sampler ZigZag > gpurun_out/r5/bench_r5b_zzfd_iso_lds.json 2>/dev/null
PDMP_REG=0 $B > gpurun_out/r5/bench_r5b_config2_lds.json 2>/dev/null
$B > gpurun_out/r5/bench_r5b_config2.json 2>/dev/null
$B --sampler BPS --target banana > gpurun_out/r5/bench_r5b_config3.json 2>/dev/null
$B --sampler Boomerang --dim 100 --sets 1 > gpurun_out/r5/bench_r5b_config5.json 2>/dev/null
$B --target dense --dim 1000 --chains 10000 --n-sk 200 > gpurun_out/r5/bench_r5b_config4.json 2>/dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/bench_r5b_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms  frac %.3f" % d["roofline"]["frac"], d["roofline"]["kernel"][:60])
    except Exception as e:
        print(f, "failed", e)
PY
