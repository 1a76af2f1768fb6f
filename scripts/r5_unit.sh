#!/bin/bash
# round 5: di_unit_kernel (config 4 with v as sign bits): the dense tests, then config 4 with the unit form and without it, and what
# the launch costs without its row stores / its loads of P's columns (PDMP_DENSE_DEBUG: measurements only, wrong results)
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_time_horizon.py tests/test_gpu_time_horizon_chunks.py tests/test_gpu_dense_threads.py -m gpu -q -k "dense or config4" --maxfail=10 -p no:cacheprovider > gpurun_out/r5_unit_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r5_unit_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off --steps 6 --warmup 2 --target dense --dim 1000 --chains 10000 --n-sk 200"
$B > gpurun_out/r5/bench_r5c_config4_unit.json 2>gpurun_out/r5/bench_r5c_config4_unit.err
PDMP_DENSE_UNIT=0 $B > gpurun_out/r5/bench_r5c_config4_general.json 2>/dev/null
for m in 1 2 3; do PDMP_DENSE_DEBUG=$m $B > gpurun_out/r5/bench_r5c_config4_unit_debug$m.json 2>/dev/null; done
$B --dim 500 > gpurun_out/r5/bench_r5c_dense_d500_unit.json 2>/dev/null
PDMP_DENSE_UNIT=0 $B --dim 500 > gpurun_out/r5/bench_r5c_dense_d500_general.json 2>/dev/null
$B --dim 250 --chains 40000 > gpurun_out/r5/bench_r5c_dense_d250_unit.json 2>/dev/null
PDMP_DENSE_UNIT=0 $B --dim 250 --chains 40000 > gpurun_out/r5/bench_r5c_dense_d250_general.json 2>/dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/bench_r5c_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms  frac %.3f" % d["roofline"]["frac"], d["roofline"]["kernel"][:70])
    except Exception as e:
        print(f, "failed", e)
PY
