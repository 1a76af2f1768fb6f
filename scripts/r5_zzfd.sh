#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_lane_kernel.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "ZigZag or zigzag or golden or random or lean" > gpurun_out/r5_tests5.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r5_tests5.log
if [ $rc -ge 124 ]; then exit $rc; fi
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off"
$B --sampler ZigZag --steps 3 --warmup 1 > gpurun_out/r5/bench_r5_zigzag_fd_reg.json 2>/dev/null && \
PDMP_REG=0 $B --sampler ZigZag --steps 3 --warmup 1 > gpurun_out/r5/bench_r5_zigzag_fd_lds.json 2>/dev/null && \
$B --sampler ZigZag --chains 12500 --steps 3 --warmup 1 > gpurun_out/r5/bench_r5_zigzag_fd_reg_12500.json 2>/dev/null && \
PDMP_REG=0 $B --sampler ZigZag --chains 12500 --steps 3 --warmup 1 > gpurun_out/r5/bench_r5_zigzag_fd_lds_12500.json 2>/dev/null && \
$B --sampler ZigZag --target banana --steps 3 --warmup 1 > gpurun_out/r5/bench_r5_zigzag_fd_banana.json 2>/dev/null && \
$B --sampler ZigZag --target diag --steps 3 --warmup 1 > gpurun_out/r5/bench_r5_zigzag_fd_diag.json 2>/dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/bench_r5_zigzag_fd_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms  %.3e ev/s  frac %.3f  %s" % (d["value"], d["roofline"]["frac"], d["roofline"]["kernel"][:50]))
    except Exception as e:
        print(f, "ERR", e)
PY
