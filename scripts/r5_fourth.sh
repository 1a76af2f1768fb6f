#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/r5_tests6.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r5_tests6.log
if [ $rc -ge 124 ]; then exit $rc; fi
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off --steps 3 --warmup 1"
PDMP_REG=0 $B --sampler ZigZag --target banana > gpurun_out/r5/bench_r5_zigzag_fd_banana_lds.json 2>/dev/null
$B --sampler ZigZag > gpurun_out/r5/bench_r5_zigzag_fd_reg.json 2>/dev/null
$B --sampler ZigZag --chains 12500 > gpurun_out/r5/bench_r5_zigzag_fd_reg_12500.json 2>/dev/null
for d in 2 8 10 16 20 26 32; do $B --dim $d > gpurun_out/r5/bench_r5_zz_d$d.json 2>/dev/null; done
$B --dim 10 --target banana > gpurun_out/r5/bench_r5_zz_d10_banana.json 2>/dev/null
$B --steps 10 --warmup 3 > gpurun_out/r5/bench_r5_config2.json 2>/dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/bench_r5_zz_d*.json"))+["gpurun_out/r5/bench_r5_zigzag_fd_banana_lds.json","gpurun_out/r5/bench_r5_zigzag_fd_reg.json","gpurun_out/r5/bench_r5_zigzag_fd_reg_12500.json","gpurun_out/r5/bench_r5_config2.json"]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"][:60])
    except Exception as e:
        print(f, "ERR", e)
PY
