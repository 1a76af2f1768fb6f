#!/bin/bash
# 128-byte lines in the one-lane kernels: parity first (lane tests + golden), then config 2 with live write traffic
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_lane_kernel.py tests/test_golden.py tests/test_gpu_tape_replay.py tests/test_gpu_time_horizon.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r5_line128_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r5_line128_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for d in 50 32 48 26; do
  python3 bench.py --cpu-seconds 0 --placement-survey off --steps 4 --warmup 1 --dim $d > gpurun_out/r5/bench_r5_line128_d$d.json 2>/dev/null || exit 1
  python3 - $d <<'PY'
import json, sys
d = int(sys.argv[1])
r = json.loads(open(f"gpurun_out/r5/bench_r5_line128_d{d}.json").read().strip().splitlines()[-1])
ev = 100000 * 999
w = r["roofline"]["traffic"] - r["roofline"]["traffic_read"]
print("d", d, "ms %.3f" % r["ms_per_step"], "frac %.3f" % r["roofline"]["frac"], "written %.5g = %.3f 64-byte lines per event (events: %.3f)" % (w, w / 64 / ev, (16 * d + 8) / 64), flush=True)
PY
done
