#!/bin/bash
# lane kernels after a change of their store path: parity (lane tests, golden sets, tape, time horizon), then config 2 and neighbours
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_lane_kernel.py tests/test_golden.py tests/test_gpu_tape_replay.py tests/test_gpu_time_horizon.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r5_line128_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/r5_line128_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for a in "--dim 50" "--dim 48" "--dim 32" "--dim 50 --target banana" "--dim 50 --chains 50000" "--dim 100"; do
  python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off --steps 6 --warmup 2 $a > gpurun_out/r5/bench_tmp.json 2>/dev/null || exit 1
  python3 - "$a" <<'PY'
import json, sys
r = json.loads(open("gpurun_out/r5/bench_tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms %.3f" % r["ms_per_step"], "frac %.3f" % r["roofline"]["frac"], r["roofline"]["kernel"][:40], flush=True)
PY
done
