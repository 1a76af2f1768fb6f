#!/bin/bash
# the driver's bench command N times on one box (run through gpurun): ms per step and roofline fraction of each run
N=${1:-3}
mkdir -p gpurun_out/drv
for i in $(seq 1 $N); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 5 > gpurun_out/drv/run_$i.json 2>/dev/null
  python - <<PY
import json
r = json.load(open("gpurun_out/drv/run_$i.json"))
print("run $i:", round(r["ms_per_step"], 2), "ms", round(r["roofline"]["frac"], 3), "%.3e" % r["value"])
PY
done
