#!/bin/bash
# config 4: di_segment_kernel with streaming and with plain row stores, and without its stores / loads of P, on the same arrays (bench.py --ab-env)
mkdir -p gpurun_out/r5
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off --steps 4 --warmup 2 --target dense --dim 1000 --chains 10000 --n-sk 200"
$B --ab-env PDMP_DENSE_DEBUG=0,4,1,2,3 > gpurun_out/r5/bench_r5c_config4_ab_general.json 2>/dev/null || exit 1
$B --dim 500 --ab-env PDMP_DENSE_DEBUG=0,4,1 > gpurun_out/r5/bench_r5c_d500_ab_general.json 2>/dev/null || exit 1
python3 - <<'PY'
import json, glob
for f in ("gpurun_out/r5/bench_r5c_config4_ab_general.json", "gpurun_out/r5/bench_r5c_d500_ab_general.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"], 3), d["ab_env"]["knob"], d["ab_env"]["mean_ms"])
PY
