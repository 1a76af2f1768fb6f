#!/bin/bash
# config 4 on one allocation: di_segment_kernel as shipped, without its row stores, without its loads of P's columns, without both
# (PDMP_DENSE_DEBUG, a knob of the launch policy; bench.py --ab-env reloads the handle's policy between the values)
mkdir -p gpurun_out/r5
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off --steps 4 --warmup 2 --target dense --dim 1000 --chains 10000 --n-sk 200"
$B --ab-env PDMP_DENSE_DEBUG=0,1,2,3 > gpurun_out/r5/bench_r5_config4_ab.json 2>/dev/null || exit 1
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r5/bench_r5_config4_ab.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 3), d["ab_env"]["knob"], d["ab_env"]["mean_ms"])
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dense" -p no:cacheprovider 2>&1 | tail -3
