#!/bin/bash
mkdir -p gpurun_out/r5
for d in 1000 1008 992; do
python3 bench.py --cpu-seconds 0 --placement-survey off --steps 6 --warmup 2 --target dense --dim $d --chains 10000 --n-sk 200 --pmc off --ab-env PDMP_DENSE_DEBUG=0,1 > gpurun_out/r5/bench_tmp.json 2>/dev/null || exit 1
python3 - $d <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r5/bench_tmp.json").read().strip().splitlines()[-1])
print("d", sys.argv[1], round(d["ms_per_step"], 3), "frac %.3f" % d["roofline"]["frac"], d["ab_env"]["mean_ms"], flush=True)
PY
done
