#!/bin/bash
# the metric's shards (2 / 4 / 8 GPUs: 50 000 / 25 000 / 12 500 chains) and the other BASELINE configs on the round's final library
mkdir -p gpurun_out/r5
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off --steps 10 --warmup 2"
for c in 50000 25000 12500; do $B --chains $c > gpurun_out/r5/bench_r5f_shard_$c.json 2>/dev/null || exit 1; done
$B --target banana > gpurun_out/r5/bench_r5f_zz_banana.json 2>/dev/null || exit 1
$B --sampler BPS --target banana --steps 4 > gpurun_out/r5/bench_r5f_config3.json 2>/dev/null || exit 1
$B --sampler Boomerang --dim 100 --sets 1 --steps 3 > gpurun_out/r5/bench_r5f_config5.json 2>/dev/null || exit 1
$B --dim 100 --steps 4 > gpurun_out/r5/bench_r5f_zz_d100.json 2>/dev/null || exit 1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/bench_r5f_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms  frac %.3f" % d["roofline"]["frac"], d["roofline"]["kernel"][:60])
PY
