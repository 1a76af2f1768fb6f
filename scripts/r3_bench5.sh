#!/bin/bash
# usage: scripts/r3_bench5.sh <tag>  -- the five workloads this round tracks, one line each (no PMC children, no CPU baseline)
O=gpurun_out/$1; mkdir -p $O
for c in 12500 25000 100000; do python bench.py --chains $c --pmc off --cpu-seconds 0 --steps 10 --warmup 2 > $O/zz_$c.json 2>$O/zz_$c.err || exit 1; done
python bench.py --target banana --pmc off --cpu-seconds 0 --steps 10 --warmup 2 > $O/zz_banana.json 2>$O/zzb.err || exit 1
python bench.py --sampler BPS --target banana --pmc off --cpu-seconds 0 --steps 5 > $O/bps_banana.json 2>$O/bps.err || exit 1
python bench.py --sampler Boomerang --dim 100 --chains 50000 --refresh-rate 0.5 --pmc off --cpu-seconds 0 --steps 5 > $O/boom.json 2>$O/boom.err || exit 1
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "%.3e ev/s  %.2f ms  frac %.3f"%(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["frac"]), r["roofline"]["kernel"][:40])
    except Exception as e: print(f, "ERR", e)
PY
