#!/bin/bash
# round 3, first GPU call: the new parity tests + baseline numbers of the workloads this round works on
set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests/test_gpu_tape_replay.py tests/test_gpu_reference_pins.py tests/test_gpu_reference_order.py tests/test_julia_fixtures.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
for c in 12500 25000 50000; do python bench.py --chains $c --pmc off --cpu-seconds 0 --steps 10 --warmup 2 > $O/zz_$c.json 2>$O/zz_$c.err || exit 1; done
python bench.py --sampler BPS --target banana --pmc off --cpu-seconds 0 --steps 5 > $O/bps_banana.json 2>$O/bps.err || exit 1
python bench.py --sampler Boomerang --dim 100 --chains 50000 --refresh-rate 0.5 --pmc off --cpu-seconds 0 --steps 5 > $O/boom.json 2>$O/boom.err || exit 1
python bench.py --pmc off --cpu-seconds 0 --steps 10 > $O/zz_100000.json 2>$O/zz.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3a/*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "%.3e ev/s  %.2f ms  frac %.3f"%(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["frac"]), r["roofline"]["kernel"][:40])
    except Exception as e: print(f, "ERR", e)
PY
