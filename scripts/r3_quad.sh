#!/bin/bash
# usage: scripts/r3_quad.sh <tag>  -- the pair form of the lane kernel at d <= 52: tests, then mid-sized launches A/B against the other forms
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_gpu_lane_pair_kernel.py tests/test_gpu_lane_kernel.py tests/test_gpu_time_horizon_chunks.py -x -q > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
B="--pmc off --cpu-seconds 0 --steps 8 --warmup 2"
for C in 12500 18000 25000 32000 40000 50000; do
  for L in 0 1 2; do
    PDMP_LANE=$L python bench.py --chains $C $B > $O/zz_${C}_lane$L.json 2>$O/zz_${C}_lane$L.err || exit 1
  done
done
for L in 0 2; do PDMP_LANE=$L python bench.py --chains 25000 --target banana $B > $O/zzb_25000_lane$L.json 2>$O/zzb_25000_lane$L.err || exit 1; done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "%.3e ev/s  %.2f ms  frac %.3f"%(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["frac"]), r["roofline"]["kernel"][:48])
    except Exception as e: print(f, "ERR", e)
PY
