#!/bin/bash
# usage: scripts/r3_pair.sh <tag>  -- the lane-pair kernel (d = 53 .. 104): its tests, then A/B lines against the G = 8 group kernel
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_gpu_lane_pair_kernel.py tests/test_gpu_lane_kernel.py -x -q > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
B="--pmc off --cpu-seconds 0 --steps 6 --warmup 2"
for L in 0 1; do
  PDMP_LANE=$L python bench.py --dim 100 $B > $O/zz100_lane$L.json 2>$O/zz100_lane$L.err || exit 1
  PDMP_LANE=$L python bench.py --dim 64 $B > $O/zz64_lane$L.json 2>$O/zz64_lane$L.err || exit 1
  PDMP_LANE=$L python bench.py --dim 100 --target banana $B > $O/zzb100_lane$L.json 2>$O/zzb100_lane$L.err || exit 1
  PDMP_LANE=$L python bench.py --dim 100 --chains 25000 $B > $O/zz100_25k_lane$L.json 2>$O/zz100_25k_lane$L.err || exit 1
done
python bench.py $B --steps 10 > $O/zz50.json 2>$O/zz50.err || exit 1
python bench.py $B --steps 10 --target banana > $O/zzb50.json 2>$O/zzb50.err || exit 1
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "%.3e ev/s  %.2f ms  frac %.3f"%(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["frac"]), r["roofline"]["kernel"][:48])
    except Exception as e: print(f, "ERR", e)
PY
