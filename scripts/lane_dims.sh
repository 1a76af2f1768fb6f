#!/bin/bash
# lane kernel against group kernel over the dimension (run through gpurun); usage: scripts/lane_dims.sh "27 32 40 48" [bench args]
DIMS=$1; shift
for d in $DIMS; do
  for L in 0 1; do
    r=$(PDMP_LANE=$L timeout -k 10 200 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --pmc off --dim $d "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], round(r['roofline']['frac'],3))")
    echo "d $d lane $L: $r"
  done
done
