#!/bin/bash
# lane kernel against group kernel over the ensemble size (run through gpurun); usage: scripts/lane_vs_group.sh "<chain counts>" [bench args]
CS=${1:-"100000 75000 50000 37500 25000 12500"}; shift
for C in $CS; do
  for L in 0 1; do
    r=$(PDMP_LANE=$L timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --pmc off --chains $C "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])")
    echo "chains $C lane $L: $r ms"
  done
done
