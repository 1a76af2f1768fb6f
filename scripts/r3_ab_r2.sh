#!/bin/bash
# A/B on ONE box: the round-2 tree (scratch/r2tree, built from ab59eec) against the working tree, alternating.  usage: r3_ab_r2.sh [reps] [workload args ...;...]
ms() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % r['roofline']['kernel_ms'])"; }
REPS=${1:-2}
for rep in $(seq $REPS); do
for args in "--target banana" "" "--sampler BPS --target banana"; do
  a=$(cd scratch/r2tree && python bench.py $args --pmc off --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | ms)
  b=$(python bench.py $args --pmc off --cpu-seconds 0 --steps 6 --warmup 2 2>/dev/null | ms)
  echo "[$args] r2 $a ms   now $b ms"
done; done
