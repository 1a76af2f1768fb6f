#!/bin/bash
# In-box sweep of the lane kernel's tuning knobs (run through gpurun): ms per step of bench.py's default workload.
# usage: scripts/lane_sweep.sh "<budgets>" "<rounds>" [bench args]
B=${1:-"3 5"}; R=${2:-"8 16"}; shift; shift
BARGS=("$@")                        # the bench arguments of every run of the sweep
run() { r=$(env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --pmc off "${BARGS[@]}" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"); echo "$* $r"; }
run PDMP_LANE=0
for b in $B; do for r in $R; do run PDMP_CELL_BUDGET=$b PDMP_LANE_ROUNDS=$r; done; done
run PDMP_LANE=0
