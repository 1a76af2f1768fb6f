#!/bin/bash
# bound instalments per pass of the lane forms over the dimension (the default of 5 was tuned at d = 50)
O=gpurun_out/r4; mkdir -p $O
: > $O/lane_budget_by_dim.jsonl
run() { env PDMP_LANE=$1 PDMP_CELL_BUDGET=$2 python3 bench.py --dim $3 --chains $4 --steps 4 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'dim':$3,'chains':$4,'mode':$1,'budget':$2,'ms':round(d['ms_per_step'],3)}))" >> $O/lane_budget_by_dim.jsonl; }
for d in 10 16 24 32 40; do for C in 50000 100000; do for b in 5 7 9; do run 1 $b $d $C; done; done; done
for d in 2 5 8 10 16 24; do for b in 5 9; do run 2 $b $d 25000; done; done
cat $O/lane_budget_by_dim.jsonl
