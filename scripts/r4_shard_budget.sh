#!/bin/bash
# bound instalments per pass for the lane forms when the launch leaves every wave alone on its SIMD (the metric's 2- and 4-GPU shards)
O=gpurun_out/r4; mkdir -p $O
: > $O/shard_budget.jsonl
for tgt in iso banana; do for C in 25000 33000 50000 65000; do for b in 5 7 9; do
  env PDMP_CELL_BUDGET=$b python3 bench.py --chains $C --target $tgt --steps 5 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'target':'$tgt','chains':$C,'budget':$b,'ms':round(d['ms_per_step'],3),'kernel':d['roofline'].get('kernel')[:40]}))" >> $O/shard_budget.jsonl
done; done; done
cat $O/shard_budget.jsonl
