#!/bin/bash
# where the one-lane form should take over from the pair form (d <= 52): around 32 768 chains = 1 024 pair-form waves, one per SIMD
O=gpurun_out/r4; mkdir -p $O
: > $O/lane_from.jsonl
for d in 16 32 50; do for C in 30000 32768 33500 36000 39000; do for mode in 2 1; do
  env PDMP_LANE=$mode python3 bench.py --dim $d --chains $C --steps 5 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'dim':$d,'chains':$C,'mode':$mode,'ms':round(d['ms_per_step'],3)}))" >> $O/lane_from.jsonl
done; done; done
cat $O/lane_from.jsonl
