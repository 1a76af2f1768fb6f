#!/bin/bash
# Zig-Zag beyond the lane kernels (d > 104): what the group kernel reaches there
O=gpurun_out/r4; mkdir -p $O
: > $O/zz_large_d.jsonl
for spec in "100 50000" "104 50000" "105 50000" "128 40000" "200 25000" "208 25000" "260 20000" "512 10000"; do
  set -- $spec
  python3 bench.py --dim $1 --chains $2 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'dim':$1,'chains':$2,'ms':round(d['ms_per_step'],3),'events_per_s':d['value'],'frac':round(d['roofline']['frac'],3),'kernel':d['roofline'].get('kernel')}))" >> $O/zz_large_d.jsonl
done
cat $O/zz_large_d.jsonl
