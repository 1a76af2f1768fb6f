#!/bin/bash
# the shipped default choice (no environment) at small dimensions, both targets
O=gpurun_out/r4; mkdir -p $O
: > $O/short_rows_default.jsonl
for tgt in iso banana; do for d in 2 5 8 10 16; do for C in 12500 25000 50000 100000; do
    python3 bench.py --dim $d --chains $C --target $tgt --steps 4 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'target':'$tgt','dim':$d,'chains':$C,'ms':round(d['ms_per_step'],3),'events_per_s':d['value'],'kernel':d['roofline'].get('kernel'),'frac':d['roofline'].get('frac')}))" >> $O/short_rows_default.jsonl
done; done; done
cat $O/short_rows_default.jsonl
