#!/bin/bash
# is the lane kernel's super-linear growth with the history length a matter of the chain stride (n_sk x d x 8 bytes) modulo the
# memory's channel interleave?  38 000 chains, one lane per chain forced, history lengths around 4000 / 1000 and d = 50 / 51
O=gpurun_out/r4; mkdir -p $O
: > $O/stride.jsonl
for spec in "4000 50" "3999 50" "4001 50" "4003 50" "4096 50" "4000 51" "1000 50" "1001 50" "1024 50" "2000 50" "2001 50" "3000 50" "3001 50"; do
  set -- $spec
  PDMP_LANE=1 python3 bench.py --chains 38000 --n-sk $1 --dim $2 --steps 4 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'n_sk':$1,'dim':$2,'stride_bytes':$1*$2*8,'ms':round(d['ms_per_step'],3),'us_per_event_column':round(d['ms_per_step']*1000/$1,3)}))" >> $O/stride.jsonl
done
cat $O/stride.jsonl
