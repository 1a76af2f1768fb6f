#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
: > $O/zzfd.jsonl
run() { env $1 python3 bench.py --sampler ZigZag --dim $2 --target $3 --chains $4 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'env':'$1','dim':$2,'target':'$3','chains':$4,'ms':round(d['ms_per_step'],3),'kernel':d['roofline'].get('kernel')}))" >> $O/zzfd.jsonl; }
for tgt in iso banana; do
  run "PDMP_GROUP_LANES=8" 50 $tgt 100000
  run "PDMP_GROUP_LANES=8 PDMP_REG=0" 50 $tgt 100000
  run "PDMP_REG=0" 50 $tgt 100000
done
run "PDMP_GROUP_LANES=8" 50 iso 12500
run "PDMP_REG=0" 50 iso 12500
cat $O/zzfd.jsonl
