#!/bin/bash
# the chain-count thresholds of the Zig-Zag launch policy (group kernel / pair form / one lane per chain) at another history length:
# n_sk = 200 and 4000 next to the n_sk = 1000 they were tuned on (d = 50 Gaussian)
O=gpurun_out/r4; mkdir -p $O
: > $O/thresholds.jsonl
for nsk in 200 4000; do
for C in 14000 18000 25000 33000 38000 42000 50000; do
  [ $nsk = 4000 ] && [ $C -gt 40000 ] && continue
  for mode in 0 2 1 auto; do
    if [ $mode = auto ]; then E=""; else E="PDMP_LANE=$mode"; fi
    env $E python3 bench.py --chains $C --n-sk $nsk --steps 5 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'n_sk':$nsk,'chains':$C,'mode':'$mode','ms':round(d['ms_per_step'],3)}))" >> $O/thresholds.jsonl
  done
done
done
cat $O/thresholds.jsonl
