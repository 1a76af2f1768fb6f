#!/bin/bash
# the Zig-Zag launch policy's chain-count thresholds (group kernel / pair form / one lane per chain) against the history length (d = 50 Gaussian)
O=gpurun_out/r5; mkdir -p $O
: > $O/thresholds_nsk.jsonl
for nsk in 500 2000 4000; do
for C in 20000 25000 33000 40000 50000 65000; do
  for mode in 0 2 1; do
    env PDMP_LANE=$mode python3 bench.py --chains $C --n-sk $nsk --steps 4 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'n_sk':$nsk,'chains':$C,'mode':'$mode','ms':round(d['ms_per_step'],3)}))" >> $O/thresholds_nsk.jsonl
  done
done
done
python3 - <<'PY'
import json, collections
t=collections.defaultdict(dict)
for ln in open("gpurun_out/r5/thresholds_nsk.jsonl"):
    d=json.loads(ln); t[(d["n_sk"],d["chains"])][d["mode"]]=d["ms"]
for k in sorted(t): print(k, t[k], "best", min(t[k], key=t[k].get))
PY
