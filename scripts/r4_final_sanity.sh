#!/bin/bash
# the BASELINE configurations once more on the round's final library (no regressions from the late additions)
O=gpurun_out/r4; mkdir -p $O
: > $O/final_sanity.jsonl
run() { name=$1; shift; python3 bench.py "$@" --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'case':'$name','ms':round(d['ms_per_step'],3),'events_per_s':d['value'],'frac':round(d['roofline']['frac'],3),'kernel':d['roofline'].get('kernel')}))" >> $O/final_sanity.jsonl; }
run "config 2 banana" --target banana
run "config 3 BPS banana" --sampler BPS --target banana
run "config 5 Boomerang d=100 shard 12500" --sampler Boomerang --dim 100 --chains 12500
run "config 5 Boomerang d=100 1e5" --sampler Boomerang --dim 100
run "config 4 dense Zig-Zag" --sampler ZigZagAD --target dense --dim 1000 --chains 10000 --n-sk 200
run "shard 12500" --chains 12500
run "shard 25000" --chains 25000
run "shard 50000" --chains 50000
run "multi driver" --driver multi
run "RHMC" --sampler RHMC --n-sk 200
run "ForwardECMC" --sampler ForwardECMC --n-sk 200
cat $O/final_sanity.jsonl
