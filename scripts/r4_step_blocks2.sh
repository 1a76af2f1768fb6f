#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
B="python3 bench.py --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off --sampler BoomerangAD"
for rep in 1 2 3; do for sb in 0 256 192 320; do
    PDMP_DENSE_STEP_BLOCKS=$sb $B 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('Boomerang rep $rep step_blocks=$sb', round(d['ms_per_step'],2))"
done; done
