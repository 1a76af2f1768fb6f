#!/bin/bash
# dense BPS / Boomerang pipeline: the step kernel with fewer workgroups (PDMP_DENSE_STEP_BLOCKS) beside the other lane's product
O=gpurun_out/r4; mkdir -p $O
B="python3 bench.py --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off"
for sb in 0 128 256 384 512 768; do
  for smp in BPS BoomerangAD; do
    PDMP_DENSE_STEP_BLOCKS=$sb $B --sampler $smp 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$smp step_blocks=$sb', round(d['ms_per_step'],2))"
  done
done
for pol in 1 3 4; do PDMP_DGEMM_POLITE=$pol PDMP_DENSE_GEMM=mfma $B --sampler BPS 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('BPS mfma polite=$pol', round(d['ms_per_step'],2))"; done
