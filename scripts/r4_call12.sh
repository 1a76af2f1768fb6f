#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pol in 1 2 3; do
PDMP_DGEMM_POLITE=$pol PDMP_DENSE_LANES=2 PDMP_DENSE_GEMM=mfma timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --sampler BPS --target dense --dim 1000 --chains 10000 --n-sk 200 --cpu-seconds 0 --pmc off > $O/bench_dense_bps_mfma_l2_p$pol.json 2> $O/bench_dense_bps_mfma_l2_p$pol.err
PDMP_DGEMM_POLITE=$pol PDMP_DENSE_LANES=2 PDMP_DENSE_GEMM=mfma timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --sampler BoomerangAD --target dense --dim 1000 --chains 10000 --n-sk 200 --cpu-seconds 0 --pmc off > $O/bench_dense_boom_mfma_l2_p$pol.json 2> $O/bench_dense_boom_mfma_l2_p$pol.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/bench_dense_b*_l2_p?.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%j['value'], 'ms/step %.2f'%j['ms_per_step'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
