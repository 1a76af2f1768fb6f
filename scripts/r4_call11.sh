#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for impl in rocblas mfma; do for lanes in 1 2; do
PDMP_DENSE_LANES=$lanes PDMP_DENSE_GEMM=$impl timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --sampler BPS --target dense --dim 1000 --chains 10000 --n-sk 200 --cpu-seconds 0 --pmc off > $O/bench_dense_bps_${impl}_l$lanes.json 2> $O/bench_dense_bps_${impl}_l$lanes.err
PDMP_DENSE_LANES=$lanes PDMP_DENSE_GEMM=$impl timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --sampler BoomerangAD --target dense --dim 1000 --chains 10000 --n-sk 200 --cpu-seconds 0 --pmc off > $O/bench_dense_boom_${impl}_l$lanes.json 2> $O/bench_dense_boom_${impl}_l$lanes.err
done
PDMP_DENSE_GEMM=$impl timeout -k 10 200 python3 bench.py --gpus 1 --steps 10 --warmup 3 --sampler ZigZagAD --target dense --dim 1000 --chains 10000 --n-sk 200 --cpu-seconds 0 --pmc off > $O/bench_config4_${impl}.json 2> $O/bench_config4_${impl}.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/bench_dense_b*_l?.json')+glob.glob('gpurun_out/r4/bench_config4_rocblas.json')+glob.glob('gpurun_out/r4/bench_config4_mfma.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%j['value'], 'ms/step %.2f'%j['ms_per_step'], 'kernel %.2f'%j['roofline']['kernel_ms'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
