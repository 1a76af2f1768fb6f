#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_dgemm_mfma.py tests/test_gpu_bench_sharded.py -x -q -p no:cacheprovider > $O/tests_dgemm.log 2>&1; echo "pytest rc=$?"; tail -6 $O/tests_dgemm.log
timeout -k 10 200 python3 scripts/bench_dgemm.py > $O/bench_dgemm.jsonl 2>&1; cat $O/bench_dgemm.jsonl
for impl in rocblas mfma; do
PDMP_DENSE_LANES=2 PDMP_DENSE_GEMM=$impl timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --sampler BPS --target dense --dim 1000 --chains 10000 --n-sk 200 --cpu-seconds 0 --pmc off > $O/bench_dense_bps_${impl}_l2.json 2> $O/bench_dense_bps_${impl}_l2.err
PDMP_DENSE_GEMM=$impl timeout -k 10 200 python3 bench.py --gpus 1 --steps 10 --warmup 3 --sampler ZigZagAD --target dense --dim 1000 --chains 10000 --n-sk 200 --cpu-seconds 0 --pmc off > $O/bench_config4_${impl}.json 2> $O/bench_config4_${impl}.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/bench_dense_bps_*_l2.json')+glob.glob('gpurun_out/r4/bench_config4_rocblas.json')+glob.glob('gpurun_out/r4/bench_config4_mfma.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%j['value'], 'ms/step %.2f'%j['ms_per_step'], 'kernel %.2f'%j['roofline']['kernel_ms'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
