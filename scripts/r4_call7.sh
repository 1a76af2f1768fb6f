#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PYTHONUNBUFFERED=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests_full2.log 2>&1; echo "pytest rc=$?"; tail -2 $O/tests_full2.log
for i in 1 2; do timeout -k 10 200 python3 bench.py --gpus 1 --steps 10 --warmup 3 --sampler ZigZagAD --target dense --dim 1000 --chains 10000 --n-sk 200 --cpu-seconds 0 --pmc off > $O/bench_config4_permlane_$i.json 2> $O/bench_config4_permlane_$i.err; echo "config4 rc=$?"; done
timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --sampler BPS --target dense --dim 1000 --chains 10000 --n-sk 200 --cpu-seconds 0 --pmc off > $O/bench_dense_bps_permlane.json 2> $O/bench_dense_bps_permlane.err; echo "dense bps rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/bench_config4_permlane_*.json')+glob.glob('gpurun_out/r4/bench_dense_bps_permlane.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%j['value'], 'ms/step %.2f'%j['ms_per_step'], 'kernel %.2f'%j['roofline']['kernel_ms'])
    except Exception as e: print(f,'ERR',e)
PY
