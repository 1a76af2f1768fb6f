#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PYTHONUNBUFFERED=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests_full3.log 2>&1; echo "pytest rc=$?"; tail -4 $O/tests_full3.log
for reg in 0 1; do
PDMP_REG=$reg timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --sampler RHMC --dim 50 --chains 100000 --n-sk 200 --cpu-seconds 0 --pmc off --sets 1 --placement-survey off > $O/bench_rhmc_reg$reg.json 2> $O/bench_rhmc_reg$reg.err
PDMP_REG=$reg timeout -k 10 200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --sampler RHMC --target banana --dim 50 --chains 100000 --n-sk 200 --cpu-seconds 0 --pmc off --sets 1 --placement-survey off > $O/bench_rhmc_banana_reg$reg.json 2> $O/bench_rhmc_banana_reg$reg.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/bench_rhmc*_reg?.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%j['value'], 'ms/step %.2f'%j['ms_per_step'], j['roofline']['kernel'][:60])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
