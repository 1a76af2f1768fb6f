#!/bin/bash
# time horizon inside the batched dense kernels: the time-horizon suites, then the dense bench lines again (registers of the step kernels moved)
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_time_horizon_chunks.py tests/test_gpu_time_horizon.py tests/test_gpu_parity.py tests/test_gpu_reference_order.py -k "dense or horizon or chunk" -x -q -s > $O/tests_dense_until.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/tests_dense_until.log
[ $rc = 0 ] || exit $rc
for cfg in "--sampler ZigZagAD" "--sampler BPS" "--sampler BoomerangAD"; do
  python3 bench.py $cfg --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['ms_per_step'],2), 'ms')"
done
