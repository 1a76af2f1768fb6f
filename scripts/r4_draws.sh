#!/bin/bash
# the drawing allocator: unit tests, then the driver's command in fresh processes (config.placement_ms[*].alloc has the draws)
O=gpurun_out/r4; mkdir -p $O
N=${1:-4}
timeout -k 10 300 python3 -m pytest tests/test_gpu_device_alloc.py -x -q > $O/tests_alloc.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests_alloc.log
for i in $(seq 1 $N); do
  timeout -k 10 280 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --pmc off > $O/bench_draws_$i.json 2> $O/bench_draws_$i.err || { echo "bench $i failed"; tail -5 $O/bench_draws_$i.err; exit 1; }
  python3 - $O/bench_draws_$i.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value %.4g ms %.3f'%(d['value'],d['ms_per_step']), [(q['allocator'], round(sum(q['kernel_ms'])/len(q['kernel_ms']),2), q.get('alloc',{}).get('draw_ms'), q.get('alloc',{}).get('seconds')) for q in d['config']['placement_ms']])
P
done
