#!/bin/bash
# the driver's command in ten fresh processes on one box (the drawing allocator as shipped)
O=gpurun_out/r4; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_ten_$i.json 2> $O/bench_ten_$i.err || { echo "run $i failed"; tail -3 $O/bench_ten_$i.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('$O/bench_ten_$i.json').read().strip().splitlines()[-1]);print($i,'%.4g'%d['value'],round(d['ms_per_step'],3),[(q['allocator'][:9],round(sum(q['kernel_ms'])/len(q['kernel_ms']),2)) for q in d['config']['placement_ms']])"
done
