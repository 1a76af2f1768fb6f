#!/bin/bash
# the driver's command in N fresh processes, one after the other: is the headline reproducible?
O=gpurun_out/r4; mkdir -p $O
N=${1:-6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 $N); do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_repeat_$i.json 2> $O/bench_repeat_$i.err; echo "run $i rc=$?"
  python3 -c "
import json,sys
j=json.loads(open('$O/bench_repeat_$i.json').read().strip().splitlines()[-1])
print('  value %.4e  ms/step %.2f  placement' % (j['value'], j['ms_per_step']), [(p['allocator'][:4], round(sum(p['kernel_ms'])/len(p['kernel_ms']),2)) for p in j['config']['placement_ms']])"
done
