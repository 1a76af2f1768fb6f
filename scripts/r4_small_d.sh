#!/bin/bash
# lane kernels with 4 .. 6 slots (d = 13 .. 24): parity, then the three forms against each other over the chain count
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_lane_kernel.py tests/test_gpu_lane_pair_kernel.py -x -q > $O/tests_small_d.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests_small_d.log
: > $O/small_d.jsonl
for d in 16 24; do for C in 12500 20000 30000 40000 60000 100000; do for mode in 0 2 1 auto; do
    if [ $mode = auto ]; then E=""; else E="PDMP_LANE=$mode"; fi
    env $E python3 bench.py --dim $d --chains $C --steps 4 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'dim':$d,'chains':$C,'mode':'$mode','ms':round(d['ms_per_step'],3)}))" >> $O/small_d.jsonl
done; done; done
for d in 16 24; do PDMP_GROUP_LANES=2 python3 bench.py --dim $d --steps 4 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(json.dumps({'dim':$d,'chains':100000,'mode':'G=2 group (old default)','ms':round(d['ms_per_step'],3)}))" >> $O/small_d.jsonl; done
cat $O/small_d.jsonl
