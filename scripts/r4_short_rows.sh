#!/bin/bash
# lane kernels of 1 and 2 slots (d = 1 .. 8, rows shorter than a 64-byte line): parity, a random sweep, then the forms against each other
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_lane_kernel.py tests/test_gpu_lane_pair_kernel.py -x -q > $O/tests_short_rows.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/tests_short_rows.log
[ $rc = 0 ] || exit $rc
timeout -k 10 600 python3 scripts/lane_random_sweep.py 300 41 1 8 1 > $O/sweep_short_lane.log 2>&1; echo "sweep lane rc=$?"; tail -2 $O/sweep_short_lane.log
timeout -k 10 600 python3 scripts/lane_random_sweep.py 200 42 1 8 2 > $O/sweep_short_pair.log 2>&1; echo "sweep pair rc=$?"; tail -2 $O/sweep_short_pair.log
: > $O/short_rows.jsonl
for d in 2 5 8; do for C in 12500 25000 50000 100000; do for mode in 0 2 1; do
    env PDMP_LANE=$mode python3 bench.py --dim $d --chains $C --steps 4 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'dim':$d,'chains':$C,'mode':'$mode','ms':round(d['ms_per_step'],3)}))" >> $O/short_rows.jsonl
done; done; done
for d in 2 5 8; do for C in 12500 25000 50000 100000; do PDMP_GROUP_LANES=1 python3 bench.py --dim $d --chains $C --steps 4 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(json.dumps({'dim':$d,'chains':$C,'mode':'G=1 group (old default)','ms':round(d['ms_per_step'],3)}))" >> $O/short_rows.jsonl; done; done
cat $O/short_rows.jsonl
