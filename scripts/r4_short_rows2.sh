#!/bin/bash
# short rows, second pass: the pair form's sweep again (d = 1 hand-off words), bound instalments per pass for the one-lane form
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python3 scripts/lane_random_sweep.py 400 42 1 8 2 > $O/sweep_short_pair.log 2>&1; echo "sweep pair rc=$?"; tail -2 $O/sweep_short_pair.log
timeout -k 10 600 python3 scripts/lane_random_sweep.py 300 43 1 4 1 > $O/sweep_short_lane2.log 2>&1; echo "sweep lane rc=$?"; tail -2 $O/sweep_short_lane2.log
: > $O/short_rows_budget.jsonl
for d in 2 5 8; do for C in 50000 100000; do for b in 3 5 9; do for rounds in 24 8; do
    env PDMP_LANE=1 PDMP_CELL_BUDGET=$b PDMP_LANE_ROUNDS=$rounds python3 bench.py --dim $d --chains $C --steps 4 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'dim':$d,'chains':$C,'budget':$b,'rounds':$rounds,'ms':round(d['ms_per_step'],3)}))" >> $O/short_rows_budget.jsonl
done; done; done; done
cat $O/short_rows_budget.jsonl
