#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -k "wide_zigzag or register_and_lds or exact_slot" -x -q > $O/tests_wide.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests_wide.log
[ $rc = 0 ] || exit $rc
: > $O/zz_wide.jsonl
run() { env $1 python3 bench.py --dim $2 --chains $3 --target $4 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'env':'$1','dim':$2,'chains':$3,'target':'$4','ms':round(d['ms_per_step'],3),'frac':round(d['roofline']['frac'],3),'kernel':d['roofline'].get('kernel')}))" >> $O/zz_wide.jsonl; }
for spec in "260 20000" "416 12000" "512 10000" "768 6000"; do
  set -- $spec
  run "PDMP_REG=1" $1 $2 iso
  run "PDMP_REG=0" $1 $2 iso
  run "PDMP_REG=0 PDMP_GROUP_LANES=16" $1 $2 iso
done
run "PDMP_REG=1" 260 20000 banana
run "PDMP_REG=0 PDMP_GROUP_LANES=16" 260 20000 banana
cat $O/zz_wide.jsonl
