#!/bin/bash
# SpeedUpZigZag in the register form: parity, then against the LDS form (old default layout: G = 4 at d = 50)
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_suzz.py tests/test_gpu_time_horizon_chunks.py tests/test_gpu_time_horizon.py tests/test_gpu_parity.py -k "suzz or SpeedUp or speed_up" -x -q > $O/tests_suzz.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/tests_suzz.log
[ $rc = 0 ] || exit $rc
: > $O/suzz_forms.jsonl
run() { env $1 python3 bench.py --sampler SpeedUpZigZagAD --dim $2 --target $3 --chains $4 --n-sk 500 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off --alloc hipmalloc 2>$O/suzz_err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'env':'$1','dim':$2,'target':'$3','chains':$4,'ms':round(d['ms_per_step'],3),'kernel':d['roofline'].get('kernel')}))" >> $O/suzz_forms.jsonl; }
for tgt in iso diag banana; do
  run "PDMP_REG=1" 50 $tgt 100000
  run "PDMP_REG=0 PDMP_GROUP_LANES=4" 50 $tgt 100000
done
run "PDMP_REG=1 PDMP_GROUP_LANES=16" 50 iso 100000
run "PDMP_REG=1" 10 iso 100000
run "PDMP_REG=0 PDMP_GROUP_LANES=1" 10 iso 100000
run "PDMP_REG=1" 10 banana 100000
run "PDMP_REG=0 PDMP_GROUP_LANES=1" 10 banana 100000
run "PDMP_REG=1" 50 iso 12500
run "PDMP_REG=0 PDMP_GROUP_LANES=4" 50 iso 12500
cat $O/suzz_forms.jsonl; tail -3 $O/suzz_err.log
