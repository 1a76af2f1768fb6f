#!/bin/bash
# round 5, second GPU pass: the -m gpu suite, then BASELINE configs 2 / 3 / 5 and two shards on one GPU
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/r5_tests3.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/r5_tests3.log
if [ $rc -ge 124 ]; then exit $rc; fi
B="python bench.py --cpu-seconds 0 --placement-survey off"
$B --steps 10 --warmup 3 --pmc off > gpurun_out/r5d_b2.json 2> gpurun_out/r5d_b2.err && \
$B --sampler BPS --target banana --steps 5 --warmup 2 > gpurun_out/r5d_b3.json 2> gpurun_out/r5d_b3.err && \
$B --sampler Boomerang --dim 100 --steps 3 --warmup 1 > gpurun_out/r5d_b5.json 2> gpurun_out/r5d_b5.err && \
$B --sampler Boomerang --dim 100 --chains 12500 --steps 5 --warmup 2 --pmc off > gpurun_out/r5d_b5s.json 2> gpurun_out/r5d_b5s.err && \
$B --sampler BPS --target banana --chains 12500 --steps 5 --warmup 2 --pmc off > gpurun_out/r5d_b3s.json 2> gpurun_out/r5d_b3s.err && \
$B --target banana --steps 5 --warmup 2 --pmc off > gpurun_out/r5d_b2b.json 2> gpurun_out/r5d_b2b.err
echo "bench rc=$?"
python - <<'PY'
import json
for f in ("r5d_b2","r5d_b3","r5d_b5","r5d_b5s","r5d_b3s","r5d_b2b"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],3), "ms", "%.3e"%d["value"], "frac %.3f"%d["roofline"]["frac"], (d["roofline"].get("valu") or {}).get("instr_per_event"), (d["roofline"].get("valu") or {}).get("busy"))
    except Exception as e:
        print(f, "ERR", e)
PY
