#!/bin/bash
# round 5, first GPU pass: the whole -m gpu suite, then the three multi-GPU BASELINE configs on one GPU
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/r5_tests1.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r5_tests1.log
if [ $rc -ge 124 ]; then exit $rc; fi
python bench.py --steps 10 --warmup 3 --pmc off --cpu-seconds 0 --placement-survey off > gpurun_out/r5_b2.json 2> gpurun_out/r5_b2.err && \
python bench.py --sampler BPS --target banana --steps 5 --warmup 2 --cpu-seconds 0 > gpurun_out/r5_b3.json 2> gpurun_out/r5_b3.err && \
python bench.py --sampler Boomerang --dim 100 --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/r5_b5.json 2> gpurun_out/r5_b5.err
echo "bench rc=$?"
python - <<'PY'
import json
for f in ("r5_b2","r5_b3","r5_b5"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],3), "ms", "%.3e"%d["value"], "frac %.3f"%d["roofline"]["frac"], (d["roofline"].get("valu") or {}).get("instr_per_event"))
    except Exception as e:
        print(f, "ERR", e)
PY
