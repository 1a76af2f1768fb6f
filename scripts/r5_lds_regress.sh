#!/bin/bash
# LDS-form kernels after the run-time grid_size (round 5) against round 4's figures: PDMP_REG=0
mkdir -p gpurun_out/r5
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off --steps 3 --warmup 1"
export PDMP_REG=0 PDMP_LANE=0
$B --sampler StickyZigZagAD --n-sk 500 > gpurun_out/r5/lds_sticky_iso.json 2>/dev/null
$B --sampler StickyZigZagAD --n-sk 500 --target banana > gpurun_out/r5/lds_sticky_banana.json 2>/dev/null
$B --sampler SpeedUpZigZagAD --n-sk 500 > gpurun_out/r5/lds_suzz_iso.json 2>/dev/null
$B --sampler ForwardECMCAD --n-sk 200 > gpurun_out/r5/lds_fecmc_iso.json 2>/dev/null
$B --sampler RHMC --n-sk 200 > gpurun_out/r5/lds_rhmc_iso.json 2>/dev/null
$B --sampler ZigZag --target banana > gpurun_out/r5/lds_zzfd_banana.json 2>/dev/null
$B --sampler BPS --target banana > gpurun_out/r5/lds_bps_banana.json 2>/dev/null
$B --target banana > gpurun_out/r5/lds_zz_banana.json 2>/dev/null
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/lds_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"][:60])
    except Exception as e:
        print(f, "ERR", e)
PY
