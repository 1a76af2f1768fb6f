#!/bin/bash
# round-5 profile collection on the GPU box (profiles/collect.sh: rocprofv3 kernel trace + PMC passes) and the bench lines DESIGN.md quotes.
# usage: scripts/r5_profiles.sh [part]   (part 1 | 2 | 3: a gpurun call each)
set -o pipefail
PART=${1:-all}
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
bash profiles/collect.sh r5_zz_iso_lane || exit 1
bash profiles/collect.sh r5_bps_banana --sampler BPS --target banana || exit 1
bash profiles/collect.sh r5_boom_fd_baseline --sampler Boomerang --dim 100 --sets 1 || exit 1
bash profiles/collect.sh r5_boom_shard_12500 --sampler Boomerang --dim 100 --chains 12500 || exit 1
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
O=gpurun_out/r5; mkdir -p $O
B="python3 bench.py --cpu-seconds 0 --placement-survey off --pmc off"
# grid_size as a run-time parameter: Zig-Zag d = 50 at grid 5 / 20 on the LDS-form group kernel (now) and on the wave-per-chain kernel (rounds 1-4)
for g in 5 20; do
  $B --grid-size $g --steps 3 --warmup 1 > $O/bench_r5_zz_grid${g}_group.json 2>/dev/null || exit 1
  PDMP_GROUP_LANES=0 $B --grid-size $g --steps 2 --warmup 1 > $O/bench_r5_zz_grid${g}_wave.json 2>/dev/null || exit 1
done
PDMP_REG=0 PDMP_LANE=0 $B --steps 3 --warmup 1 > $O/bench_r5_zz_grid10_lds.json 2>/dev/null || exit 1
$B --sampler SpeedUpZigZagAD --grid-size 5 --n-sk 500 --steps 3 --warmup 1 > $O/bench_r5_suzz_grid5.json 2>/dev/null || exit 1
$B --sampler ForwardECMCAD --grid-size 8 --n-sk 200 --steps 3 --warmup 1 > $O/bench_r5_fecmc_grid8.json 2>/dev/null || exit 1
$B --sampler StickyZigZagAD --grid-size 5 --n-sk 500 --steps 3 --warmup 1 > $O/bench_r5_sticky_grid5.json 2>/dev/null || exit 1
# the other samplers under arithmetic contract v2
$B --sampler RHMC --n-sk 200 --steps 3 --warmup 1 > $O/bench_r5_rhmc.json 2>/dev/null || exit 1
$B --sampler ForwardECMCAD --n-sk 200 --steps 3 --warmup 1 > $O/bench_r5_fecmc.json 2>/dev/null || exit 1
$B --sampler StickyZigZagAD --n-sk 500 --steps 3 --warmup 1 > $O/bench_r5_sticky.json 2>/dev/null || exit 1
$B --sampler SpeedUpZigZagAD --n-sk 500 --steps 3 --warmup 1 > $O/bench_r5_suzz.json 2>/dev/null || exit 1
$B --sampler ZigZag --steps 3 --warmup 1 > $O/bench_r5_zigzag_fd.json 2>/dev/null || exit 1
$B --sampler BoomerangAD --dim 100 --refresh-rate 0.5 --steps 3 --warmup 1 > $O/bench_r5_boomerangad.json 2>/dev/null || exit 1
$B --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 10 --warmup 2 > $O/bench_r5_config4.json 2>/dev/null || exit 1
for c in 50000 25000 12500; do $B --chains $c --steps 10 --warmup 2 > $O/bench_r5_shard_$c.json 2>/dev/null || exit 1; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/bench_r5_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 3), "ms  %.3e ev/s  frac %.3f  %s" % (d["value"], d["roofline"]["frac"], d["roofline"]["kernel"][:50]))
    except Exception as e:
        print(f, "ERR", e)
PY
fi
if [ "$PART" = 3 ] || [ "$PART" = all ]; then
O=gpurun_out/r5; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r5_driver_command.json 2> $O/bench_r5_driver_command.err || exit 1
PDMP_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-seconds 0 > $O/bench_r5_rehearsal_gpus2.json 2> $O/bench_r5_rehearsal_gpus2.err || exit 1
PDMP_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-seconds 0 --gather moments > $O/bench_r5_rehearsal_gpus2_moments.json 2>/dev/null || exit 1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
tail -c 600 $O/bench_r5_driver_command.json
fi
