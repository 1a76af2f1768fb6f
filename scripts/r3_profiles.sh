#!/bin/bash
# round-3 profile collection on the GPU box (final state of the round): rocprofv3 kernel trace + PMC passes of the tracked workloads,
# the driver's bench command and the other bench lines DESIGN.md quotes.  usage: scripts/r3_profiles.sh [part]   (part 1 | 2 | 3: a gpurun call each)
set -o pipefail
PART=${1:-all}
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
bash profiles/collect.sh r3_zz_iso_lane || exit 1
bash profiles/collect.sh r3_zz_banana_lane --target banana || exit 1
bash profiles/collect.sh r3_bps_banana --sampler BPS --target banana || exit 1
bash profiles/collect.sh r3_boom_fd --sampler Boomerang --dim 100 --chains 50000 --refresh-rate 0.5 || exit 1
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
bash profiles/collect.sh r3_zz_12500 --chains 12500 || exit 1
bash profiles/collect.sh r3_zz_25000_pair --chains 25000 || exit 1
bash profiles/collect.sh r3_zz_d100_pair --dim 100 || exit 1
bash profiles/collect.sh r3_zz_unaligned --chains 99999 --n-sk 997 --dim 51 || exit 1
fi
if [ "$PART" = 3 ] || [ "$PART" = all ]; then
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r3_driver_command.json 2> gpurun_out/bench_r3_driver_command.err || exit 1
python3 bench.py --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 10 --warmup 2 --cpu-seconds 0 > gpurun_out/bench_r3_config4.json 2>/dev/null || exit 1
python3 bench.py --sampler BPS --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off > gpurun_out/bench_r3_dense_bps.json 2>/dev/null || exit 1
python3 bench.py --sampler BoomerangAD --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off > gpurun_out/bench_r3_dense_boomerang.json 2>/dev/null || exit 1
python3 bench.py --gpus 1 --driver multi --steps 5 --warmup 2 > gpurun_out/bench_r3_driver_multi.json 2>/dev/null || exit 1
python3 bench.py --chains 25000 --steps 10 --warmup 2 --cpu-seconds 0 > gpurun_out/bench_r3_shard_25000.json 2>/dev/null || exit 1
python3 bench.py --chains 12500 --steps 10 --warmup 2 --cpu-seconds 0 > gpurun_out/bench_r3_shard_12500.json 2>/dev/null || exit 1
python3 bench.py --chains 50000 --steps 10 --warmup 2 --cpu-seconds 0 > gpurun_out/bench_r3_shard_50000.json 2>/dev/null || exit 1
python3 scripts/bench_from_skeleton.py > gpurun_out/from_skeleton_r3.json 2>/dev/null || exit 1
python3 scripts/bench_until.py > gpurun_out/until_bench.json 2>/dev/null || exit 1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
tail -c 600 gpurun_out/bench_r3_driver_command.json
fi
