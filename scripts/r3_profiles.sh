#!/bin/bash
# round-3 profile collection on the GPU box: rocprofv3 kernel trace + PMC passes of the tracked workloads, the driver's bench command
set -o pipefail
bash profiles/collect.sh r3_zz_iso_lane || exit 1
bash profiles/collect.sh r3_zz_banana_lane --target banana || exit 1
bash profiles/collect.sh r3_bps_banana --sampler BPS --target banana || exit 1
PDMP_LANE_BPS=1 bash profiles/collect.sh r3_bps_banana_lane --sampler BPS --target banana || exit 1
bash profiles/collect.sh r3_boom_fd --sampler Boomerang --dim 100 --chains 50000 --refresh-rate 0.5 || exit 1
bash profiles/collect.sh r3_zz_12500 --chains 12500 || exit 1
bash profiles/collect.sh r3_zz_unaligned --chains 99999 --n-sk 997 --dim 51 || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r3_driver_command.json 2> gpurun_out/bench_r3_driver_command.err || exit 1
python3 bench.py --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 10 --warmup 2 --cpu-seconds 0 > gpurun_out/bench_r3_config4.json 2>/dev/null || exit 1
python3 bench.py --gpus 1 --driver multi --steps 5 --warmup 2 > gpurun_out/bench_r3_driver_multi.json 2>/dev/null || exit 1
python3 scripts/bench_from_skeleton.py > gpurun_out/from_skeleton_r3.json 2>/dev/null || exit 1
tail -c 600 gpurun_out/bench_r3_driver_command.json
