#!/bin/bash
# round-4 profile collection on the GPU box: rocprofv3 kernel trace + PMC passes of the tracked workloads (profiles/collect.sh), and the
# bench lines DESIGN.md quotes.  usage: scripts/r4_profiles.sh [part]   (part 1 | 2 | 3: a gpurun call each)
set -o pipefail
PART=${1:-all}
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
bash profiles/collect.sh r4_zz_iso_lane || exit 1
bash profiles/collect.sh r4_zz_banana_lane --target banana || exit 1
bash profiles/collect.sh r4_bps_banana --sampler BPS --target banana || exit 1
bash profiles/collect.sh r4_boom_fd_baseline --sampler Boomerang --dim 100 --sets 1 || exit 1
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
bash profiles/collect.sh r4_config4_dense --target dense --dim 1000 --chains 10000 --n-sk 200 || exit 1
bash profiles/collect.sh r4_rhmc_reg --sampler RHMC --n-sk 200 || exit 1
bash profiles/collect.sh r4_zz_12500 --chains 12500 || exit 1
bash profiles/collect.sh r4_boom_shard_12500 --sampler Boomerang --dim 100 --chains 12500 || exit 1
fi
if [ "$PART" = 3 ] || [ "$PART" = all ]; then
O=gpurun_out/r4; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_r4_driver_command.json 2> $O/bench_r4_driver_command.err || exit 1
python3 bench.py --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 10 --warmup 2 --cpu-seconds 0 > $O/bench_r4_config4.json 2>/dev/null || exit 1
python3 bench.py --sampler BPS --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off > $O/bench_r4_dense_bps.json 2>/dev/null || exit 1
python3 bench.py --sampler BoomerangAD --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off > $O/bench_r4_dense_boomerang.json 2>/dev/null || exit 1
PDMP_DENSE_GEMM=mfma python3 bench.py --sampler BPS --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off > $O/bench_r4_dense_bps_mfma.json 2>/dev/null || exit 1
PDMP_DENSE_GEMM=mfma python3 bench.py --sampler BoomerangAD --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off > $O/bench_r4_dense_boomerang_mfma.json 2>/dev/null || exit 1
python3 bench.py --gpus 1 --driver multi --steps 5 --warmup 2 > $O/bench_r4_driver_multi.json 2>/dev/null || exit 1
for c in 50000 25000 12500; do python3 bench.py --chains $c --steps 10 --warmup 2 --cpu-seconds 0 > $O/bench_r4_shard_$c.json 2>/dev/null || exit 1; done
python3 bench.py --sampler BPS --target banana --steps 6 --warmup 2 --cpu-seconds 0 > $O/bench_r4_config3.json 2>/dev/null || exit 1
python3 bench.py --target banana --steps 6 --warmup 2 --cpu-seconds 0 > $O/bench_r4_zz_banana.json 2>/dev/null || exit 1
python3 bench.py --sampler Boomerang --dim 100 --steps 5 --warmup 1 --cpu-seconds 0 --sets 1 --placement-survey off > $O/bench_r4_config5_1e5.json 2>/dev/null || exit 1
python3 scripts/bench_from_skeleton.py > $O/from_skeleton_r4.json 2>/dev/null || exit 1
python3 scripts/bench_dgemm.py > $O/bench_dgemm_r4.jsonl 2>/dev/null || exit 1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
tail -c 400 $O/bench_r4_driver_command.json
fi
