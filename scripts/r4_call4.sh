#!/bin/bash
# Round 4, GPU call 4: the new entry points' tests, then bench lines (default = driver command; banana; config 3; config 5 at BASELINE parameters)
O=gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_device_alloc.py tests/test_gpu_init_states.py tests/test_gpu_time_horizon_chunks.py tests/test_gpu_lane_kernel.py tests/test_gpu_reference_pins.py "tests/test_gpu_time_horizon.py::test_multi_device_entry_with_one_device" -x -q > $O/tests_call4.log 2>&1; echo "pytest rc=$?" >> $O/tests_call4.log
tail -3 $O/tests_call4.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_a.json 2> $O/bench_default_a.err; echo "bench a rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_b.json 2> $O/bench_default_b.err; echo "bench b rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 6 --warmup 2 --target banana --cpu-seconds 0 > $O/bench_zz_banana.json 2> $O/bench_zz_banana.err; echo "banana rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 6 --warmup 2 --sampler BPS --target banana --cpu-seconds 0 > $O/bench_config3.json 2> $O/bench_config3.err; echo "config3 rc=$?"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 5 --warmup 1 --sampler Boomerang --dim 100 --cpu-seconds 0 --sets 1 --placement-survey off > $O/bench_config5_1e5.json 2> $O/bench_config5_1e5.err; echo "config5 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 6 --warmup 2 --sampler Boomerang --dim 100 --chains 12500 --cpu-seconds 0 > $O/bench_config5_shard12500.json 2> $O/bench_config5_shard12500.err; echo "config5 shard rc=$?"
echo done
