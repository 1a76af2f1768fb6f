#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lane_kernel.py tests/test_gpu_dense_threads.py tests/test_gpu_device_alloc.py tests/test_gpu_grid_sizes.py tests/test_gpu_bench_sharded.py tests/test_gpu_reference_order.py -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/r5_tests4.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -12 gpurun_out/r5_tests4.log
if [ $rc -ge 124 ]; then exit $rc; fi
bash scripts/r5_profiles.sh 2
