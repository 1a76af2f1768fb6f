#!/bin/bash
# 128-byte lines in the one-lane kernels: the whole GPU suite, then random sweeps of the lane kernels against the checker
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/r5_tests9.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r5_tests9.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python scripts/lane_random_sweep.py 600 9051 25 52 1 > gpurun_out/r5/lane_sweep_line128_600.log 2>&1; echo "sweep 25..52 rc=$?"; tail -2 gpurun_out/r5/lane_sweep_line128_600.log
timeout -k 10 300 python scripts/lane_random_sweep.py 200 9052 1 24 1 > gpurun_out/r5/lane_sweep_line128_small_200.log 2>&1; echo "sweep 1..24 rc=$?"; tail -2 gpurun_out/r5/lane_sweep_line128_small_200.log
