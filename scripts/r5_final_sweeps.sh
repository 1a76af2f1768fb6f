#!/bin/bash
# random sweeps of every sampler on the round's last library
mkdir -p gpurun_out/r5
timeout -k 10 800 python scripts/random_sweep.py 400 9081 > gpurun_out/r5/random_sweep_final_400.log 2>&1; echo "random sweep rc=$?"; tail -1 gpurun_out/r5/random_sweep_final_400.log
timeout -k 10 400 python scripts/sticky_random_sweep.py 150 9082 > gpurun_out/r5/sticky_sweep_final_150.log 2>&1; echo "sticky sweep rc=$?"; tail -1 gpurun_out/r5/sticky_sweep_final_150.log
