#!/bin/bash
# 128-byte lines in the pair form (d = 53 .. 104): random sweep of the pair forms against the checker
mkdir -p gpurun_out/r5
timeout -k 10 700 python scripts/lane_random_sweep.py 400 9071 53 104 2 > gpurun_out/r5/lane_pair_sweep_line128_400.log 2>&1; echo "pair sweep 53..104 rc=$?"; tail -1 gpurun_out/r5/lane_pair_sweep_line128_400.log
timeout -k 10 400 python scripts/lane_random_sweep.py 200 9072 25 52 2 > gpurun_out/r5/lane_quadpair_sweep_200.log 2>&1; echo "quad pair sweep 25..52 rc=$?"; tail -1 gpurun_out/r5/lane_quadpair_sweep_200.log
