#!/bin/bash
# ForwardECMC in the register form: parity (its own suite, both forms, time horizon, tapes), then the A/B at d = 50, 10^5 chains x 200
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fecmc.py tests/test_gpu_parity.py tests/test_gpu_time_horizon.py tests/test_gpu_tape_replay.py tests/test_gpu_reference_pins.py -x -q -k "fecmc or ForwardECMC or forward_ecmc or register_and_lds or FECMC" > $O/tests_fecmc.log 2>&1; echo "pytest rc=$?"; tail -4 $O/tests_fecmc.log
B="python3 bench.py --sampler ForwardECMC --n-sk 200 --steps 3 --warmup 1 --cpu-seconds 0 --pmc off --placement-survey off"
$B > $O/bench_fecmc_reg1.json 2>/dev/null || exit 1
PDMP_REG=0 $B > $O/bench_fecmc_reg0_sameG.json 2>/dev/null || exit 1
PDMP_REG=0 PDMP_GROUP_LANES=16 $B > $O/bench_fecmc_reg0_G16.json 2>/dev/null || exit 1
$B --target banana > $O/bench_fecmc_banana_reg1.json 2>/dev/null || exit 1
PDMP_REG=0 PDMP_GROUP_LANES=16 $B --target banana > $O/bench_fecmc_banana_reg0_G16.json 2>/dev/null || exit 1
$B --dim 10 > $O/bench_fecmc_d10_reg1.json 2>/dev/null || exit 1
PDMP_REG=0 $B --dim 10 > $O/bench_fecmc_d10_reg0.json 2>/dev/null || exit 1
for f in $O/bench_fecmc_*.json; do python3 -c "
import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);print('$f','%.2f ms'%d['ms_per_step'],'%.3g ev/s'%d['value'],d['config'].get('lanes_per_chain'),d['roofline']['kernel'][:70])"; done
