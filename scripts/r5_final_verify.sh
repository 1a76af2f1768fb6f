#!/bin/bash
# the round's last library: the whole GPU suite, random sweeps of every kernel family against the checker, then the driver's command
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/r5_tests10.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r5_tests10.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 500 python scripts/lane_random_sweep.py 500 9061 25 52 1 > gpurun_out/r5/lane_sweep_final_500.log 2>&1; echo "lane sweep 25..52 rc=$?"; tail -1 gpurun_out/r5/lane_sweep_final_500.log
timeout -k 10 300 python scripts/lane_random_sweep.py 200 9062 1 24 1 > gpurun_out/r5/lane_sweep_final_small_200.log 2>&1; echo "lane sweep 1..24 rc=$?"; tail -1 gpurun_out/r5/lane_sweep_final_small_200.log
timeout -k 10 400 python scripts/lane_random_sweep.py 300 9063 27 104 2 > gpurun_out/r5/lane_pair_sweep_final_300.log 2>&1; echo "pair sweep rc=$?"; tail -1 gpurun_out/r5/lane_pair_sweep_final_300.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_r5_driver_command_final.json 2> gpurun_out/r5/bench_r5_driver_command_final.err; echo "bench rc=$?"
python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r5/bench_r5_driver_command_final.json").read().strip().splitlines()[-1])
print("value %.4g" % r["value"], "ms %.3f" % r["ms_per_step"], "frac %.3f" % r["roofline"]["frac"], "traffic %.4g" % (r["roofline"]["traffic"] or 0), "cpu", r.get("cpu_baseline", {}).get("value"))
PY
