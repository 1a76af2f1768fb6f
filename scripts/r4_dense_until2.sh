#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_time_horizon_chunks.py tests/test_gpu_time_horizon.py -k "dense" -x -q > $O/tests_dense_until.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests_dense_until.log
[ $rc = 0 ] || exit $rc
for i in 1 2 3; do
  python3 bench.py --sampler ZigZagAD --target dense --dim 1000 --chains 10000 --n-sk 200 --steps 5 --warmup 2 --cpu-seconds 0 --pmc off --placement-survey off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 4', round(d['ms_per_step'],2), 'ms')"
done
# time horizon on the dense target at config-4 size: in-kernel chunks against the capacity-doubling re-run
python3 - <<'P'
import time, torch, numpy as np, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import pdmpflux.jl_amd as P
from pdmpflux.jl_amd import ensemble
from oracle import oracle as O
from helpers import make_pair, make_init
for kind, T in (("ZigZagAD", 0.25), ("BPS", 30.0)):
    s, cfg = make_pair(P, O, kind, 1000, "dense")
    C = 10000 if kind == "ZigZagAD" else 4000
    x, v = make_init(kind, 1000, C, seed=3)
    dev = torch.device("cuda:0")
    xd, vd = torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev)
    for name, fn in (("chunked (in-kernel horizon)", lambda: ensemble.sample_skeleton_until_chunked_device(s, T, xd, vd, seed=5, init_capacity=64, compact=False)),
                     ("re-run with doubling capacity", lambda: ensemble.sample_skeleton_until_device(s, T, xd, vd, seed=5, init_capacity=64, compact=False))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        cnt = out["count"]
        print(kind, "T =", T, name, round(dt * 1e3, 1), "ms; events", int(cnt.sum().item()), "max columns", int(cnt.max().item()), flush=True)
        del out
        torch.cuda.empty_cache()
P
