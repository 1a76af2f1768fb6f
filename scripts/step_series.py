"""Per-step times of bench.py's default workload (one process, N steps, hipEvent per step): shows whether a slow run is slow throughout or
only while the GPU ramps up.  Usage (GPU box): python scripts/step_series.py [steps]"""
import sys, time
sys.path.insert(0, ".")
import torch
import pdmpflux.jl_amd as P
from pdmpflux.jl_amd import ensemble
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
C, d, n_sk = 100000, 50, 1000
x = torch.randn((C, d), dtype=torch.float64, device=dev, generator=g)
v = torch.randint(0, 2, (C, d), device=dev, generator=g).to(torch.float64) * 2 - 1
s = P.ZigZagAD(d, P.GaussianIso())
out = {}
ts = []
for i in range(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ensemble.sample_skeleton_device(s, n_sk, x, v, seed=i, out=out, check=False)
    b.record(); b.synchronize()
    ts.append(a.elapsed_time(b))
print(" ".join(f"{t:.1f}" for t in ts))
