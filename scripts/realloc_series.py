"""Does the speed of a launch depend on where its output arrays lie?  One process, bench.py's default workload: the arrays are allocated,
timed for a few launches, freed (torch.cuda.empty_cache) and allocated again, several times.  Usage (GPU box): python scripts/realloc_series.py"""
import sys
sys.path.insert(0, ".")
import torch
import pdmpflux.jl_amd as P
from pdmpflux.jl_amd import ensemble
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
C, d, n_sk = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 50, 1000
x = torch.randn((C, d), dtype=torch.float64, device=dev, generator=g)
v = torch.randint(0, 2, (C, d), device=dev, generator=g).to(torch.float64) * 2 - 1
s = P.ZigZagAD(d, P.GaussianIso())
for rep in range(6):
    out = {}
    ts = []
    for i in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ensemble.sample_skeleton_device(s, n_sk, x, v, seed=i, out=out, check=False)
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    print("allocation", rep, "X at", hex(out["X"].data_ptr()), " ".join(f"{t:.1f}" for t in ts[1:]), flush=True)
    del out
    torch.cuda.empty_cache()
    if rep % 2 == 1:                       # every other time: hold a spacer so that the next arrays land elsewhere
        spacer = torch.empty((1 << 28) * (rep + 1), dtype=torch.uint8, device=dev)
