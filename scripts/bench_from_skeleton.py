#!/usr/bin/env python3
"""Times sample_from_skeleton (src/sample.jl:475-513) on the output of bench.py's workload: HBM-bound,
algorithmic bytes = 8(2d+1) per skeleton event read + 8d per sample written."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pdmpflux.jl_amd as P
from pdmpflux.jl_amd import ensemble

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=100_000); ap.add_argument("--n-sk", type=int, default=1000)
ap.add_argument("--dim", type=int, default=50); ap.add_argument("--samples", type=int, default=1000)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((a.chains, a.dim), dtype=torch.float64, device=dev, generator=g)
v = torch.randint(0, 2, (a.chains, a.dim), device=dev, generator=g).to(torch.float64) * 2 - 1
s = P.ZigZagAD(a.dim, P.GaussianIso())
out = ensemble.sample_skeleton_device(s, a.n_sk, x, v, seed=1)
smp = torch.empty((a.chains, a.samples, a.dim), dtype=torch.float64, device=dev)
ensemble.sample_from_skeleton_device(s, a.samples, out["X"], out["V"], out["t"], out=smp)
torch.cuda.synchronize()
ensemble.kernel_timing(s, reset=True)
t0 = time.perf_counter()
for _ in range(a.steps):
    ensemble.sample_from_skeleton_device(s, a.samples, out["X"], out["V"], out["t"], out=smp)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps
tm = ensemble.kernel_timing(s)
ms = tm["kernel_ms_total"] / tm["launches"]
d = a.dim
# algorithmic bytes: every DISTINCT skeleton event a sample falls into is read once (x and v rows + its t entry; the binary search's
# other t reads stay in cache), every sample is written once -- counted exactly from the event times, not bounded by min(n_sk, N)
t = out["t"]
tm = torch.arange(1, a.samples + 1, dtype=torch.float64, device=dev).unsqueeze(0) * (t[:, -1:] / a.samples)
idx = torch.searchsorted(t, tm, right=True) - 1
touched = int((idx[:, 1:] != idx[:, :-1]).sum().item()) + a.chains
alg = touched * 8 * (2 * d + 1) + a.chains * a.samples * 8 * d
mean = smp.mean(dim=(0, 1)); var = smp.var(dim=(0, 1))
print(json.dumps({"kernel": "from_skeleton_kernel", "samples_per_s": a.chains * a.samples / wall, "kernel_ms": ms,
                  "algorithmic_GB": alg / 1e9, "distinct_events_read": touched, "events_in_skeleton": a.chains * a.n_sk, "achieved_GBs": alg / ms / 1e6, "frac_of_8TBs": alg / ms / 1e6 / 8000,
                  "max_abs_mean": float(mean.abs().max()), "var_range": [float(var.min()), float(var.max())]}))
