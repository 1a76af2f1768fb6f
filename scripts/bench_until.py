#!/usr/bin/env python3
"""Time-horizon variant at config-2 size: sample_skeleton(sampler, T, ...) for 10^5 Zig-Zag chains at d = 50, T = 50 (~1000 events per
chain), against the fixed-count launch of n_sk = 1000 -- time per run (wall, synchronised; medians of 5) and peak device memory.
    python scripts/bench_until.py [--chains 100000] [--T 50] > gpurun_out/until.json"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pdmpflux.jl_amd as P
from pdmpflux.jl_amd import ensemble

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=100_000)
ap.add_argument("--T", type=float, default=50.0)
ap.add_argument("--dim", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((a.chains, a.dim), dtype=torch.float64, device=dev, generator=g)
v = torch.randint(0, 2, (a.chains, a.dim), device=dev, generator=g).to(torch.float64) * 2 - 1
s = P.ZigZagAD(a.dim, P.GaussianIso())
res = {"chains": a.chains, "dim": a.dim, "T": a.T}


def timed(fn, reps=5):
    """Peak device memory from a first run out of an emptied allocator cache; times from the following runs, which reuse the cached
    blocks (a fresh hipMalloc of 80 GB takes about a second and is not what is being compared)."""
    torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    out = fn(0)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated()
    ts = []
    for i in range(1, reps + 1):
        out = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(i)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], peak, out


n_fixed = 1000
ms, peak, out = timed(lambda i: ensemble.sample_skeleton_device(s, n_fixed, x, v, seed=10 + i))
res["fixed_count"] = {"n_sk": n_fixed, "ms": ms, "peak_bytes": peak, "mean_t_end": float(out["t"][:, -1].mean().item())}
del out
for label, cap, compact in (("until_cap1024_chunks", 1024, False), ("until_cap1280_one_chunk", 1280, False), ("until_cap1024_packed", 1024, True)):
    st = {}
    ms, peak, out = timed(lambda i: ensemble.sample_skeleton_until_chunked_device(s, a.T, x, v, seed=10 + i, init_capacity=cap, stats=st, compact=compact))
    cnt = out["count"]
    dense = int(cnt.sum().item()) * 8 * (2 * a.dim + 1)
    res[label] = {"init_capacity": cap, "ms": ms, "peak_bytes": peak, "dense_bytes": dense, "peak_over_dense": peak / dense,
                  "ms_over_fixed": ms / res["fixed_count"]["ms"], "chunks": st.get("chunks"), "capacities": st.get("capacities"), "chains": st.get("chains"),
                  "events_mean": float(cnt.double().mean().item()), "events_max": int(cnt.max().item())}
    del out
# (the capacity-doubling re-run of round 2 would allocate chains x 2048 columns = 164 GB beside the 82 GB of its first attempt: not run here)
res["rerun_doubling_padded_bytes_computed"] = a.chains * (1024 + 2048) * 8 * (2 * a.dim + 1)
print(json.dumps(res, indent=1))
